"""summary of tools/trace_prconfig.sh's kernel trace (library kernels only)"""
import csv
import glob
import json
import sys

O = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)[0])))
d = json.load(open(O + "/trace.json"))
out = ["# Round 5 -- rocprofv3 --kernel-trace --stats -- python bench.py --workload prconfig --frames 480 --steps 3 --no-cpu", "",
       f"The published workload (PRconfig.yaml as shipped) on 480 frames, final kernels of round 5; bench line under the profiler: "
       f"{d['value']:.0f} frames/s resident, host to host {d['host_to_host']['frames_per_s']:.0f} frames/s.", "",
       "Library kernels (+ the runtime's copy kernel of the host-to-host legs); the torch / MIOpen kernels that synthesise the recordings are left out.", "",
       "| kernel | calls | avg us | total ms |", "|---|---|---|---|"]
OURS = ("caf_", "doppler_col", "ls_", "levinson", "nlms", "frontend", "cfar", "__amd_rocclr_copyBuffer")
for r in rows:
    n = r["Name"]
    if not n.replace("void ", "").startswith(OURS):     # torch / MIOpen kernels that synthesise the recordings before the timed region
        continue
    out.append(f"| `{n.replace('void ', '').split('(')[0][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
               f"{float(r['TotalDurationNs']) / 1e6:.2f} |")
open(O + "/r05_trace_prconfig_kernel_stats.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:26]))
