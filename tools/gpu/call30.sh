#!/bin/bash
# kernel traces of the rows either side of the path (front end, CFAR) and of the published workload's step
mkdir -p gpurun_out/r04_c30; O=$PWD/gpurun_out/r04_c30; R=$PWD
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end_kernel_forms" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_fe -o t -- python tools/frontend_bench.py > $O/t_fe.log 2>&1)
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_cfar -o t -- python tools/cfar_bench.py > $O/t_cfar.log 2>&1)
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_prconfig -o t -- python bench.py --workload prconfig --frames 480 --steps 3 --no-cpu > $O/t_prconfig.log 2>&1)
cd $R
python3 - <<'P'
import csv, glob, os
O = "gpurun_out/r04_c30"
out = ["# Round 4 -- rocprofv3 --kernel-trace --stats of the rows either side of the path\n"]
for name, cmd in (("t_fe", "python tools/frontend_bench.py"), ("t_cfar", "python tools/cfar_bench.py"),
                  ("t_prconfig", "python bench.py --workload prconfig --frames 480 --steps 3 --no-cpu")):
    fs = glob.glob(f"{O}/{name}/**/*kernel_stats.csv", recursive=True)
    out.append(f"\n## `rocprofv3 --kernel-trace --stats -- {cmd}`\n")
    log = open(f"{O}/{name}.log").read().strip().splitlines()
    for l in log:
        if l.startswith(("front end", "CFAR_2D", "max |")):
            out.append("    " + l)
        if l.startswith("{"):
            import json
            d = json.loads(l)
            out.append(f"    bench line: {d['value']:.0f} frames/s, {d['ms_per_step']:.1f} ms per {d['config']['frames_per_gpu_per_step']} frames; with store {d['with_store']['frames_per_s']:.0f}")
    if not fs:
        out.append("(no stats file)")
        continue
    rows = list(csv.DictReader(open(fs[0])))
    out.append("\n| kernel | calls | avg us | % |\n|---|---|---|---|")
    for r in rows[:12]:
        out.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
open(f"{O}/r04_trace_frontend_cfar_prconfig_kernel_stats.md", "w").write("\n".join(out) + "\n")
print("\n".join(out)[-3500:])
P
