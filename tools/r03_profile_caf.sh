#!/bin/bash
# rocprofv3 passes over the CAF micro-benchmark (tools/caf_bench.py): kernel trace + stats, then one --pmc pass per
# counter group (never combined with a trace domain).  usage: tools/r03_profile_caf.sh <tag> <caf_bench args...>
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_caf_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $R/tools/caf_bench.py "$@" > $out/trace.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o pmc -- python $R/tools/caf_bench.py "$@" > $out/pmc_$i.log 2>&1
done
python3 - "$out" "$*" <<'PY'
import csv, glob, sys, collections, json
out, args = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
line = [l for l in open(out + "/trace.log") if l.startswith("{")]
with open(out + "/summary.md", "w") as fh:
    fh.write(f"rocprofv3 over `python tools/caf_bench.py {args}` (kernel trace + one --pmc pass per counter group)\n\n")
    if line:
        fh.write("bench line: `" + line[-1].strip() + "`\n\n")
    fh.write("| kernel | launches | avg us / launch | FETCH_SIZE x2 MB / launch | WRITE_SIZE MB / launch | VALU active / wave-cycles | WAIT_ANY | WAIT_INST_ANY | LDS conflict / LDS active |\n|---|---|---|---|---|---|---|---|---|\n")
    for k in sorted(set(list(acc) + list(dur))):
        if not k.startswith(("caf_", "doppler", "shift_", "transpose", "ls_", "nlms")) and "rocfft" not in k.lower() and "fft" not in k.lower():
            continue
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        d = dur.get(k, [])
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        fh.write(f"| `{k[:60]}` | {len(d)} | {sum(d) / max(len(d), 1):.1f} | {2 * c.get('FETCH_SIZE', 0) / 1e3:.2f} | {c.get('WRITE_SIZE', 0) / 1e3:.2f} | "
                 f"{100 * c.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} % | {100 * c.get('SQ_WAIT_ANY', 0) / wc:.0f} % | {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} % | "
                 f"{100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):.0f} % |\n")
    fh.write("\nFETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950.\n")
print(open(out + "/summary.md").read())
PY
