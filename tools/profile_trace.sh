#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench command: tools/profile_trace.sh <tag> <bench args...>   (GPU box)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $R/bench.py --no-cpu "$@" > $out/bench.log 2>&1
grep '^{' $out/bench.log | tail -1 > $out/bench_line.json
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/kernel_stats.md", "w") as fh:
    fh.write("| kernel | calls | avg us / launch | % of GPU time |\n|---|---|---|---|\n")
    for r in rows[:14]:
        fh.write(f"| {r['Name'].replace('void ', '').split('(')[0][:70]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |\n")
print(open(out + "/kernel_stats.md").read())
PY
cut -c1-300 $out/bench_line.json
