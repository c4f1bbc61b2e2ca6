#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a shorter run -- never combined with trace domains
# Raw CSVs land in gpurun_out/prof_<tag>/; tools/profile_summarize.py turns them into profiles/*.md / traffic_latest.json.
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python $R/bench.py --no-cpu > $out/bench_trace.log 2>&1
grep '^{' $out/bench_trace.log | tail -1 > $out/bench_line.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- python $R/bench.py --no-cpu --frames 64 --steps 2 --warmup 1 > $out/bench_pmc_$c.log 2>&1
done
find $out -name "*.csv" | sed "s|$R/||"
