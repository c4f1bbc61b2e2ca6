#!/bin/bash
# Everything the round's profiles/ are built from, in one gpurun call:  tools/profile_all.sh r02
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
tools/profile_round.sh $tag > /dev/null 2>&1
python bench.py > gpurun_out/prof_$tag/bench_full_line.json 2> gpurun_out/prof_$tag/bench_full.err
for w in cfg3 cfg4 cfg5; do
  python bench.py --workload $w --no-cpu > gpurun_out/prof_$tag/bench_$w.json 2> /dev/null
done
tools/profile_trace.sh ${tag}_cfg3 --workload cfg3 --frames 256 --steps 2 --warmup 1 > /dev/null 2>&1
tools/profile_trace.sh ${tag}_cfg5 --workload cfg5 --steps 20 --warmup 2 > /dev/null 2>&1
tools/profile_pmc.sh ${tag}_cfg5 --workload cfg5 --steps 4 --warmup 1 > /dev/null 2>&1
tools/profile_pmc.sh ${tag}_cfg3 --workload cfg3 --no-clutter --frames 64 --steps 2 --warmup 1 > /dev/null 2>&1
tools/profile_pmc.sh ${tag}_cfg2 --frames 256 --steps 2 --warmup 1 > /dev/null 2>&1
ls gpurun_out/prof_$tag gpurun_out/trace_${tag}_cfg3 gpurun_out/pmc_${tag}_cfg5 | head -40
cut -c1-200 gpurun_out/prof_$tag/bench_full_line.json
