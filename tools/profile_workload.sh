#!/bin/bash
# rocprofv3 --kernel-trace --stats of a secondary workload: tools/profile_workload.sh cfg3 1024   (on the GPU box)
w=$1; fr=$2
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -o $w -- python $R/bench.py --workload $w --frames $fr --no-cpu --steps 3 --warmup 1 > $R/gpurun_out/bench_$w.log 2>&1
grep '^{' $R/gpurun_out/bench_$w.log | tail -1 | cut -c1-200
