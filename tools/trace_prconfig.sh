#!/bin/bash
# Kernel trace of the published workload (bench.py --workload prconfig) -> gpurun_out/<dir>/r05_trace_prconfig_kernel_stats.md
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-trace_prconfig}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --workload prconfig --frames 480 --steps 3 --no-cpu > $O/trace.log 2>&1
grep "^{" $O/trace.log | tail -1 > $O/trace.json
cd $R; python tools/trace_prconfig_summary.py $O
