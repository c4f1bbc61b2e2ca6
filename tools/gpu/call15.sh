#!/bin/bash
# round 4, GPU call 15: bench.py's weak-scaling gather plumbing (sub-batch gathers under the step's compute) through a
# communicator of one rank; the headline line with and without it
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c15; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py -q -k "independent_pass or bench_workload or gather" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for g in none prc none prc; do
  timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 --gather $g 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gather=$g', round(d['value']), round(d['ms_per_step'],2), d['gather_path'])"
done
