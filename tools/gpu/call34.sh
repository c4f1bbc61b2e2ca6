#!/bin/bash
# final check of the tree after the front-end staging changes
mkdir -p gpurun_out/r04_c34; O=gpurun_out/r04_c34
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 200 python tests/fuzz_parity.py 71 60 1 $O/fuzz_short.md > $O/fuzz1.txt 2>&1; tail -1 $O/fuzz1.txt
timeout 900 python bench.py --workload prconfig --steps 3 > $O/bench_prconfig.json 2> $O/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_c34/bench_prconfig.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['resident_stages_ms'], d['host_to_host']['frames_per_s'], d['host_to_host']['with_store']['frames_per_s'], d['roofline']['frac'], d['roofline']['ms_per_launch'])
P
