#!/bin/bash
# AUTO switch of the LS chain at 120 taps: every GPU test, then the published workload
mkdir -p gpurun_out/r04_c38; O=gpurun_out/r04_c38
timeout 600 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 600 python bench.py --workload prconfig --steps 3 --no-cpu > $O/bench_prconfig.json 2> $O/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_c38/bench_prconfig.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['resident_stages_ms'], d['host_to_host']['frames_per_s'], d['resident_vs_pipelined_maps_max_err_of_peak'])
P
