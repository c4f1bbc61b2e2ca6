#!/bin/bash
# round 4, GPU call 10: the illuminators of a "turns" frame in one launch per stage (segment kernel z dimension, Doppler channels)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c10; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -q -k "multi or cfg5 or caf_golden or caf_full" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for m in turns shared turns; do timeout 150 python tools/caf_bench.py --shape cfg5 --frames 16 --nref 4 --multi $m >> $O/caf.jsonl 2>>$O/err.txt; done
for i in 1 2; do timeout 300 python bench.py --workload cfg5 --no-cpu 2>>$O/err.txt | tail -1 >> $O/bench_cfg5.jsonl; done
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print("multi=%s multi us/frame %.1f singles us/frame %.1f" % (d["multi"], d["multi_us_per_frame"], d["singles_ms"]*1e3/d["frames"]))
for l in open("$O/bench_cfg5.jsonl"):
    d=json.loads(l); print("bench cfg5", round(d["value"],1), d["hbm_frac_of_peak"], {k:round(v["avg_ms_per_launch"],4) for k,v in d["kernels"].items()})
PY
