#!/bin/bash
# round 4, GPU call 4: shared kernel without prefetch; long-FIR CAF (parity + FFT form vs time-domain at config 2);
# prconfig raw->frame golden test; the prconfig workload; full GPU suite on the new default build; headline line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c4; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in m_noahead m_noahead_tw2; do
  PRCORE_LIB=$R/build/libprcore_$v.so timeout 150 python tools/caf_bench.py --shape cfg5 --frames 16 --nref 4 --multi shared --tag $v >> $O/caf.jsonl 2>>$O/caf.err
done
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "long_filter or caf_golden" > $O/pytest_long.txt 2>&1; tail -3 $O/pytest_long.txt
timeout 300 python -m pytest tests/test_gpu_stream.py -x -q -k "prconfig" > $O/pytest_prconfig.txt 2>&1; tail -3 $O/pytest_prconfig.txt
for m in 2 1; do
  timeout 200 python tools/caf_bench.py --shape cfg2 --frames 16 --long-fir --caf-method $m --tag longfir_m$m --reps 3 >> $O/caf_long.jsonl 2>>$O/caf.err
  timeout 200 python tools/caf_bench.py --shape cfg1 --frames 64 --long-fir --caf-method $m --tag longfir_m$m --reps 3 >> $O/caf_long.jsonl 2>>$O/caf.err
done
timeout 400 python bench.py --workload prconfig --frames 240 --steps 2 --no-cpu > $O/bench_prconfig_240.json 2> $O/bench_prconfig_240.err; tail -c 1500 $O/bench_prconfig_240.json; tail -5 $O/bench_prconfig_240.err
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt
timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 2>>$O/bench.err | tail -1 > $O/bench_default.json
python - <<PY
import json
for f in ("$O/caf.jsonl", "$O/caf_long.jsonl"):
    for l in open(f):
        d=json.loads(l); print(d["tag"], d["shape"], "method", d["method"], "seg us/surf %.2f"%d["seg_us_per_surface"], "exec us/surf %.2f"%d["exec_us_per_surface"], ("multi us/frame %.1f" % d["multi_us_per_frame"]) if "multi_us_per_frame" in d else "")
d=json.load(open("$O/bench_default.json")); print(round(d["value"]), round(d["ms_per_step"],2), {k:round(v["avg_ms_per_launch"],4) for k,v in d["kernels"].items()}, d["roofline"]["frac"])
PY
