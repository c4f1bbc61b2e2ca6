#!/bin/bash
# round 4, GPU call 7: full suite (tile-major buffer, double-accumulating FIR fallback); wavefront CAF kernel at four waves per SIMD
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c7; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt | head -5
for rep in 1 2; do for v in default w4_pk w4_sc w3_sc; do
  L=""; [ $v != default ] && L="PRCORE_LIB=$R/build/libprcore_$v.so"
  env $L timeout 150 python tools/caf_bench.py --shape cfg2 --frames 256 --tag $v >> $O/caf.jsonl 2>>$O/err.txt
  env $L timeout 150 python tools/caf_bench.py --shape cfg1 --frames 1024 --tag $v >> $O/caf.jsonl 2>>$O/err.txt
done; done
for v in default w4_pk w4_sc default w4_pk; do
  L=""; [ $v != default ] && L="PRCORE_LIB=$R/build/libprcore_$v.so"
  env $L timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],2), {k:round(v['avg_ms_per_launch'],4) for k,v in d['kernels'].items()})"
done
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print(d["tag"], d["shape"], "seg us/surf %.3f"%d["seg_us_per_surface"], "exec us/surf %.3f"%d["exec_us_per_surface"])
PY
