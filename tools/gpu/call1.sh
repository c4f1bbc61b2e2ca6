#!/bin/bash
# round 4, GPU call 1: packed-f32 butterflies -- microbenchmark, A/B on the CAF shapes and the headline step, parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c1; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 120 tools/ubench/dft16pk ) > $O/dft16pk.txt 2>&1
PK=$R/build/libprcore_pk.so
for rep in 1 2; do
  for lib in default $PK; do
    L=""; [ $lib != default ] && L="PRCORE_LIB=$lib"
    env $L timeout 200 python tools/caf_bench.py --shape cfg5 --frames 16 --tag r$rep >> $O/caf.jsonl 2>>$O/caf.err
    env $L timeout 200 python tools/caf_bench.py --shape cfg3 --frames 64 --tag r$rep >> $O/caf.jsonl 2>>$O/caf.err
    env $L timeout 200 python tools/caf_bench.py --shape cfg2 --frames 256 --tag r$rep >> $O/caf.jsonl 2>>$O/caf.err
  done
done
for lib in default $PK default $PK; do
  L=""; [ $lib != default ] && L="PRCORE_LIB=$lib"
  env $L timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 2>>$O/bench.err | tail -1 >> $O/bench.jsonl
done
PRCORE_LIB=$PK timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_pk.txt 2>&1
tail -5 $O/pytest_pk.txt
cat $O/dft16pk.txt
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print(d["lib"][-20:], d["shape"], d["tag"], "seg us/surf %.2f"%d["seg_us_per_surface"], "dop ms %.4f"%d["doppler_ms"], "exec us/surf %.2f"%d["exec_us_per_surface"])
for l in open("$O/bench.jsonl"):
    try: d=json.loads(l)
    except Exception: print("bad line", l[:100]); continue
    print(round(d["value"]), round(d["ms_per_step"],2), {k:round(v["avg_ms_per_launch"],4) for k,v in d["kernels"].items()})
PY
