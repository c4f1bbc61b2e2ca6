#!/bin/bash
mkdir -p gpurun_out/s3
cd "$(dirname "$0")/.."
L=$PWD/passiveradar_amd
tools/ubench/dft16 > gpurun_out/s3/dft16.log 2>&1; cat gpurun_out/s3/dft16.log
tools/ubench/pkfma > gpurun_out/s3/pkfma.log 2>&1; cat gpurun_out/s3/pkfma.log
B=gpurun_out/s3/ab.jsonl
: > $B
run() { timeout 300 env "$@" 2>>gpurun_out/s3/ab.err | tail -1 >> $B; }
for lib in libprcore.so libprcore_noload.so libprcore_nobar.so libprcore_nox2.so libprcore_nox12.so libprcore_nolds_noload.so libprcore_w2.so; do
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg5 --frames 16 --tag exp
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg3 --frames 128 --tag exp
done
python - <<'PY'
import json
for ln in open("gpurun_out/s3/ab.jsonl"):
    try: d = json.loads(ln)
    except Exception: print("??", ln[:200]); continue
    print(d["lib"].split("/")[-1], d["shape"], "seg_ms", round(d["segments_ms"], 3), "us/surface", round(d["seg_us_per_surface"], 1), "GB/s", round(d["seg_GBps"]))
PY
