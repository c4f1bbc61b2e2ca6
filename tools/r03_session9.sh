#!/bin/bash
mkdir -p gpurun_out/s9
cd "$(dirname "$0")/.."
L=$PWD/passiveradar_amd
PRCORE_LIB=$L/libprcore_fpr.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "caf or doppler or multi or cfg5" > gpurun_out/s9/pytest_fpr.log 2>&1
echo "pr pytest rc=$?"; tail -2 gpurun_out/s9/pytest_fpr.log
B=gpurun_out/s9/ab.jsonl
: > $B
run() { timeout 300 env "$@" 2>>gpurun_out/s9/ab.err | tail -1 >> $B; }
for rep in 1 2 3; do
for lib in libprcore.so libprcore_fpr.so; do
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg2 --frames 256 --tag wave
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg1 --frames 1024 --tag wave
done
done
python - <<'PY'
import json
for ln in open("gpurun_out/s9/ab.jsonl"):
    try: d = json.loads(ln)
    except Exception: print("??", ln[:200]); continue
    print(d["lib"].split("/")[-1], d["shape"], "seg_ms", round(d["segments_ms"], 3), "us/surface", round(d["seg_us_per_surface"], 2), "GB/s", round(d["seg_GBps"]))
PY
