#!/bin/bash
mkdir -p gpurun_out/s5
cd "$(dirname "$0")/.."
L=$PWD/passiveradar_amd
for lib in libprcore_mn2.so libprcore_mn2fp.so; do
  PRC_CAF_MULTI_MODE=0 PRCORE_LIB=$L/$lib timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "multi or cfg5" > gpurun_out/s5/pytest_$lib.log 2>&1
  echo "$lib pytest rc=$?"; tail -2 gpurun_out/s5/pytest_$lib.log
done
B=gpurun_out/s5/ab.jsonl
: > $B
run() { timeout 300 env "$@" 2>>gpurun_out/s5/ab.err | tail -1 >> $B; }
for fr in 8 16; do
for lib in libprcore.so libprcore_mn2.so libprcore_mn2fp.so libprcore_mn2p.so; do
  run PRC_CAF_MULTI_MODE=0 PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg5 --frames $fr --nref 4 --tag multi
done
done
python - <<'PY'
import json
for ln in open("gpurun_out/s5/ab.jsonl"):
    try: d = json.loads(ln)
    except Exception: print("??", ln[:200]); continue
    print(d["lib"].split("/")[-1], d["shape"], d["frames"], "multi_ms", round(d["multi_ms"], 3), "singles_ms", round(d["singles_ms"], 3), "us/frame", round(d["multi_us_per_frame"], 1), "seg_ms", round(d["segments_ms"], 3), "dop_ms", round(d["doppler_ms"], 3))
PY
