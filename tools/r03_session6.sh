#!/bin/bash
mkdir -p gpurun_out/s6
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "caf or doppler or multi or cfg5 or cache_sized" > gpurun_out/s6/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/s6/pytest.log
B=gpurun_out/s6/ab.jsonl
: > $B
run() { timeout 300 env "$@" 2>>gpurun_out/s6/ab.err | tail -1 >> $B; }
for lib in libprcore.so libprcore_dk.so; do export PRCORE_LIB=$PWD/passiveradar_amd/$lib; for d in 2; do
  run python tools/caf_bench.py --shape cfg1 --frames 1024 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg2 --frames 256 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg3 --frames 128 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg5 --frames 16 --doppler $d --tag doppler
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/s6/ab.jsonl"):
    try: d = json.loads(ln)
    except Exception: print("??", ln[:200]); continue
    print(d["lib"].split("/")[-1], d["shape"], d["frames"], "doppler", d["doppler"], "dop_ms", round(d["doppler_ms"], 3), "seg_ms", round(d["segments_ms"], 3), "exec_ms", round(d["execute_ms"], 3))
PY
