#!/bin/bash
mkdir -p gpurun_out/s4
cd "$(dirname "$0")/.."
L=$PWD/passiveradar_amd
for lib in libprcore_tt.so libprcore_fvt.so; do
  PRCORE_LIB=$L/$lib timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "caf or doppler or multi or cfg5" > gpurun_out/s4/pytest_$lib.log 2>&1
  echo "$lib pytest rc=$?"; tail -2 gpurun_out/s4/pytest_$lib.log
done
B=gpurun_out/s4/ab.jsonl
: > $B
run() { timeout 300 env "$@" 2>>gpurun_out/s4/ab.err | tail -1 >> $B; }
for rep in 1 2; do
for lib in libprcore.so libprcore_tt.so; do
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg5 --frames 16 --tag team
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg3 --frames 128 --tag team
done
for lib in libprcore.so libprcore_fve.so libprcore_ft.so libprcore_fvt.so; do
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg2 --frames 256 --tag wave
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg1 --frames 1024 --tag wave
done
done
python - <<'PY'
import json
for ln in open("gpurun_out/s4/ab.jsonl"):
    try: d = json.loads(ln)
    except Exception: print("??", ln[:200]); continue
    print(d["lib"].split("/")[-1], d["shape"], "seg_ms", round(d["segments_ms"], 3), "us/surface", round(d["seg_us_per_surface"], 2), "GB/s", round(d["seg_GBps"]), "dop_ms", round(d["doppler_ms"], 3), "exec_ms", round(d["execute_ms"], 3))
PY
