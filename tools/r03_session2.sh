#!/bin/bash
mkdir -p gpurun_out/s2
cd "$(dirname "$0")/.."
L=$PWD/passiveradar_amd
for mode in 1 0; do
  PRC_CAF_MULTI_MODE=$mode timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "caf or doppler or multi or cfg5 or cache_sized" > gpurun_out/s2/pytest_caf_mode$mode.log 2>&1
  echo "mode $mode pytest rc=$?"; tail -3 gpurun_out/s2/pytest_caf_mode$mode.log
done
B=gpurun_out/s2/ab.jsonl
: > $B
run() { timeout 300 env "$@" 2>>gpurun_out/s2/ab.err | tail -1 >> $B; }
run PRC_CAF_MULTI_MODE=1 python tools/caf_bench.py --shape cfg5 --frames 8 --nref 4 --tag multi_scratch
run PRC_CAF_MULTI_MODE=0 python tools/caf_bench.py --shape cfg5 --frames 8 --nref 4 --tag multi_regs
run PRC_CAF_MULTI_MODE=1 PRCORE_LIB=$L/libprcore_msf.so python tools/caf_bench.py --shape cfg5 --frames 8 --nref 4 --tag multi_scratch_factored
run PRC_CAF_MULTI_MODE=1 python tools/caf_bench.py --shape cfg3 --frames 64 --nref 4 --tag multi_scratch
run PRC_CAF_MULTI_MODE=0 python tools/caf_bench.py --shape cfg3 --frames 64 --nref 4 --tag multi_regs
for d in 1 2; do
  run python tools/caf_bench.py --shape cfg1 --frames 512 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg2 --frames 256 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg3 --frames 128 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg5 --frames 16 --doppler $d --tag doppler
done
python - <<'PY'
import json
for ln in open("gpurun_out/s2/ab.jsonl"):
    try: d = json.loads(ln)
    except Exception: print("??", ln[:200]); continue
    keys = ["tag", "shape", "frames", "nref", "doppler", "segments_ms", "doppler_ms", "execute_ms", "multi_ms", "singles_ms", "seg_GBps"]
    print({k: (round(d[k], 3) if isinstance(d.get(k), float) else d.get(k)) for k in keys if k in d}, d["lib"].split("/")[-1])
PY
