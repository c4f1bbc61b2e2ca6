#!/bin/bash
# GPU box: parity of the eight-wavefront 4096-point CAF segment kernel, then its A/B against the four-wavefront one
#   tools/ab_team8.sh <outdir> [variant libs ...]
out=${1:-gpurun_out/ab_team8}; shift
mkdir -p $out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_fft_team.py -q -x 2>&1 | tail -3 | tee $out/tests_fft.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "caf_team or cfg3_digest or cfg5_digest" 2>&1 | tail -3 | tee $out/tests_caf.log
run() { # name, lib ('' = shipped), args...
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then PRCORE_LIB=$R/build/libprcore_$lib.so python tools/caf_bench.py --tag $name "$@" 2>/dev/null | grep '^{' >> $out/ab.jsonl
  else python tools/caf_bench.py --tag $name "$@" 2>/dev/null | grep '^{' >> $out/ab.jsonl; fi
}
: > $out/ab.jsonl
for rep in 1 2; do
  run team4 "" --shape cfg5 --frames 16 --nref 4 --multi turns --team8 0
  run team8_6w "" --shape cfg5 --frames 16 --nref 4 --multi turns --team8 1
  for v in "$@"; do run team8_$v $v --shape cfg5 --frames 16 --nref 4 --multi turns --team8 1; done
  run team4 "" --shape cfg3 --frames 64 --team8 0
  run team8_6w "" --shape cfg3 --frames 64 --team8 1
  for v in "$@"; do run team8_$v $v --shape cfg3 --frames 64 --team8 1; done
done
python - <<PY
import json
rows=[json.loads(l) for l in open("$out/ab.jsonl")]
print("| build | shape | segment kernel us / surface | multi us / frame (4 illuminators) |")
print("|---|---|---|---|")
for r in rows:
    print(f"| {r['tag']} | {r['shape']} | {r['seg_us_per_surface']:.2f} | {r.get('multi_us_per_frame', float('nan')):.1f} |")
PY
