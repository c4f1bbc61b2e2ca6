#!/bin/bash
# round-3 GPU session 1: parity of the new CAF paths, then A/B of the wide-span kernels
mkdir -p gpurun_out/s1
cd "$(dirname "$0")/.."
L=$PWD/passiveradar_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "caf or doppler or multi or cfg5 or cache_sized" > gpurun_out/s1/pytest_caf.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1/pytest_caf.log
tail -5 gpurun_out/s1/pytest_caf.log
B=gpurun_out/s1/ab.jsonl
: > $B
run() { timeout 300 env "$@" 2>>gpurun_out/s1/ab.err | tail -1 >> $B; }
# config 5, four illuminators: multi kernel variants
for lib in libprcore.so libprcore_m0.so libprcore_ml.so libprcore_m3.so; do
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg5 --frames 8 --nref 4 --tag multi
done
# single-reference team kernel variants at configs 3 and 5
for lib in libprcore.so libprcore_ve.so libprcore_vef.so; do
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg3 --frames 128 --tag single
  run PRCORE_LIB=$L/$lib python tools/caf_bench.py --shape cfg5 --frames 16 --tag single
done
# Doppler stage: rocFFT path vs column kernel; group size
for d in 1 2; do
  run python tools/caf_bench.py --shape cfg2 --frames 256 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg3 --frames 128 --doppler $d --tag doppler
  run python tools/caf_bench.py --shape cfg5 --frames 16 --doppler $d --tag doppler
done
for mb in 32 64 192 100000; do
  run python tools/caf_bench.py --shape cfg2 --frames 256 --group-mb $mb --tag group
  run python tools/caf_bench.py --shape cfg5 --frames 8 --nref 4 --group-mb $mb --tag group
done
cat $B
