#!/bin/bash
mkdir -p gpurun_out/s7
cd "$(dirname "$0")/.."
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s7/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/s7/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/s7/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/s7/bench_default.log | cut -c1-600
for w in cfg5 cfg4 cfg3; do
  timeout 900 python bench.py --workload $w --no-cpu > gpurun_out/s7/bench_$w.log 2>&1; echo "$w rc=$?"; tail -1 gpurun_out/s7/bench_$w.log | cut -c1-400
done
