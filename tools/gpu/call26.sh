#!/bin/bash
mkdir -p gpurun_out/r04_c26; O=gpurun_out/r04_c26
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cfar" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/cfar_bench.py > $O/cfar_bench.txt 2>&1; tail -3 $O/cfar_bench.txt
