#!/bin/bash
mkdir -p gpurun_out/r04_c39; O=gpurun_out/r04_c39
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q -k "front_end or prconfig" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 120 python tools/frontend_bench.py 2 3 4 2 2>&1 | grep "method" | tee $O/fe_bench.txt
