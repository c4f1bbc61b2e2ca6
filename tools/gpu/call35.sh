#!/bin/bash
mkdir -p gpurun_out/r04_c35; O=gpurun_out/r04_c35
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q -k "front_end or prconfig or raw_to_frames" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 120 python tools/frontend_bench.py 2 2>&1 | grep "method 2"
