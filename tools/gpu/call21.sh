#!/bin/bash
# front end: the group form against one output per thread
mkdir -p gpurun_out/r04_c21; O=gpurun_out/r04_c21
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q -k "front_end or prconfig or raw_to_frames" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 300 python tools/frontend_bench.py > $O/fe_bench.txt 2>&1; tail -5 $O/fe_bench.txt
