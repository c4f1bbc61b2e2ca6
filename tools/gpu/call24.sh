#!/bin/bash
mkdir -p gpurun_out/r04_c24; O=gpurun_out/r04_c24
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -x -q -k "front_end or prconfig or raw_to_frames or resampl" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/frontend_bench.py > $O/fe_bench.txt 2>&1; tail -4 $O/fe_bench.txt
