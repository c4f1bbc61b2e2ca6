#!/bin/bash
# RCCL self-peer link check on one GPU
mkdir -p gpurun_out/r04_c20; O=gpurun_out/r04_c20
timeout 300 python -m pytest tests/test_gpu_stream.py -x -q -k "loopback or single_rank or headline" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
