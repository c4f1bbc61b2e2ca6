#!/bin/bash
mkdir -p gpurun_out/r04_c37; O=gpurun_out/r04_c37
timeout 400 python tools/ls_chain_bench.py 64 > $O/ls_chain.txt 2>&1; tail -9 $O/ls_chain.txt
