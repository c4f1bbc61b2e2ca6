#!/bin/bash
mkdir -p gpurun_out/r04_c41; O=gpurun_out/r04_c41
timeout 110 python tests/fuzz_parity.py 81 70 1 $O/r04_fuzz_parity_final_build.md > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt | cut -c1-300
