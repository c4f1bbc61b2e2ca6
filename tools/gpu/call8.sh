#!/bin/bash
# round 4, GPU call 8: time-domain LS kernels with double accumulation; fuzz of the final build (AUTO and LS method 4)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c8; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -q -k "config3_tap_count or independent_pass or ls_direct or toeplitz or ls_multiple" > $O/pytest_ls.txt 2>&1; tail -4 $O/pytest_ls.txt | head -3
timeout 400 python tests/fuzz_parity.py 41 170 1 $O/r04_fuzz_parity.md > $O/fuzz1.txt 2>&1; tail -2 $O/fuzz1.txt
PR_FUZZ_LS_METHOD=4 timeout 400 python tests/fuzz_parity.py 42 170 1 $O/r04_fuzz_parity_ls4.md > $O/fuzz2.txt 2>&1; tail -2 $O/fuzz2.txt
timeout 300 python tests/fuzz_parity.py 43 100 4 $O/r04_fuzz_parity_4threads.md > $O/fuzz3.txt 2>&1; tail -2 $O/fuzz3.txt
