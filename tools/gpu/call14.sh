#!/bin/bash
# round 4, GPU call 14: the final build -- full GPU suite, smoke, fuzz (AUTO, LS method 4, four caller threads)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c14; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt | head -2
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python tests/fuzz_parity.py 51 170 1 $O/r04_fuzz_parity.md > $O/fuzz1.txt 2>&1; tail -2 $O/fuzz1.txt | cut -c1-300
PR_FUZZ_LS_METHOD=4 timeout 400 python tests/fuzz_parity.py 52 170 1 $O/r04_fuzz_parity_ls4.md > $O/fuzz2.txt 2>&1; tail -2 $O/fuzz2.txt | cut -c1-300
timeout 300 python tests/fuzz_parity.py 53 100 4 $O/r04_fuzz_parity_4threads.md > $O/fuzz3.txt 2>&1; tail -2 $O/fuzz3.txt | cut -c1-300
