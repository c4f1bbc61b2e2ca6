#!/bin/bash
# validation of the tree with the group front end, separable CFAR, loopback: full GPU suite, fuzz, default line
mkdir -p gpurun_out/r04_c28; O=gpurun_out/r04_c28
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
timeout 400 python tests/fuzz_parity.py 61 150 1 $O/r04_fuzz_parity.md > $O/fuzz1.txt 2>&1; tail -2 $O/fuzz1.txt
timeout 300 python tests/fuzz_parity.py 63 90 4 $O/r04_fuzz_parity_4threads.md > $O/fuzz3.txt 2>&1; tail -2 $O/fuzz3.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 500 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
