#!/bin/bash
# pipelined published workload + the piecewise front end
mkdir -p gpurun_out/r04_c19; O=gpurun_out/r04_c19
timeout 300 python -m pytest tests/test_gpu_stream.py -x -q -k "prconfig or front_end" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python bench.py --workload prconfig --steps 3 --frames 240 --no-cpu > $O/bench_prconfig_240.json 2> $O/bench_240.err; tail -c 1800 $O/bench_prconfig_240.json; tail -3 $O/bench_240.err
timeout 900 python bench.py --workload prconfig --steps 3 > $O/bench_prconfig.json 2> $O/bench.err; tail -c 2500 $O/bench_prconfig.json; tail -3 $O/bench.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; tail -2 $O/bench_default.err
