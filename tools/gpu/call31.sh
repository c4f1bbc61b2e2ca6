#!/bin/bash
mkdir -p gpurun_out/r04_c31; O=gpurun_out/r04_c31
timeout 900 python bench.py --workload prconfig --steps 3 > $O/bench_prconfig.json 2> $O/bench.err; tail -c 4200 $O/bench_prconfig.json; tail -5 $O/bench.err
