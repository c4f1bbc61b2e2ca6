#!/bin/bash
mkdir -p gpurun_out/r04_c27; O=gpurun_out/r04_c27
timeout 900 python bench.py --workload prconfig --steps 3 > $O/bench_prconfig.json 2> $O/bench.err; tail -c 2600 $O/bench_prconfig.json; tail -3 $O/bench.err
timeout 300 python -m pytest tests/test_gpu_stream.py -x -q -k "prconfig" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
