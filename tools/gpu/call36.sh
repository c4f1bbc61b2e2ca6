#!/bin/bash
# config-3 and config-5 lines on the final tree
mkdir -p gpurun_out/r04_c36; O=gpurun_out/r04_c36
timeout 400 python bench.py --workload cfg5 --no-cpu > $O/bench_cfg5.json 2> $O/cfg5.err; python -c "
import json;d=json.loads(open('$O/bench_cfg5.json').read().strip().splitlines()[-1]);print('cfg5',d['value'],d['roofline']['frac'])"
timeout 600 python bench.py --workload cfg3 --no-cpu > $O/bench_cfg3.json 2> $O/cfg3.err; python -c "
import json;d=json.loads(open('$O/bench_cfg3.json').read().strip().splitlines()[-1]);print('cfg3',d['value'],d['roofline']['frac'])"
