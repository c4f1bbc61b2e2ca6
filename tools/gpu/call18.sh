#!/bin/bash
# round 4, GPU call 18: final tree -- full GPU suite, smoke, the config-5 and default lines
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c18; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; grep -E "passed|failed" $O/pytest_all.txt | tail -1
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke
timeout 400 python bench.py --workload cfg5 > $O/bench_cfg5.json 2>$O/err5.txt; python -c "import json; d=json.loads(open('$O/bench_cfg5.json').read().strip().splitlines()[-1]); print('cfg5', round(d['value'],1), round(d['hbm_frac_of_peak'],4))"
timeout 400 python bench.py --no-cpu > $O/bench_default.json 2>$O/err2.txt; python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('cfg2', round(d['value'],1), round(d['hbm_frac_of_peak'],4), d['roofline']['frac'])"
