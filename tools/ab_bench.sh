#!/bin/bash
# A/B on one box: PRCORE_LIB selects the library build.  usage: tools/ab_bench.sh "<bench args>" libA.so libB.so
args="$1"; shift
for rep in 1 2 3; do for lib in "$@"; do
  PRCORE_LIB=$PWD/$lib python bench.py --no-cpu $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v['avg_ms_per_launch']*v['launches_per_step'],3) for k,v in d['kernels'].items()})"
done; done
