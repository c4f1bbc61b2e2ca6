#!/bin/bash
mkdir -p gpurun_out/r04_c33; O=gpurun_out/r04_c33; rm -f $O/fe_variants.txt
for v in "" fe_pipe; do
  if [ -n "$v" ]; then export PRCORE_LIB=$PWD/build/libprcore_$v.so; else unset PRCORE_LIB; fi
  echo "== ${v:-shipped}" >> $O/fe_variants.txt
  timeout 300 python tools/frontend_bench.py 2>&1 | grep "method 2\|max" >> $O/fe_variants.txt
  timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end" 2>&1 | tail -1 >> $O/fe_variants.txt
done
cat $O/fe_variants.txt
