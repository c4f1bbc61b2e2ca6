#!/bin/bash
mkdir -p gpurun_out/r04_c32; O=gpurun_out/r04_c32; rm -f $O/fe_variants.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for v in "" fe_general; do
  if [ -n "$v" ]; then export PRCORE_LIB=$PWD/build/libprcore_$v.so; else unset PRCORE_LIB; fi
  echo "== ${v:-shipped}" >> $O/fe_variants.txt
  timeout 300 python tools/frontend_bench.py 2 2>&1 | grep "method 2" >> $O/fe_variants.txt
done
cat $O/fe_variants.txt
