#!/bin/bash
# front end group kernel: where the time goes (ablation builds)
mkdir -p gpurun_out/r04_c22; O=gpurun_out/r04_c22
for v in "" fe_norot fe_nofir fe_neither; do
  if [ -n "$v" ]; then export PRCORE_LIB=$PWD/build/libprcore_$v.so; else unset PRCORE_LIB; fi
  echo "== ${v:-shipped}" >> $O/fe_ablation.txt
  timeout 300 python tools/frontend_bench.py 2>&1 | grep "method 2" >> $O/fe_ablation.txt
done
cat $O/fe_ablation.txt
