#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c16; mkdir -p $O; cd $R
export TMPDIR=/tmp
for fr in 1024 5632; do
  timeout 200 python -X faulthandler bench.py --no-cpu --steps 4 --warmup 2 --gather prc --frames $fr > $O/out_$fr.txt 2> $O/err_$fr.txt; echo "frames $fr rc=$?"; tail -c 400 $O/out_$fr.txt; grep -v amdgpu.ids $O/err_$fr.txt | tail -25
done
