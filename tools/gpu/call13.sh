#!/bin/bash
# round 4, GPU call 13: Doppler tiles XCD-contiguous (shipped) vs launch order
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c13; mkdir -p $O; cd $R
export TMPDIR=/tmp
for rep in 1 2; do for v in default dop_rr; do
  L=""; [ $v != default ] && L="PRCORE_LIB=$R/build/libprcore_$v.so"
  for sh in "cfg5 16" "cfg3 64" "cfg2 256"; do set -- $sh; env $L timeout 150 python tools/caf_bench.py --shape $1 --frames $2 --tag $v >> $O/caf.jsonl 2>>$O/err.txt; done
done; done
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print(d["tag"], d["shape"], "dop ms %.4f"%d["doppler_ms"])
PY
