#!/bin/bash
# round 4, GPU call 3: the rewritten shared-surveillance kernel (modes turns / shared / pairs; build variants), parity
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c3; mkdir -p $O; cd $R
export TMPDIR=/tmp
cb() { # cb <variant> <shape> <frames> <mode>
  local L=""; [ $1 != default ] && L="PRCORE_LIB=$R/build/libprcore_$1.so"
  env $L timeout 150 python tools/caf_bench.py --shape $2 --frames $3 --nref 4 --multi $4 --tag $1 >> $O/caf.jsonl 2>>$O/caf.err
}
for m in turns shared pairs; do cb default cfg5 16 $m; done
for v in m_scalar m_tw2reg m_nbuf1; do cb $v cfg5 16 shared; done
for m in turns shared pairs; do cb default cfg3 32 $m; done
cb default cfg5 16 shared
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -k "multi" > $O/pytest_multi.txt 2>&1; tail -3 $O/pytest_multi.txt
timeout 300 python -m pytest tests/test_gpu_stream.py -x -q -k "prconfig or bench_cfg5" > $O/pytest_prconfig.txt 2>&1; tail -5 $O/pytest_prconfig.txt
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print("%-10s %s multi=%s seg us/surf %7.2f dop ms %.4f | multi us/frame %8.2f  singles us/frame %8.2f  shared-bytes GB/s %7.1f" % (d["tag"], d["shape"], d["multi"], d["seg_us_per_surface"], d["doppler_ms"], d["multi_us_per_frame"], d["singles_ms"]*1e3/d["frames"], d["multi_GBps_shared_bytes"]))
PY
