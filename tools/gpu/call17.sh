#!/bin/bash
# round 4, GPU call 17: the two frames that cover the same samples in consecutive slots of one XCD (PRC_OPT_CAF_PAIR_FRAMES)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c17; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "workgroup_orders" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for pf in 0 1; do
  timeout 150 python tools/caf_bench.py --shape cfg5 --frames 16 --nref 4 --multi turns --pair-frames $pf >> $O/caf.jsonl 2>>$O/err.txt
  timeout 150 python tools/caf_bench.py --shape cfg5 --frames 16 --nref 1 --pair-frames $pf >> $O/caf.jsonl 2>>$O/err.txt
  timeout 150 python tools/caf_bench.py --shape cfg3 --frames 64 --nref 1 --pair-frames $pf >> $O/caf.jsonl 2>>$O/err.txt
done; done
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print(d["shape"], "nref", d["nref"], "pair", d["pair_frames"], "seg us/surf %.2f"%d["seg_us_per_surface"], ("multi us/frame %.1f" % d["multi_us_per_frame"]) if "multi_us_per_frame" in d else "")
PY
