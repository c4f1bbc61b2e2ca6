#!/bin/bash
# round 4, GPU call 12: workgroup order of the 4096-point segment kernel -- XCD-contiguous vs round-robin, one and four channels
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c12; mkdir -p $O; cd $R
export TMPDIR=/tmp
for rep in 1 2; do for x in 0 1; do
  timeout 150 python tools/caf_bench.py --shape cfg5 --frames 16 --nref 4 --multi turns --xcd-contig $x >> $O/caf.jsonl 2>>$O/err.txt
  timeout 150 python tools/caf_bench.py --shape cfg5 --frames 16 --nref 1 --xcd-contig $x >> $O/caf.jsonl 2>>$O/err.txt
  timeout 150 python tools/caf_bench.py --shape cfg3 --frames 64 --nref 1 --xcd-contig $x >> $O/caf.jsonl 2>>$O/err.txt
  timeout 150 python tools/caf_bench.py --shape cfg3 --frames 32 --nref 4 --multi turns --xcd-contig $x >> $O/caf.jsonl 2>>$O/err.txt
done; done
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print(d["shape"], "nref", d["nref"], "contig", d["xcd_contig"], "seg us/surf %.2f"%d["seg_us_per_surface"], ("multi us/frame %.1f singles %.1f" % (d["multi_us_per_frame"], d["singles_ms"]*1e3/d["frames"])) if "multi_us_per_frame" in d else "")
PY
