#!/bin/bash
# Scalar-unit load of every kernel of the path (round 5: the front end's row loop turned out to be bound by the CU's ONE
# scalar unit): one rocprofv3 --pmc pass (counters only) per workload, SQ_INSTS_SALU / SQ_INSTS_SMEM / SQ_ACTIVE_INST_SCA
# next to SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU.   bash tools/salu_pmc.sh   (on the GPU box; output under gpurun_out/salu/)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/salu; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { local name=$1; shift
  for grp in "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
    g=$(echo $grp | cut -c1-14 | tr ' ' '_')
    (cd $R && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/${name}_$g -o pmc -- "$@" > $O/${name}_$g.log 2>&1)
  done; }
run cfg2 python bench.py --no-cpu --frames 256 --steps 2 --warmup 1 --no-overlap --traffic none
run cfg3 python bench.py --no-cpu --workload cfg3 --frames 64 --steps 1 --warmup 1 --traffic none
run cfg5 python bench.py --no-cpu --workload cfg5 --steps 2 --warmup 1 --traffic none
run prconfig python bench.py --no-cpu --workload prconfig --frames 64 --steps 1
cd $R; python - <<'PY'
import csv, glob, collections, os
OURS = ("caf_", "doppler_col", "ls_", "levinson", "nlms", "frontend", "cfar")
short = lambda n: n.replace("void ", "").split("(")[0]
for wl in ("cfg2", "cfg3", "cfg5", "prconfig"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/salu/{wl}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith(OURS):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n## {wl}\n\n| kernel | launches | VALU insts | SALU insts | SMEM insts | SALU / VALU | scalar active / VALU active |\n|---|---|---|---|---|---|---|")
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        v = c.get("SQ_INSTS_VALU", 0) or 1
        print(f"| `{k[:60]}` | {len(acc[k].get('SQ_INSTS_VALU', []))} | {v:.3g} | {c.get('SQ_INSTS_SALU', 0):.3g} | {c.get('SQ_INSTS_SMEM', 0):.3g} | "
              f"{c.get('SQ_INSTS_SALU', 0) / v:.2f} | {c.get('SQ_ACTIVE_INST_SCA', 0) / (c.get('SQ_ACTIVE_INST_VALU', 0) or 1):.2f} |")
PY
