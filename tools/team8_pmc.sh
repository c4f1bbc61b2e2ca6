#!/bin/bash
# Counters of the 4096-point CAF segment kernel in its two team sizes (PRC_OPT_CAF_TEAM8 0 / 1): one rocprofv3 --pmc pass per
# counter group (counters only), config-5 shape, 16 frames x 4 illuminators in one launch.   tools/team8_pmc.sh <outdir>
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/${1:-gpurun_out/team8_pmc}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for t8 in 0 1; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    (cd $R && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/t$t8.g$i -o pmc -- python tools/caf_bench.py --shape cfg5 --frames 16 --nref 4 --multi turns --team8 $t8 --reps 2 > $O/t$t8.g$i.log 2>&1)
  done
done
cd $R; python - "$O" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
print("| segment kernel | launches | waves / launch | VALU insts / wave | LDS insts / wave | VALU active / wave-cycles | WAIT_ANY | WAIT_INST_ANY | WAIT_INST_LDS | LDS active / wave-cycles | LDS conflict / LDS active | 2 x FETCH + WRITE MB / surface |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for t8, name in ((0, "caf_fft_team_kernel"), (1, "caf_fft_team8_kernel")):
    acc = collections.defaultdict(list)
    big = 0
    rows = []
    for f in glob.glob(f"{O}/t{t8}.g*/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
    for r in rows: big = max(big, int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) == big: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    if not c: continue
    wc, waves = c.get("SQ_WAVE_CYCLES", 1), c.get("SQ_WAVES", 1)
    pct = lambda k: f"{100 * c.get(k, 0) / wc:.0f} %"
    mb = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1e3 / 64 / 1e6
    print(f"| `{name}` | {len(acc.get('SQ_WAVES', []))} | {waves:.0f} | {c.get('SQ_INSTS_VALU', 0) / waves:.0f} | {c.get('SQ_INSTS_LDS', 0) / waves:.0f} | {pct('SQ_ACTIVE_INST_VALU')} | "
          f"{pct('SQ_WAIT_ANY')} | {pct('SQ_WAIT_INST_ANY')} | {pct('SQ_WAIT_INST_LDS')} | {pct('SQ_ACTIVE_INST_LDS')} | "
          f"{100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):.0f} % | {mb:.1f} |")
PY
