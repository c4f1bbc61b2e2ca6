#!/bin/bash
# front end group kernel after the partial-sum layout fix: ablations, chunk sizes, counters
mkdir -p gpurun_out/r04_c25; O=gpurun_out/r04_c25
for v in "" fe_norot fe_nofir fe_neither fe_chunk4 fe_chunk16; do
  if [ -n "$v" ]; then export PRCORE_LIB=$PWD/build/libprcore_$v.so; else unset PRCORE_LIB; fi
  echo "== ${v:-shipped}" >> $O/fe_variants.txt
  timeout 300 python tools/frontend_bench.py 2 2>&1 | grep "method 2" >> $O/fe_variants.txt
done
unset PRCORE_LIB
cat $O/fe_variants.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_CYCLES"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_g$i -o pmc -- python tools/frontend_bench.py 2 > $GRAFT_REPO_ROOT/$O/pmc_g$i.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04_c25/pmc_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frontend" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
P
