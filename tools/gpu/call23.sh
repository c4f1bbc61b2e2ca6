#!/bin/bash
# front end group kernel: variants + counters
mkdir -p gpurun_out/r04_c23; O=gpurun_out/r04_c23
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "front_end" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in "" fe_chunk16 fe_waves4 fe_waves4c16 fe_waves16; do
  if [ -n "$v" ]; then export PRCORE_LIB=$PWD/build/libprcore_$v.so; else unset PRCORE_LIB; fi
  echo "== ${v:-shipped}" >> $O/fe_variants.txt
  timeout 300 python tools/frontend_bench.py 2 2>&1 | grep "method 2" >> $O/fe_variants.txt
done
unset PRCORE_LIB
cat $O/fe_variants.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_g$i -o pmc -- python tools/frontend_bench.py 2 > $GRAFT_REPO_ROOT/$O/pmc_g$i.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04_c23/pmc_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frontend" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
P
