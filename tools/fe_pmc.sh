R=$PWD; O=$R/gpurun_out/r05l; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && FE_BALANCES=0 timeout 150 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o pmc -- python tools/frontend_bench.py > $O/g$i.log 2>&1; echo "group $i rc $?")
done
cd $R; python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r05l/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "frontend_group_kernel" in k:
            key="x2" if "13, 2>" in k or "ILi13ELi2" in k else "x1"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in acc:
    print(key, {n: round(sum(v)/len(v)) for n,v in sorted(acc[key].items())}, "launches", {n:len(v) for n,v in acc[key].items()}.get("SQ_WAVE_CYCLES"))
PY
