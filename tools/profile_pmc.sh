#!/bin/bash
# SQ / LDS counters of one workload's kernels, one rocprofv3 --pmc pass per counter group (never combined with
# trace domains): tools/profile_pmc.sh <tag> <bench args...>       (on the GPU box, through gpurun)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$i -o pmc -- python $R/bench.py --no-cpu "$@" > $out/run_$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(("caf_", "ls_", "nlms", "shift_", "transpose")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in sorted(acc.items()):
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write(f"   {c:28s} {sum(v) / len(v):16.1f}   (launches {len(v)})\n")
print(open(out + "/summary.txt").read()[:6000])
PY
