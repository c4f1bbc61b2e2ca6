#!/bin/bash
# round 4, GPU call 2: where the CAF kernels' time goes with packed butterflies (ablations), early surveillance loads in
# the wavefront kernel, the window-half stash of the team kernel (time + FETCH_SIZE), occupancy variants
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c2; mkdir -p $O; cd $R
export TMPDIR=/tmp
cb() { # cb <variant> <shape> <frames>
  local L=""; [ $1 != default ] && L="PRCORE_LIB=$R/build/libprcore_$1.so"
  env $L timeout 120 python tools/caf_bench.py --shape $2 --frames $3 --tag $1 >> $O/caf.jsonl 2>>$O/caf.err
}
for v in pk2 t_noload t_nobar t_nox t_nox_noload t_fact t_reuse t_reuse_fact t_2w pk2 t_reuse; do cb $v cfg5 16; done
for v in pk2 t_fact t_2w t_noload; do cb $v cfg3 64; done
for v in pk2 w_latev w_noload pk2 w_latev; do cb $v cfg2 256; done
# parity of the stash variants on the CAF tests
PRCORE_LIB=$R/build/libprcore_t_reuse.so timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "caf" > $O/pytest_reuse.txt 2>&1
tail -3 $O/pytest_reuse.txt
# traffic of the segment kernel with and without the stash
cd /tmp
for v in pk2 t_reuse; do
  PRCORE_LIB=$R/build/libprcore_$v.so timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$v -o pmc -- python $R/tools/caf_bench.py --shape cfg5 --frames 16 --reps 2 > $O/pmc_$v.log 2>&1
done
cd $R
python - <<PY
import json, glob, csv, collections
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print("%-14s %s seg us/surf %7.2f  dop ms %.4f  exec us/surf %7.2f" % (d["tag"], d["shape"], d["seg_us_per_surface"], d["doppler_ms"], d["exec_us_per_surface"]))
for v in ("pk2","t_reuse"):
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % v, recursive=True):
        acc=collections.defaultdict(lambda:[0,0.0])
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"][:40]; acc[k][0]+=1; acc[k][1]+=float(row["Counter_Value"])
        for k,(n,s) in acc.items():
            if "caf_fft_team" in k or "doppler" in k: print(v, k, "launches", n, "FETCH_SIZE/launch (KB units x1024 B?)", s/n)
PY
