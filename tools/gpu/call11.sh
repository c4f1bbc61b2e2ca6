#!/bin/bash
# round 4, GPU call 11: XCD-contiguous segment order + illuminators of a frame in consecutive slots of one XCD
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c11; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -q -k "multi or cfg5 or cfg3 or caf_golden or caf_full or caf_team or caf_edge or caf_vs" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for sh in "cfg5 16 4" "cfg3 64 1" "cfg5 16 1"; do set -- $sh; timeout 150 python tools/caf_bench.py --shape $1 --frames $2 --nref $3 --multi turns >> $O/caf.jsonl 2>>$O/err.txt; done
for i in 1 2; do timeout 300 python bench.py --workload cfg5 --no-cpu 2>>$O/err.txt | tail -1 >> $O/bench_cfg5.jsonl; done
cd /tmp
for g in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$g -o pmc -- python $R/bench.py --workload cfg5 --no-cpu --steps 2 --warmup 1 > $O/pmc_$g.log 2>&1; done
cd $R
python - <<PY
import json, glob, csv, collections
for l in open("$O/caf.jsonl"):
    d=json.loads(l); print(d["shape"], "nref", d["nref"], "seg us/surf %.2f"%d["seg_us_per_surface"], "dop ms %.4f"%d["doppler_ms"], ("multi us/frame %.1f singles %.1f" % (d["multi_us_per_frame"], d["singles_ms"]*1e3/d["frames"])) if "multi_us_per_frame" in d else "")
for l in open("$O/bench_cfg5.jsonl"):
    d=json.loads(l); print("bench cfg5", round(d["value"],1), round(d["hbm_frac_of_peak"],4), {k:round(v["avg_ms_per_launch"],4) for k,v in d["kernels"].items()})
for g in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % g, recursive=True):
        acc=collections.defaultdict(lambda:[0,0.0])
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"][:44]; acc[k][0]+=1; acc[k][1]+=float(row["Counter_Value"])
        for k,(n,s) in acc.items():
            if "caf_fft_team" in k or "doppler" in k: print(g, k, "launches", n, "KB/launch", round(s/n), "-> MB per surface (x2 for FETCH):", round(s/n*1e3*(2 if g=="FETCH_SIZE" else 1)/64/1e6,1))
PY
