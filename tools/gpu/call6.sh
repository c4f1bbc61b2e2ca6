#!/bin/bash
# round 4, GPU call 6: tile-major slow-time buffer (Doppler loads contiguous): full suite, CAF timings, cfg5 / cfg3 lines
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c6; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt | head -3
for sh in "cfg5 16" "cfg3 64" "cfg2 256" "cfg1 256"; do set -- $sh; timeout 150 python tools/caf_bench.py --shape $1 --frames $2 >> $O/caf.jsonl 2>>$O/err.txt; done
timeout 300 python bench.py --workload cfg5 --no-cpu 2>>$O/err.txt | tail -1 > $O/bench_cfg5.json
timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 2>>$O/err.txt | tail -1 > $O/bench_default.json
python - <<PY
import json
for l in open("$O/caf.jsonl"):
    d=json.loads(l); n,R,F={"cfg5":(1<<23,2048,2048),"cfg3":(5000000,1024,1024),"cfg2":(2400000,256,512),"cfg1":(262144,256,256)}[d["shape"]]
    b=16.0*F*(R+1)*d["frames"]; print(d["shape"], "seg us/surf %.2f"%d["seg_us_per_surface"], "doppler ms %.4f -> %.2f TB/s" % (d["doppler_ms"], b/d["doppler_ms"]/1e9), "exec us/surf %.2f"%d["exec_us_per_surface"])
for f in ("bench_cfg5","bench_default"):
    d=json.load(open("$O/%s.json"%f)); print(f, round(d["value"],1), round(d["ms_per_step"],2), {k:round(v["avg_ms_per_launch"],4) for k,v in d["kernels"].items()}, d.get("hbm_frac_of_peak"))
PY
