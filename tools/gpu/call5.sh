#!/bin/bash
# round 4, GPU call 5: NLMS wavefronts per SIMD; Doppler kernel with contiguous tiles (upper bound of a tile-major layout);
# prconfig workload at the published size; multi AUTO test
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_c5; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python tools/nlms_wg_probe.py > $O/nlms_wg.txt 2>&1; cat $O/nlms_wg.txt | grep -v amdgpu.ids
timeout 120 python tools/caf_bench.py --shape dop2048x8 --frames 4112 --tag contiguous >> $O/dop.jsonl 2>>$O/err.txt
timeout 120 python tools/caf_bench.py --shape cfg5 --frames 16 --tag strided >> $O/dop.jsonl 2>>$O/err.txt
timeout 120 python tools/caf_bench.py --shape dop512x16 --frames 4112 --tag contiguous >> $O/dop.jsonl 2>>$O/err.txt
timeout 120 python tools/caf_bench.py --shape cfg2 --frames 256 --tag strided >> $O/dop.jsonl 2>>$O/err.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "multi_auto or nlms" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python bench.py --workload prconfig --steps 3 > $O/bench_prconfig.json 2> $O/bench_prconfig.err; tail -c 2500 $O/bench_prconfig.json; tail -3 $O/bench_prconfig.err
python - <<PY
import json
for l in open("$O/dop.jsonl"):
    d=json.loads(l); n,R,F={"cfg5":(1<<23,2048,2048),"cfg2":(2400000,256,512),"dop2048x8":(131072,7,2048),"dop512x16":(32768,15,512)}[d["shape"]]
    b=16.0*F*(R+1)*d["frames"]; print(d["shape"], d["tag"], "doppler ms %.4f"%d["doppler_ms"], "-> %.2f TB/s" % (b/d["doppler_ms"]/1e9))
PY
