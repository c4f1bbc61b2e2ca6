mkdir -p gpurun_out/lstc2
run() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu --steps 3 --warmup 1 $ARGS > gpurun_out/lstc2/$name.json 2>gpurun_out/lstc2/$name.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/lstc2/$name.json").read().strip().splitlines()[-1])
k=j["kernels"]
print("$name", round(j["value"]), "corr", round(k["ls_correlate"]["avg_ms_per_launch"],3), "solve", round(k["ls_solve"]["avg_ms_per_launch"],4), "fused", round(k["ls_fir_subtract"]["avg_ms_per_launch"],3))
PY
}
ARGS="--ls-method 0" run m0 A=1
for p in 8 16 32 64; do ARGS="--ls-method 4" run m4_p$p PRC_LS_TEAM_PIECES=$p; done
ARGS="--ls-method 4" run m4_tw2reg_p16 PRC_LS_TEAM_PIECES=16 PRCORE_LIB=$PWD/passiveradar_amd/libprcore_tw2reg.so
ARGS="--ls-method 0" run m0b A=1
