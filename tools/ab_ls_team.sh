# A/B runs of the 4096-point LS chain on one box: bash tools/ab_ls_team.sh  (prints one line per variant)
mkdir -p gpurun_out/lstc5
run() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu $ARGS > gpurun_out/lstc5/$name.json 2>gpurun_out/lstc5/$name.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/lstc5/$name.json").read().strip().splitlines()[-1])
k=j["kernels"]
print("$name", round(j["value"]), "corr", round(k["ls_correlate"]["avg_ms_per_launch"],3), "solve", round(k["ls_solve"]["avg_ms_per_launch"],4), "fused", round(k["ls_fir_subtract"]["avg_ms_per_launch"],3), "frac", round(j["roofline"]["frac"],3))
PY
}
ARGS="" run auto A=1
for v in nofft noload nofftnoload strided nowait; do ARGS="" run $v PRCORE_LIB=$PWD/passiveradar_amd/libprcore_lt_$v.so; done
ARGS="" run auto_b A=1
