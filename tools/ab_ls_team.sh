# A/B runs of the 4096-point LS chain on one box: bash tools/ab_ls_team.sh name1 name2 ...  (libprcore_<name>.so built by
# tools/build_variant.sh; "default" = the shipped library).  Prints one line per variant.
mkdir -p gpurun_out/ab_ls
for name in "$@"; do
  lib=""; [ "$name" != default ] && lib="PRCORE_LIB=$PWD/passiveradar_amd/libprcore_$name.so"
  env $lib timeout 120 python bench.py --no-cpu > gpurun_out/ab_ls/$name.json 2>gpurun_out/ab_ls/$name.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_ls/$name.json").read().strip().splitlines()[-1])
k=j["kernels"]
print("$name", round(j["value"]), "corr", round(k["ls_correlate"]["avg_ms_per_launch"],3), "solve", round(k["ls_solve"]["avg_ms_per_launch"],4), "fused", round(k["ls_fir_subtract"]["avg_ms_per_launch"],3), "frac", round(j["roofline"]["frac"],3))
PY
done
