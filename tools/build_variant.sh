#!/bin/bash
# A/B library builds: recompile SOME translation units with extra flags (objects under build/obj_<name>/) and link them
# with the objects of a base build for every other unit -> build/libprcore_<name>.so
#     tools/build_variant.sh <name> <unit>="<flags>" [<unit>="<flags>" ...]
#     BASE=pk tools/build_variant.sh t_reuse caf_fft_team="-DFT_PK"      (base: build/obj_pk/ instead of the shipped objects)
# Select a build at run time with PRCORE_LIB=$PWD/build/libprcore_<name>.so (tools/ab_bench.sh).  build/ is git-ignored and
# travels to the GPU box with the snapshot.
set -e
name="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/passiveradar_amd/csrc"
make -s -C "$src" -j8                                   # the shipped objects (base of bases)
if [ -n "$BASE" ]; then base="$root/build/obj_$BASE"; else base="$src"; fi
out="$root/build/obj_$name"; mkdir -p "$out"
units=()
args=()
for kv in "$@"; do units+=("${kv%%=*}"); args+=("X_${kv%%=*}=${kv#*=}"); done
targets=(); for u in "${units[@]}"; do targets+=("$out/$u.o"); done
make -s -C "$src" -j8 OBJDIR="$out" "${args[@]}" "${targets[@]}"
objs=()
for f in "$src"/*.hip; do
  u=$(basename "$f" .hip)
  if [ -f "$out/$u.o" ] && [[ " ${units[*]} " == *" $u "* ]]; then objs+=("$out/$u.o")
  elif [ -f "$base/$u.o" ]; then objs+=("$base/$u.o")
  else objs+=("$src/$u.o"); fi
done
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -shared -fPIC -o "$root/build/libprcore_$name.so" "${objs[@]}" -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib
echo "built build/libprcore_$name.so  ($*${BASE:+; base $BASE})"
