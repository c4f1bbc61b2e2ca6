#!/bin/bash
# A/B library builds: recompile some translation units with extra flags (objects kept apart, under build/obj_<name>/) and
# link build/libprcore_<name>.so.      tools/build_variant.sh <name> <unit>="<flags>" [<unit>="<flags>" ...]
#   e.g.  tools/build_variant.sh pk caf_fft=-DFT_PK caf_fft_team=-DFT_PK
# Select it at run time with PRCORE_LIB=$PWD/build/libprcore_<name>.so (tools/ab_bench.sh).  Units not named are compiled
# with the shipped flags.  build/ is git-ignored and travels to the GPU box with the snapshot.
set -e
name="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
args=()
for kv in "$@"; do args+=("X_${kv%%=*}=${kv#*=}"); done
make -s -C "$root/passiveradar_amd/csrc" -j8 OBJDIR="$root/build/obj_$name" LIB="$root/build/libprcore_$name.so" "${args[@]}"
echo "built build/libprcore_$name.so  ($*)"
