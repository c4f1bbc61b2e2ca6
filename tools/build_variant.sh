#!/bin/bash
# A/B library builds: recompile ONE translation unit with extra flags and link it with the default objects.
#   tools/build_variant.sh <name> <file.hip> "<flags>"   ->  passiveradar_amd/libprcore_<name>.so
# (select it at run time with PRCORE_LIB=$PWD/passiveradar_amd/libprcore_<name>.so; see tools/ab_bench.sh)
set -e
name="$1"; src="$2"; flags="$3"
cd "$(dirname "$0")/../passiveradar_amd/csrc"
make -s -j8
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-fast-math -ffp-contract=on -fno-slp-vectorize"
obj="/tmp/prc_variant_${name}_$(basename "$src" .hip).o"
$HIPCC $CXXFLAGS $flags -c "$src" -o "$obj"
others=$(ls *.o | grep -v "^$(basename "$src" .hip).o$")
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "../libprcore_${name}.so" $obj $others -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib
echo "built passiveradar_amd/libprcore_${name}.so  ($src: $flags)"
