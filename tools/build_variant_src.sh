#!/bin/bash
# Build an ablation variant of ONE kernel source into build/variants/lib_<name>.so (same ABI; load with SINNERF_HIP_LIB=...).
# usage: tools/build_variant_src.sh name source.hip "-DFLAG ..."
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; flags=$3
mkdir -p $R/build/variants
cd $R/sinnerf_amd/csrc
base=${src%.hip}
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w $flags -c $src -o $R/build/variants/${base}_$name.o
objs=$(ls *.o | grep -v "^$base.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$name.so $objs $R/build/variants/${base}_$name.o
echo "built $name ($src $flags)"
