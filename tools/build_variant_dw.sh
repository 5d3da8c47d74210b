#!/bin/bash
# Build a variant of the weight-gradient host plan / kernels into build/variants/lib_<name>.so.  usage: tools/build_variant_dw.sh name "-D..."
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/variants
cd $R/sinnerf_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w "$@" -c sn_dw.hip -o $R/build/variants/dw_$name.o
objs=$(ls *.o | grep -v "^sn_dw.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$name.so $objs $R/build/variants/dw_$name.o
echo "built $name"
