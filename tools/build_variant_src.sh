#!/bin/bash
# Build a variant of ONE kernel source into build/variants/lib_<name>.so.  usage: tools/build_variant_src.sh name source.hip "-D..."
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift; shift
mkdir -p $R/build/variants
cd $R/sinnerf_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w "$@" -c $src -o $R/build/variants/src_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$name.so $objs $R/build/variants/src_$name.o
echo "built $name"
