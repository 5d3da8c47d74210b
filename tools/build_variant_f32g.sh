#!/bin/bash
# Build a timing / tuning variant of the fp32 inference kernel (csrc/sn_mlp_fwd_f32g.hip) into build/variants/lib_f32g_<name>.so (same
# ABI; load it with SINNERF_HIP_LIB=...; only the frame render's instantiation is compiled).
# usage: tools/build_variant_f32g.sh name [-DSN_F32G_FD=16 -DSN_F32G_NO_ATOMICS ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/variants
cd $R/sinnerf_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w -DSN_F32G_AB "$@" -c sn_mlp_fwd_f32g.hip -o $R/build/variants/f32g_$name.o
objs=$(ls *.o | grep -v "^sn_mlp_fwd_f32g.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_f32g_$name.so $objs $R/build/variants/f32g_$name.o
echo "built f32g variant $name: $@"
