#!/bin/bash
# Build a timing / tuning variant of the fp32 forward kernel (csrc/sn_mlp_fwd_f32g.hip) into build/variants/lib_f32g_<name>.so (same
# ABI; load it with SINNERF_HIP_LIB=...; only ONE instantiation is compiled: the frame render's, or -- with -DSN_F32G_AB_STORE -- the
# training forward's, replacing that translation unit of the library).
# usage: tools/build_variant_f32g.sh name [-DSN_F32G_FD=16 -DSN_F32G_NO_ATOMICS -DSN_F32G_AB_STORE ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/variants
cd $R/sinnerf_amd/csrc
tu=""; skip="^sn_mlp_fwd_f32g.o"
if echo "$@" | grep -q SN_F32G_AB_STORE; then tu="-DSN_F32G_TU_STORE"; skip="^sn_mlp_fwd_f32g_store.o"; fi
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w -DSN_F32G_AB $tu "$@" -c sn_mlp_fwd_f32g.hip -o $R/build/variants/f32g_$name.o
objs=$(ls *.o | grep -v "$skip")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_f32g_$name.so $objs $R/build/variants/f32g_$name.o
echo "built f32g variant $name: $@"
