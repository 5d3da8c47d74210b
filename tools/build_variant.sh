#!/bin/bash
# Build an ablation / tuning variant of the hand-scheduled bf16 trunk into build/variants/lib_<name>.so (same ABI; load it
# with SINNERF_HIP_LIB=...).  usage: tools/build_variant.sh name knob=value ...      (knobs: tools/gen_bf16_trunk.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
EXTRA=""
if [ "$1" = "--cflags" ]; then EXTRA="$2"; shift; shift; fi
mkdir -p $R/build/variants
python3 $R/tools/gen_bf16_trunk.py $R/build/variants/trunk_$name.inc "$@" > $R/build/variants/$name.log
cd $R/sinnerf_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w -DSN_TRUNK_INC="\"$R/build/variants/trunk_$name.inc\"" $EXTRA \
  -c sn_mlp_fwd_bf16_v3.hip -o $R/build/variants/v3_$name.o
objs=$(ls *.o | grep -v sn_mlp_fwd_bf16_v3.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$name.so $objs $R/build/variants/v3_$name.o
echo "built $name: $(cat $R/build/variants/$name.log)"
