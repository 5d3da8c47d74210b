#!/bin/bash
# Build an ablation / tuning variant of the hand-scheduled backward CHAIN into build/variants/lib_<name>.so
# usage: tools/build_variant_c.sh name [--cflags "-D..."] knob=value ...   (knobs: tools/gen_bf16_chain.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
EXTRA=""
if [ "$1" = "--cflags" ]; then EXTRA="$2"; shift; shift; fi
mkdir -p $R/build/variants
python3 $R/tools/gen_bf16_chain.py $R/build/variants/chain_t_$name.inc "$@" > $R/build/variants/$name.log
cd $R/sinnerf_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w -DSN_CHAIN_T_INC="\"$R/build/variants/chain_t_$name.inc\"" $EXTRA \
  -c sn_mlp_bwd_bf16_t.hip -o $R/build/variants/c_$name.o
objs=$(ls *.o | grep -v "^sn_mlp_bwd_bf16_t.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$name.so $objs $R/build/variants/c_$name.o
echo "built $name: $(cat $R/build/variants/$name.log)"
