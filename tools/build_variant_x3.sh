#!/bin/bash
# Build a tuning variant of the generated bf16x3 training kernels (forward trunk + backward chain) into build/variants/lib_<name>.so
# (same ABI; load it with SINNERF_HIP_LIB=...).  usage: tools/build_variant_x3.sh name knob=value ...
# (knobs: tools/gen_x3_trunk.py / tools/gen_x3_chain.py; a knob prefixed "t." / "c." goes to the trunk / the chain only)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
TK=""; CK=""
for kv in "$@"; do
  case $kv in
    t.*) TK="$TK ${kv#t.}";;
    c.*) CK="$CK ${kv#c.}";;
    *) TK="$TK $kv"; CK="$CK $kv";;
  esac
done
mkdir -p $R/build/variants
python3 $R/tools/gen_x3_trunk.py $R/build/variants/x3_trunk_$name.inc $TK > $R/build/variants/$name.log
python3 $R/tools/gen_x3_chain.py $R/build/variants/x3_chain_$name.inc $CK >> $R/build/variants/$name.log
cd $R/sinnerf_amd/csrc
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -w"
hipcc $F -DSN_X3_TRUNK_INC="\"$R/build/variants/x3_trunk_$name.inc\"" -c sn_mlp_fwd_bf16x3_t.hip -o $R/build/variants/x3f_$name.o &
hipcc $F -DSN_X3_CHAIN_INC="\"$R/build/variants/x3_chain_$name.inc\"" -c sn_mlp_bwd_bf16x3_t.hip -o $R/build/variants/x3c_$name.o &
wait
objs=$(ls *.o | grep -v "^sn_mlp_fwd_bf16x3_t.o$" | grep -v "^sn_mlp_bwd_bf16x3_t.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/lib_$name.so $objs $R/build/variants/x3f_$name.o $R/build/variants/x3c_$name.o
rm -f $R/build/variants/x3f_$name.o $R/build/variants/x3c_$name.o
echo "built $name: $(tr '\n' ' ' < $R/build/variants/$name.log)"
