#!/bin/bash
# dynamic instruction mix of the fused MLP kernels (one PMC pass per dtype)
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp; cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES"
for dt in fp32 bf16; do
  timeout 300 $P --pmc $C -o mix_$dt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --dtype $dt > $R/gpurun_out/mix_$dt.log 2>&1; echo "mix $dt exit $?"
done
C2="SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
for dt in fp32 bf16; do
  timeout 300 $P --pmc $C2 -o mix2_$dt -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --dtype $dt > $R/gpurun_out/mix2_$dt.log 2>&1; echo "mix2 $dt exit $?"
done
ls $R/gpurun_out/prof | grep mix
