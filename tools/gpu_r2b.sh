#!/bin/bash
# round-2 call B: bf16 parity after the angle-doubling embedding + timing of v3 variants + PMC of the bf16 fine-pass launch
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest bf16"; timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_bf16_configs_gpu.py tests/test_grads_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "bf16" > gpurun_out/pytest_bf16.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_bf16.log
echo "== mlp_time"; timeout 600 python tools/mlp_time.py sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_nodbl.so build/variants/lib_skip.so build/variants/lib_skipnodbl.so "$@" 2>&1 | tee gpurun_out/mlp_time.log
cd /tmp
for v in base:$R/sinnerf_amd/csrc/libsinnerf_hip.so skip:$R/build/variants/lib_skip.so; do
  name=${v%%:*}; lib=${v#*:}
  SINNERF_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcb -o ${name}_mix --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES -- python $R/tools/mlp_time.py --child 0 > $R/gpurun_out/pmcb_${name}_mix.log 2>&1
  SINNERF_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcb -o ${name}_sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python $R/tools/mlp_time.py --child 0 > $R/gpurun_out/pmcb_${name}_sq.log 2>&1
done
ls $R/gpurun_out/pmcb
