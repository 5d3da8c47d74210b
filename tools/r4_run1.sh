#!/bin/bash
# round 4, GPU call 1: the whole GPU suite (new: reference callers, convergence-length PSNR, sharded bench records), the bf16 MFMA
# sustained-peak micro-benchmark, and the floor table of the bf16 inference kernel (ablation builds: time, cycles, clock, busy).
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ubench mfma_bf16_peak"; timeout 120 tools/ubench/mfma_bf16_peak 25 > gpurun_out/r4_mfma_bf16_peak.txt 2>&1; cat gpurun_out/r4_mfma_bf16_peak.txt
echo "== floor table (un-profiled wall)"; timeout 600 python tools/mlp_time.py sinnerf_amd/csrc/libsinnerf_hip.so build/variants/lib_nofrag.so build/variants/lib_noepi.so build/variants/lib_nodma.so build/variants/lib_nobar.so build/variants/lib_nosigma.so build/variants/lib_bare.so build/variants/lib_v3_skip.so 2>&1 | tee gpurun_out/r4_floor_wall.txt
echo "== floor table (PMC)"; bash tools/ab_infer.sh r4floor build/variants/lib_nofrag.so build/variants/lib_noepi.so build/variants/lib_nodma.so build/variants/lib_nobar.so build/variants/lib_bare.so build/variants/lib_v3_skip.so
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -25 gpurun_out/pytest_gpu.log
echo "== bench fp32"; timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench.log | wc -c; tail -1 gpurun_out/bench.log | cut -c1-1500
