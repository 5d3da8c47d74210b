mkdir -p gpurun_out
timeout 300 tools/ubench/mfma_bf16_peak 25 2>&1 | tee gpurun_out/ubench_mfma_bf16_peak.txt
