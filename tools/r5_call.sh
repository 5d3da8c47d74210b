mkdir -p gpurun_out
R=$PWD
{
for rep in 1 2 3 4; do
for lib in "" "$R/build/variants/lib_oldfwd.so"; do
  echo "== lib=${lib:-in-tree (generated forward + generated chain)}"
  SINNERF_HIP_LIB=$lib python tools/x3_step_time.py bf16x3 2>&1 | grep -v amdgpu.ids
done
done
} 2>&1 | tee gpurun_out/ab_x3_step_fwd_choice.txt
