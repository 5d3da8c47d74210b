mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
for rep in 1 2 3; do
for lib in "" "$R/build/variants/lib_oldstage.so"; do
  echo "== lib=${lib:-in-tree (conflict-free staging)}"
  SINNERF_HIP_LIB=$lib python tools/chain_time.py 2>&1 | grep -v amdgpu.ids
  SINNERF_HIP_LIB=$lib python tools/fwd_train_time.py 2>&1 | grep -v amdgpu.ids
done
done
} 2>&1 | tee gpurun_out/ab_f32_staging.txt
timeout 900 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench exit $?"; tail -c 1500 gpurun_out/bench_default.log
