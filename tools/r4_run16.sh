#!/bin/bash
# A/B of the asm weight DMA (counted LDS waits) + row stores read one step ahead: build/variants/lib_head.so (git 968029c) vs in-tree
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
echo "== pytest (x3, grads, parity)"; timeout 1200 python -m pytest tests/test_bf16x3_gpu.py tests/test_grads_gpu.py tests/test_parity_gpu.py tests/test_classic_heads_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_r16.log 2>&1; echo "exit $?"; tail -3 gpurun_out/pytest_r16.log
cat > /tmp/inf.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["R"])
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import rendering
dev = torch.device("cuda:0")
rays = torch.from_numpy(O.lego_rays(400, 400, 0)).to(dev)
z = torch.sort(torch.rand((rays.shape[0], 128), device=dev) * 4 + 2, -1)[0].contiguous()
for dt in ("fp32", "bf16x3"):
    m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype=dt)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        for _ in range(2): rendering._mlp(m, rays, z, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): rendering._mlp(m, rays, z, False)
        e1.record(); torch.cuda.synchronize()
    print("%s inference fine pass %.3f ms" % (dt, e0.elapsed_time(e1) / 3))
PY
for lib in build/variants/lib_head.so ""; do
  echo "== ${lib:-in-tree}"
  SINNERF_HIP_LIB=${lib:+$PWD/$lib} python /tmp/inf.py 2>&1 | tail -2
  SINNERF_HIP_LIB=${lib:+$PWD/$lib} python tools/x3_step_time.py fp32 bf16x3 2>&1 | tail -2
done
cd /tmp
for lib in build/variants/lib_head.so ""; do
  n=$(basename ${lib:-intree} .so)
  SINNERF_HIP_LIB=${lib:+$R/$lib} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r16 -o $n -- python $R/tools/x3_step_time.py fp32 bf16x3 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/r16/**/*_kernel_stats.csv", recursive=True)):
    print(f.split("/")[-1])
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 1.5: print("   %-60s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
