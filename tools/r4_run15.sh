#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest bf16x3"; timeout 900 python -m pytest tests/test_bf16x3_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_bf16x3.log 2>&1; echo "exit $?"; tail -3 gpurun_out/pytest_bf16x3.log
cat > /tmp/x3t.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["R"])
from oracle import oracle_np as O
import sinnerf_amd
from sinnerf_amd import rendering
dev = torch.device("cuda:0")
rays = torch.from_numpy(O.lego_rays(400, 400, 0)).to(dev)
z = torch.sort(torch.rand((rays.shape[0], 128), device=dev) * 4 + 2, -1)[0].contiguous()
m = sinnerf_amd.NeRF(use_new_activation=True, compute_dtype="bf16x3")
m.load_state_dict({k: torch.from_numpy(v) for k, v in O.init_params(1, True).items()})
m = m.to(dev).eval()
with torch.no_grad():
    for _ in range(2): rendering._mlp(m, rays, z, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): rendering._mlp(m, rays, z, False)
    e1.record(); torch.cuda.synchronize()
print("%.3f ms" % (e0.elapsed_time(e1) / 3))
PY
export R=$PWD
for lib in "" build/variants/lib_x3_noepi.so build/variants/lib_x3_nodma.so build/variants/lib_x3_nofrag.so build/variants/lib_x3_nobar.so; do
  echo -n "x3 inference fine pass, ${lib:-shipped}: "; SINNERF_HIP_LIB=${lib:+$PWD/$lib} python /tmp/x3t.py 2>&1 | tail -1
done
cd /tmp
for lib in "" build/variants/lib_x3_noepi.so build/variants/lib_x3_nodma.so; do
  SINNERF_HIP_LIB=${lib:+$R/$lib} timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x3abl -o $(basename ${lib:-shipped} .so) --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- python /tmp/x3t.py > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, statistics
for f in sorted(glob.glob("gpurun_out/x3abl/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for r in csv.DictReader(open(f)):
        if "bf16x3" not in r["Kernel_Name"]: continue
        k = r["Dispatch_Id"]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    ks = list(dur); med = lambda c: statistics.median(per[k][c] for k in ks); ms = statistics.median(dur.values()); cyc = med("GRBM_GUI_ACTIVE") / 8
    print("%-28s %7.3f ms %7.2f Mcyc clock %.2f busy %.3f parked %.3f issue-wait %.3f active %.3f lds-wait %.3f" % (f.split("/")[-1][:28], ms, cyc / 1e6, cyc / ms / 1e6,
          med("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc, med("SQ_WAIT_ANY") / med("SQ_WAVE_CYCLES"), med("SQ_WAIT_INST_ANY") / med("SQ_WAVE_CYCLES"),
          med("SQ_ACTIVE_INST_ANY") / med("SQ_WAVE_CYCLES"), med("SQ_WAIT_INST_LDS") / med("SQ_WAVE_CYCLES")))
PY
