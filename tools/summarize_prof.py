#!/usr/bin/env python3
"""Summarise gpurun_out/prof/* (rocprofv3 CSV outputs of tools/gpu_round.sh) into profiles/<tag>_*.
usage: python tools/summarize_prof.py r01_run2"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/stats_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
out = {"command": "rocprofv3 --kernel-trace [--stats | --pmc ...] --output-format csv -- python bench.py --steps N --warmup W --no-cpu-baseline",
       "note": "PMC passes are separate runs (pmc1: SQ/GRBM, pmc2: FETCH_SIZE, pmc3: WRITE_SIZE); values per dispatch of the fused MLP kernel"}
per = collections.defaultdict(dict)
for f in ("pmc1", "pmc2", "pmc3"):
    p = f"{src}/{f}_counter_collection.csv"
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        if "mlp_fwd" not in r["Kernel_Name"]:
            continue
        key = (f, int(r["Dispatch_Id"]))
        per[key]["grid"] = int(r["Grid_Size"])
        per[key]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        per[key]["vgpr"] = int(r["VGPR_Count"]); per[key]["agpr"] = int(r["Accum_VGPR_Count"]); per[key]["lds"] = int(r["LDS_Block_Size"])
        per[key][r["Counter_Name"]] = float(r["Counter_Value"])
disp = []
for (f, d), v in sorted(per.items()):
    v = dict(v, pass_=f, dispatch=d)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles are summed over 1024 SIMDs
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        v["clock_ghz"] = cyc / (v["dur_ms"] * 1e6)
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
        v["wave_parked_frac(SQ_WAIT_ANY/SQ_WAVE_CYCLES)"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    disp.append(v)
out["dispatches"] = disp
# HBM traffic of the dominant launch (largest grid): FETCH_SIZE / WRITE_SIZE are in KiB.  The gfx950 2x correction of
# MI355X_MICROARCH.md applies to wide (16 B/lane) coalesced reads; this kernel's reads are 4 B/lane z_vals + L2-resident
# weights, so the raw FETCH_SIZE is reported and the corrected value given alongside.
# the kernels are persistent (grid = #CUs): the fine pass of the frame render is the longest dispatch of each pass
POINTS = int(sys.argv[2]) if len(sys.argv) > 2 else 160000 * 128
f = sorted([v for v in disp if "FETCH_SIZE" in v], key=lambda v: -v["dur_ms"])
w = sorted([v for v in disp if "WRITE_SIZE" in v], key=lambda v: -v["dur_ms"])
m = sorted([v for v in disp if "mfma_busy_frac" in v], key=lambda v: -v["dur_ms"])
if m:
    out["fine_pass_sq"] = {k: m[0][k] for k in ("dur_ms", "clock_ghz", "mfma_busy_frac", "wave_parked_frac(SQ_WAIT_ANY/SQ_WAVE_CYCLES)", "vgpr", "agpr")}
if f and w:
    fetch, write = f[0]["FETCH_SIZE"] * 1024, w[0]["WRITE_SIZE"] * 1024
    out["traffic"] = {"kernel": "mlp_fwd_f32g_kernel fine pass", "points": POINTS, "fetch_bytes": fetch,
                      "fetch_bytes_2x_corrected": 2 * fetch, "write_bytes": write, "hbm_bytes": 2 * fetch + write,
                      "algorithmic_bytes": POINTS * 20}
# dynamic instruction mix of the fused MLP kernels (wave-level instruction counts summed over the launch)
mix = {}
for dt in ("fp32", "bf16"):
    p = f"{src}/mix_{dt}_counter_collection.csv"
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(p)):
        if "mlp_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    if agg.get("SQ_INSTS_MFMA"):
        other = agg["SQ_INSTS_VALU"] - agg["SQ_INSTS_MFMA"] + agg["SQ_INSTS_SALU"] + agg["SQ_INSTS_LDS"]
        mix[dt] = dict(agg, non_mfma_valu_per_mfma=(agg["SQ_INSTS_VALU"] - agg["SQ_INSTS_MFMA"]) / agg["SQ_INSTS_MFMA"],
                       salu_per_mfma=agg["SQ_INSTS_SALU"] / agg["SQ_INSTS_MFMA"], lds_per_mfma=agg["SQ_INSTS_LDS"] / agg["SQ_INSTS_MFMA"],
                       other_per_mfma_excl_vmem=other / agg["SQ_INSTS_MFMA"])
if mix:
    out["instruction_mix"] = mix
p = f"{src}/pmc1_bf16_counter_collection.csv"
if os.path.exists(p):
    b = collections.defaultdict(dict)
    for r in csv.DictReader(open(p)):
        if "mlp_fwd" in r["Kernel_Name"]:
            k = int(r["Dispatch_Id"])
            b[k]["dur_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            b[k][r["Counter_Name"]] = float(r["Counter_Value"])
    if b:
        v = max(b.values(), key=lambda x: x["dur_ms"])
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        out["fine_pass_sq_bf16"] = {"dur_ms": v["dur_ms"], "clock_ghz": cyc / (v["dur_ms"] * 1e6),
                                    "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc}
json.dump(out, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
if "traffic" in out:
    import subprocess
    head = subprocess.run(["git", "describe", "--always", "--dirty"], capture_output=True, text=True).stdout.strip() or "unknown"
    sys.path.insert(0, os.getcwd())
    # the hash of the kernel sources the profile was measured ON: written by tools/gpu_round.sh on the measuring box (never recomputed here)
    measured_sha = open("gpurun_out/kernel_sources_sha.txt").read().strip() if os.path.exists("gpurun_out/kernel_sources_sha.txt") else "unstamped"
    json.dump(dict(out["traffic"], source=f"profiles/{tag}_pmc.json", git_head=head, kernel_sources_sha=measured_sha),
              open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "dispatches"}, indent=1)[:5000])
