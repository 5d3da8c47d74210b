#!/usr/bin/env python3
"""Copy the summaries of one tools/gpu_round.sh run from gpurun_out/ into profiles/<tag>_* (tracked).
usage: python tools/save_round.py r02_run5"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
G, P = "gpurun_out", "profiles"
subprocess.run([sys.executable, "tools/summarize_prof.py", tag], check=True, stdout=subprocess.DEVNULL)
for src, dst in (("prof/stats_bf16_kernel_stats.csv", "bf16_kernel_stats.csv"), ("prof/stats_train_kernel_stats.csv", "train_kernel_stats.csv"),
                 ("train_bench.json", "train_bench.json"), ("train_pmc_cycles.json", "train_pmc_cycles.json"),
                 ("prof/stats_x3_kernel_stats.csv", "infer_launch_kernel_stats.csv"), ("x3_infer_pmc.txt", "infer_launch_pmc.txt"), ("x3_infer.log", "infer_launch.txt")):
    if os.path.exists(f"{G}/{src}"):
        shutil.copy(f"{G}/{src}", f"{P}/{tag}_{dst}")
for log, dst in (("bench.log", "bench_fp32.json"), ("bench_bf16.log", "bench_bf16.json")):
    lines = [l for l in open(f"{G}/{log}") if l.startswith("{")]
    json.dump(json.loads(lines[-1]), open(f"{P}/{tag}_{dst}", "w"), indent=1)

# HBM bytes of the training kernels: FETCH_SIZE / WRITE_SIZE (KiB) of the fine-pass dispatch (the longest of each kernel)
per = collections.defaultdict(lambda: collections.defaultdict(dict))
for cnt, f in (("FETCH_SIZE", "pmc_train_fetch"), ("WRITE_SIZE", "pmc_train_write")):
    p = f"{G}/prof/{f}_counter_collection.csv"
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] != cnt:
            continue
        name = r["Kernel_Name"][:72]
        d = per[name][int(r["Dispatch_Id"])]
        d["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        d[cnt] = d.get(cnt, 0.0) + float(r["Counter_Value"])
POINTS = 524288
out = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/train_bench.py  (4096 rays, 64+64; fp32 step, bf16 mixed-precision step, bf16x3 step)",
       "note": "median over the fine-pass-sized dispatches of each kernel (524 288 points). FETCH_SIZE doubled per MI355X_MICROARCH.md (wide coalesced reads are tallied at half size on gfx950); WRITE_SIZE as reported. dw_kernel: fp32 and bf16-state launches share one kernel name -- split by duration.",
       "kernels": {}}
import statistics
for name, disp in per.items():
    mx = max(d["ms"] for d in disp.values())
    if mx < 0.2:
        continue
    groups = {"": [d for d in disp.values() if d["ms"] > 0.6 * mx]}
    if "dw_kernel" in name and not any("dw_narrow_bf16" in n for n in per):    # (before the bf16-state narrow problems had a kernel
        # of their own, fp32 and bf16-state narrow launches shared this name: the fp32 launch is ~2x the bf16-state one)
        big = groups[""]
        groups = {" [longer launches]": big, " [shorter launches]": [d for d in disp.values() if 0.25 * mx < d["ms"] <= 0.6 * mx]}
    for suffix, ds in groups.items():
        f = [d["FETCH_SIZE"] for d in ds if "FETCH_SIZE" in d]
        w = [d["WRITE_SIZE"] for d in ds if "WRITE_SIZE" in d]
        if not f or not w:
            continue
        fetch, write, ms = 2 * 1024 * statistics.median(f), 1024 * statistics.median(w), statistics.median(d["ms"] for d in ds)
        mode = "bf16x3" if ("bf16x3" in name or "dw_kernel" in name) else "bf16" if ("bf16" in name or "shorter" in suffix) else "fp32"   # step the launch belongs to
        out["kernels"][name + suffix] = {"step": mode, "fetch_bytes_x2": fetch, "write_bytes": write, "ms": ms,
                                         "hbm_TB_per_s": (fetch + write) / ms / 1e9, "bytes_per_point": (fetch + write) / POINTS}
# the hash of the kernel sources the profile was measured ON: written by tools/gpu_round.sh on the measuring box (never recomputed here)
out["kernel_sources_sha"] = (open(f"{G}/kernel_sources_sha.txt").read().strip() if os.path.exists(f"{G}/kernel_sources_sha.txt") else "unstamped")
out["git_head"] = subprocess.run(["git", "describe", "--always", "--dirty"], capture_output=True, text=True).stdout.strip() or "unknown"
json.dump(out, open(f"{P}/{tag}_train_pmc.json", "w"), indent=1)
if out["kernels"]:                      # the pointer bench.py follows (a lexicographic sort of the file names picked r03_run7 over r03_run33)
    json.dump({"file": f"{tag}_train_pmc.json", "note": "written by tools/save_round.py: the profile bench.py's train_hbm_roofline reads (never picked by sorting file names)"},
              open(f"{P}/train_pmc_latest.json", "w"), indent=1)
print("saved", sorted(os.path.basename(p) for p in glob.glob(f"{P}/{tag}_*")))
for k, v in out["kernels"].items():
    print("  %-84s %.3f ms  %.2f TB/s  %.0f B/point" % (k, v["ms"], v["hbm_TB_per_s"], v["bytes_per_point"]))
