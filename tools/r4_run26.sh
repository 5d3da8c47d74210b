#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp R=$PWD
./tools/ubench/lds_b128_banks | tee gpurun_out/lds_b128_banks.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r26 -o u --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- $R/tools/ubench/lds_b128_banks > /dev/null 2>&1
cd $R
python - <<'PY' | tee -a gpurun_out/lds_b128_banks.txt
import csv, glob, collections, re
names = ["identity", "stride128_g8", "half_swap", "j16_alias", "pitch144", "pitch136"]; ops = ["ds_write_b128", "ds_read_b128", "ds_write_b64"]
for f in glob.glob("gpurun_out/r26/**/u_counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); nm = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); nm[r["Dispatch_Id"]] = r["Kernel_Name"]
    for d in sorted(per, key=int):
        c = per[d]
        if c["SQ_INSTS_LDS"] < 1e6: continue
        m = re.search(r"k<(\d+), (\d+)>", nm[d])
        print("%-14s %-14s cycles/instr %.2f  conflict cycles/instr %.2f" % (ops[int(m.group(2))], names[int(m.group(1))], c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"], c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"]))
PY
