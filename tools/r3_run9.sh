#!/bin/bash
# memory-path counters of the hand-scheduled training forward / chain (and the weight-gradient kernel as the 6.9 TB/s yardstick)
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
P="rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/mp"
for s in fwd_t_time chain_t_time; do
timeout 300 $P --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY -o ${s}_a -- python $R/tools/$s.py > $R/gpurun_out/mp_${s}_a.log 2>&1; echo "$s a exit $?"
timeout 300 $P --pmc TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum -o ${s}_b -- python $R/tools/$s.py > $R/gpurun_out/mp_${s}_b.log 2>&1; echo "$s b exit $?"
timeout 300 $P --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE -o ${s}_c -- python $R/tools/$s.py > $R/gpurun_out/mp_${s}_c.log 2>&1; echo "$s c exit $?"
done
cd $R
python - <<'PY' | tee gpurun_out/mem_path.txt
import csv, glob, collections
for s in ("fwd_t_time", "chain_t_time"):
    for tag in "abc":
        fs = glob.glob(f"gpurun_out/mp/**/{s}_{tag}_counter_collection.csv", recursive=True)
        if not fs: print(s, tag, "no counters"); continue
        per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
        for r in csv.DictReader(open(fs[0])):
            if "_t_kernel" not in r["Kernel_Name"]: continue
            k = (r["Kernel_Name"][:40], r["Dispatch_Id"]); per[k][r["Counter_Name"]] += float(r["Counter_Value"]); dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for name in sorted({k[0] for k in dur}):
            ks = sorted([k for k in dur if k[0] == name], key=lambda k: dur[k])
            k = ks[len(ks) // 2]
            print(s, tag, name, "ms %.3f" % dur[k], {n: "%.4g" % v for n, v in per[k].items()})
PY
