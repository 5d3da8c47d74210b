cd /tmp; export TMPDIR=/tmp
R=/root/repo
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 6000 > $R/gpurun_out/counters_list.txt
for v in full:$R/sinnerf_amd/csrc/libsinnerf_hip.so nodma:$R/build/variants/lib_nodma7.so thin2:$R/build/variants/lib_thin27.so; do
  name=${v%%:*}; lib=${v#*:}
  for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
    SINNERF_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcx -o ${name}_$tag --pmc $pass -- python $R/tools/mlp_time.py --child 0 > $R/gpurun_out/pmcx_${name}_$tag.log 2>&1
  done
done
ls $R/gpurun_out/pmcx | head
