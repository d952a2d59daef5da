#!/bin/bash
# SQ counters of the attention forward, one wave group vs two wave groups one segment apart.  GPU box; counters in their own passes.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  for shape in "" "--long"; do
    d=/tmp/pp_sq_${i}${shape}
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $d -o c -- python $R/tools/probes/attn_pp_counters.py $shape > $O/r6_attn_pp_sq_${i}${shape}.log 2>&1
    echo "### counter set $i  shape ${shape:-B128_L417}"
    python $R/tools/pmc_sq.py $(find $d -name "*counter_collection.csv" | head -1) --match attn32
  done
done
