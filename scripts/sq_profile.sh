#!/bin/bash
# SQ counter passes over scripts/sq_target.py (instruction mix, active / wait cycles, LDS conflicts) -> gpurun_out/sq/summary.json
REPO=$(pwd); OUT=$REPO/gpurun_out/sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/scripts/sq_target.py > $OUT/p$i.log 2> $OUT/p$i.err
  echo "pass $i rc=$?"
done
cd $REPO
python3 scripts/sq_summary.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 > $OUT/summary.json
find $OUT -name "*counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete
cat $OUT/summary.json
