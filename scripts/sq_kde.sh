#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/sq_kde; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; export EPOCH32=0
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/scripts/sq_target.py > $OUT/p$i.log 2> $OUT/p$i.err
done
cd $REPO
python3 scripts/sq_summary.py $OUT/p1 $OUT/p2 --match "kde_mode|unroll_median" 
python3 scripts/time_kde.py
