#!/bin/bash
mkdir -p gpurun_out/r3i
O=gpurun_out/r3i
bash scripts/ab_variants.sh run "base new" > $O/ab_phaseb.txt 2>&1
python -m pytest tests/test_gpu_epoch_r2.py tests/test_gpu_status_r3.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
python scripts/diag_persistent.py > $O/diag_persistent.txt 2>&1
cat $O/summary.txt $O/ab_phaseb.txt; tail -3 $O/t1.log; head -20 $O/diag_persistent.txt
