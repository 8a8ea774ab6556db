#!/bin/bash
mkdir -p gpurun_out/r3k
O=gpurun_out/r3k
python -m pytest tests/test_gpu_parity.py tests/test_gpu_epoch_r2.py -x -q -m gpu -k "packed or epoch or generator or iteration or signals or repeatable or window" > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
for spg in 32 8 1; do bash scripts/ab_variants.sh run "base new" scripts/time_dw.py --spg $spg | sort | uniq -c; done > $O/dw_ab.txt 2>&1
bash scripts/ab_variants.sh run "base new" >> $O/dw_ab.txt 2>&1
cat $O/summary.txt; tail -3 $O/t1.log; cat $O/dw_ab.txt
