#!/bin/bash
mkdir -p gpurun_out/r3m
O=gpurun_out/r3m
python -m pytest tests/test_gpu_epoch_r2.py tests/test_gpu_status_r3.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "epoch or signal or trajectory or multivariate or window" > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/summary.txt
bash scripts/ab_variants.sh run "base new" > $O/ab_split.txt 2>&1
bash scripts/ab_variants.sh run "base new" scripts/time_graph.py --graph-only --spg 8 >> $O/ab_split.txt 2>&1
python scripts/diag_persistent.py > $O/diag_persistent.txt 2>&1
cat $O/summary.txt; tail -3 $O/t1.log $O/t2.log; cat $O/ab_split.txt; head -19 $O/diag_persistent.txt
