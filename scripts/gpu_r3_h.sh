#!/bin/bash
mkdir -p gpurun_out/r3h
O=gpurun_out/r3h
for rep in 1 2 3; do for c in 0 1; do echo -n "gen_warm=$c: "; HYPAD_GEN_WARM=$c python scripts/time_graph.py --graph-only 2>&1 | tail -1; done; done > $O/ab_warm.txt 2>&1
for c in 0 1; do echo -n "gen_warm=$c 8 signals: "; HYPAD_GEN_WARM=$c python scripts/time_graph.py --graph-only --spg 8 2>&1 | tail -1; done >> $O/ab_warm.txt 2>&1
for c in 0 1; do echo "gen_warm=$c"; HYPAD_GEN_WARM=$c python scripts/epoch_ab.py 2>&1 | grep "^kind"; done >> $O/ab_warm.txt 2>&1
python -m pytest tests/test_gpu_epoch_r2.py tests/test_gpu_parity.py -x -q -m gpu -k "generator or epoch or iteration or repeatable" > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
cat $O/summary.txt $O/ab_warm.txt; tail -3 $O/t1.log
