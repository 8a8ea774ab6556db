#!/bin/bash
mkdir -p gpurun_out/r3g
O=gpurun_out/r3g
python -m pytest tests/test_gpu_epoch_r2.py tests/test_gpu_status_r3.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
bash scripts/ab_variants.sh run "noprefetch new" > $O/ab_prefetch.txt 2>&1
for rep in 1 2 3; do for c in 1 0; do echo -n "clear_each=$c: "; HYPAD_CRITIC_CLEAR=$c python scripts/time_graph.py --graph-only 2>&1 | tail -1; done; done > $O/ab_clear.txt 2>&1
cat $O/summary.txt; tail -3 $O/t1.log; cat $O/ab_prefetch.txt $O/ab_clear.txt
