#!/bin/bash
# round 3, third GPU pass: XCD placement of the resident critic launch (A/B on one box), scoring kernels, drop-in path
mkdir -p gpurun_out/r3c
O=gpurun_out/r3c
python -m pytest tests/test_gpu_status_r3.py tests/test_gpu_scoring_r3.py tests/test_gpu_rccl_r3.py -x -q -m gpu -s > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scor or kde or critic_smoothing or sharded or drop or host or iteration" > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/summary.txt
for rep in 1 2; do
  HYPAD_CRITIC_XCD=0 timeout 600 python bench.py --no-cpu-baseline --no-scoring --no-drop-in > $O/bench_xcd0_$rep.json 2> $O/bench_xcd0_$rep.err; echo "bench xcd0 rc=$?" >> $O/summary.txt
  HYPAD_CRITIC_XCD=1 timeout 600 python bench.py --no-cpu-baseline --no-scoring --no-drop-in > $O/bench_xcd1_$rep.json 2> $O/bench_xcd1_$rep.err; echo "bench xcd1 rc=$?" >> $O/summary.txt
done
timeout 900 python bench.py --no-cpu-baseline --no-secondary > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?" >> $O/summary.txt
cat $O/summary.txt
for f in t1 t2; do echo "== $f"; tail -n 8 $O/$f.log; done
