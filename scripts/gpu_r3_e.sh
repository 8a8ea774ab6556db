#!/bin/bash
# round 3, fifth GPU pass: 32 x 32 weight tiles in the dW + Adam launch (many signals per GPU), A/B on one box
mkdir -p gpurun_out/r3e
O=gpurun_out/r3e
python -m pytest tests/test_gpu_epoch_r2.py tests/test_gpu_status_r3.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "epoch or signal or multi or packed or trajectory" > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/summary.txt
for n in 8 16 32; do
  for t in 0 1; do
    HYPAD_DW_TILE32=$t timeout 600 python bench.py --signals-per-gpu $n --steps 20 --warmup 5 --no-cpu-baseline --no-scoring --no-drop-in > $O/bench_s${n}_t$t.json 2> $O/bench_s${n}_t$t.err; echo "bench s$n t$t rc=$?" >> $O/summary.txt
  done
done
timeout 600 python bench.py --no-cpu-baseline --no-scoring --no-drop-in > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" >> $O/summary.txt
cat $O/summary.txt
for f in t1 t2; do echo "== $f"; tail -n 6 $O/$f.log; done
