#!/bin/bash
mkdir -p gpurun_out/r3f
O=gpurun_out/r3f
python scripts/diag_unroll.py > $O/diag_unroll.txt 2>&1
python -m pytest tests/test_gpu_scoring_r3.py tests/test_gpu_sharded_r2.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scor or kde or critic_smoothing or sharded" > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-drop-in > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
HYPAD_UNROLL_TILE=64 timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-drop-in --no-sharded-scoring > $O/bench_t64.json 2> $O/bench_t64.err; echo "bench t64 rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/diag_unroll.txt
for f in t1 t2; do echo "== $f"; tail -n 4 $O/$f.log; done
