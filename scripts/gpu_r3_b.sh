#!/bin/bash
# round 3, second GPU pass: scoring kernels (rolling mean, statistics, LDS-staged un-roll + pivot filter, symmetric KDE screen), status channel, RCCL
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
python -m pytest tests/test_gpu_status_r3.py tests/test_gpu_autograd_r2.py tests/test_gpu_scoring_r3.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scor or kde or critic_smoothing or sharded" > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/summary.txt
python -m pytest tests/test_gpu_sharded_r2.py tests/test_gpu_rccl_r3.py tests/test_gpu_pins_r2.py -x -q -m gpu -s > $O/t3.log 2>&1; echo "t3 rc=$?" >> $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-drop-in > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
HYPAD_UNROLL_FILTER=0 timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-drop-in --no-sharded-scoring > $O/bench_nofilter.json 2> $O/bench_nofilter.err; echo "bench2 rc=$?" >> $O/summary.txt
cat $O/summary.txt
for f in t1 t2 t3; do echo "== $f"; tail -n 6 $O/$f.log; done
