#!/bin/bash
mkdir -p gpurun_out/r3l
O=gpurun_out/r3l
python -m pytest tests/test_gpu_status_r3.py -x -q -m gpu > $O/t1.log 2>&1; echo "t1 rc=$?" > $O/summary.txt
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-scoring --no-drop-in --host-shuffle > $O/bench_hostshuf_$rep.json 2> $O/bench_hostshuf_$rep.err; echo "bench host rc=$?" >> $O/summary.txt
  timeout 300 python bench.py --no-cpu-baseline --no-scoring --no-drop-in > $O/bench_devshuf_$rep.json 2> $O/bench_devshuf_$rep.err; echo "bench dev rc=$?" >> $O/summary.txt
done
cat $O/summary.txt; tail -5 $O/t1.log
