#!/bin/bash
# round 3, first GPU pass: the new tests first, then the whole -m gpu suite, then the default bench
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_status_r3.py tests/test_gpu_rccl_r3.py tests/test_gpu_autograd_r2.py -x -q -m gpu -s > gpurun_out/r3a/new_tests.log 2>&1
echo "new tests rc=$?" > gpurun_out/r3a/summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kde or critic_smoothing or score" > gpurun_out/r3a/kde_tests.log 2>&1
echo "kde tests rc=$?" >> gpurun_out/r3a/summary.txt
timeout 900 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
echo "bench rc=$?" >> gpurun_out/r3a/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3a/all_tests.log 2>&1
echo "all tests rc=$?" >> gpurun_out/r3a/summary.txt
cat gpurun_out/r3a/summary.txt
tail -5 gpurun_out/r3a/new_tests.log gpurun_out/r3a/kde_tests.log gpurun_out/r3a/all_tests.log
