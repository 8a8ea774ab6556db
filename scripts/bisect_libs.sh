#!/bin/bash
# build the library of several commits into ab_libs/<sha>.so (container), then on the GPU box run the flaky tests N times per library
set -u
if [ "$1" = "prepare" ]; then
  shift
  mkdir -p ab_libs
  for c in "$@"; do
    rm -rf /tmp/bis_$c && mkdir -p /tmp/bis_$c && git archive $c | tar -x -C /tmp/bis_$c
    (cd /tmp/bis_$c && python -m hypad_amd.build > /dev/null 2>&1) && cp /tmp/bis_$c/hypad_amd/lib/libhypad_hip.so ab_libs/$c.so && echo built $c
  done
else
  shift
  for c in "$@"; do
    fails=0
    for i in 1 2 3 4 5 6; do
      HYPAD_LIB_PATH=$(pwd)/ab_libs/$c.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "other_window_sizes or hoisted_critic_phase_other_shapes" 2>&1 | grep -q failed && fails=$((fails+1))
    done
    echo "$c: $fails of 6 runs had a failure"
  done
fi
