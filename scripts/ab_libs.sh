#!/bin/bash
# A/B timing on one box: the library of the last commit (built from `git stash`ed sources is not possible on the GPU box: no .git
# there) vs the working tree.  Usage (in the container): bash scripts/ab_libs.sh prepare   -> builds gpurun_ab/base.so from HEAD
#                        (on the GPU box):   bash scripts/ab_libs.sh run       -> alternates the two libraries
set -u
if [ "$1" = "prepare" ]; then
  rm -rf /tmp/ab_head && mkdir -p /tmp/ab_head && git archive HEAD | tar -x -C /tmp/ab_head
  (cd /tmp/ab_head && python -m hypad_amd.build > /dev/null) && mkdir -p ab_libs && cp /tmp/ab_head/hypad_amd/lib/libhypad_hip.so ab_libs/base.so
  python -m hypad_amd.build > /dev/null && cp hypad_amd/lib/libhypad_hip.so ab_libs/new.so
  ls -la ab_libs
else
  for rep in 1 2 3 4; do
    for v in base new; do
      echo -n "$v: "; HYPAD_LIB_PATH=$(pwd)/ab_libs/$v.so timeout 300 python scripts/${2:-time_graph.py} ${3:-} ${4:-} ${5:-} 2>&1 | tail -1
    done
  done
fi
