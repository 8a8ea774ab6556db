#!/bin/bash
# A/B/C... timing on one box: several builds of the library from the working tree, each with its own -D switches.
#   (container)  bash scripts/ab_variants.sh prepare name1="-DX=0" name2="-DY=0 -DZ=1" ...   -> ab_libs/<name>.so  (+ base.so = HEAD, head.so = working tree as is)
#   (GPU box)    bash scripts/ab_variants.sh run "name1 name2 ..." script.py [args]
set -u
if [ "$1" = "prepare" ]; then
  shift
  mkdir -p ab_libs
  if [ "${NO_BASE:-0}" != "1" ]; then
    rm -rf /tmp/ab_head && mkdir -p /tmp/ab_head && git archive HEAD | tar -x -C /tmp/ab_head
    (cd /tmp/ab_head && python -m hypad_amd.build > /dev/null) && cp /tmp/ab_head/hypad_amd/lib/libhypad_hip.so ab_libs/base.so
  fi
  for spec in "$@"; do
    name="${spec%%=*}"; flags="${spec#*=}"
    rm -rf /tmp/ab_var && mkdir -p /tmp/ab_var/hypad_amd && cp -r hypad_amd/csrc hypad_amd/build.py hypad_amd/__init__.py /tmp/ab_var/hypad_amd/ && cp -r include /tmp/ab_var/
    (cd /tmp/ab_var && HYPAD_FLAGS="$flags" python -c "import sys; sys.path.insert(0, 'hypad_amd'); import build; build.build(force=True)" > /dev/null) && cp /tmp/ab_var/hypad_amd/lib/libhypad_hip.so ab_libs/$name.so
    echo "$name: $flags"
  done
  ls -la ab_libs
else
  names="$2"; script="$3"; shift 3
  for rep in 1 2 3; do
    for v in $names; do
      echo -n "$v: "; HYPAD_LIB_PATH=$(pwd)/ab_libs/$v.so timeout 300 python scripts/$script "$@" 2>&1 | tail -1
    done
  done
fi
