#!/bin/bash
# A/B of compile-time variants on ONE box.
#   here:        bash scripts/ab_variants.sh build NAME "-DFLAG=.. -DFLAG2=.."   (repeat per variant; NAME=base with "" for the working tree)
#   on the box:  bash scripts/ab_variants.sh run "NAME1 NAME2 .." [script] [args..]  (default script: scripts/time_graph.py --graph-only)
set -u
if [ "$1" = "build" ]; then
  mkdir -p ab_libs /tmp/ab_build_$2
  HYPAD_FLAGS="$3" python - <<PY
import os, shutil, subprocess, sys
sys.path.insert(0, ".")
from hypad_amd import build as b
objdir = "/tmp/ab_build_$2"
flags = b.FLAGS + ["-DHYPAD_DIAG=0"] + os.environ.get("HYPAD_FLAGS", "").split()
procs = []
for src in b.SOURCES:
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    procs.append((obj, subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(b.CSRC, src), "-o", obj])))
objs = []
for obj, p in procs:
    assert p.wait() == 0, obj
    objs.append(obj)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "ab_libs/$2.so", *objs])
print("built ab_libs/$2.so with", os.environ.get("HYPAD_FLAGS", ""))
PY
else
  names=$2; script=${3:-scripts/time_graph.py}; shift; shift; shift
  args=${@:---graph-only}
  for rep in 1 2 3; do
    for v in $names; do
      echo -n "$v: "; HYPAD_LIB_PATH=$(pwd)/ab_libs/$v.so timeout 300 python $script $args 2>&1 | tail -1
    done
  done
fi
