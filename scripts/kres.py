#!/usr/bin/env python3
"""Register / spill census of one csrc file's kernels (hipcc -Rpass-analysis=kernel-resource-usage):
    python scripts/kres.py critic_fused.hip [--root /other/tree] [extra hipcc flags]"""
import os
import re
import subprocess
import sys

args = sys.argv[1:]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if "--root" in args:
    i = args.index("--root")
    root = args[i + 1]
    del args[i:i + 2]
src, extra = args[0], args[1:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed", "-DHYPAD_DIAG=0", *extra,
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(root, "hypad_amd", "csrc", src), "-o", "/tmp/kres.%d.o" % os.getpid()]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp")
keys = ("TotalSGPRs", "VGPRs:", "AGPRs", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill")
cur = None
for line in out.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur:
            print(cur)
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|\(hypad.*$", "", name)[:60].ljust(60)
    elif cur and t.startswith(keys):
        cur += " | " + t.replace(" [bytes/lane]", "").replace(" [waves/SIMD]", "")
if cur:
    print(cur)
try:
    os.remove("/tmp/kres.%d.o" % os.getpid())
except OSError:
    pass
