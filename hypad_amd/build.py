"""Build libhypad_hip.so (gfx950) in-tree with hipcc.  `python -m hypad_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libhypad_hip.so")
SOURCES = ["api_misc.hip", "ops_hyper.hip", "ops_dense.hip", "train_iters.hip", "critic_fused.hip", "scoring.hip", "diag.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed", "-Wno-unused-result"]


def _newest_source():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "hypad.h")]
    return max(os.path.getmtime(f) for f in files)


def _fresh():
    return os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source()


def build(force=False, verbose=False):
    if not force and _fresh():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    # one builder at a time (bench.py runs one process per GPU and every rank calls build()): the others wait on the lock
    # and then find the library fresh
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _fresh():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = LIB + ".tmp.%d" % os.getpid()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs])
    os.replace(tmp, LIB)                       # atomic: a concurrent importer never maps a half-written library
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
