"""Build libhypad_hip.so (gfx950) in-tree with hipcc.  `python -m hypad_amd.build [--force] [--dev]`.

Two libraries:
* ``lib/libhypad_hip.so`` -- the product: exactly the entry points include/hypad.h declares.
* ``lib/libhypad_hip_dev.so`` (``--dev``; loaded when ``HYPAD_DEV_LIB=1``) -- the same sources compiled with
  ``-DHYPAD_DIAG=1`` plus ``csrc/diag.hip``: shader-clock stamps inside the training kernels and the ``hypad_diag_*``
  micro-benchmarks the scripts under ``scripts/diag_*.py`` drive.  Never loaded by the product path, tests or bench.py.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libhypad_hip.so")
DEV_LIB = os.path.join(LIB_DIR, "libhypad_hip_dev.so")
SOURCES = ["api_misc.hip", "ops_hyper.hip", "ops_dense.hip", "lstm_seq.hip", "train_iters.hip", "critic_fused.hip", "scoring.hip", "host_rng.cpp"]
DEV_SOURCES = SOURCES + ["diag.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17", "-Wno-pass-failed", "-Wno-unused-result"]


def source_digest():
    """sha256 over the kernel sources and the ABI header: what a committed profile (profiles/*.json) was measured on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "hypad.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _newest_source():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "hypad.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(f) for f in files)


def _fresh(lib):
    return os.path.exists(lib) and os.path.getmtime(lib) >= _newest_source()


def build(force=False, verbose=False, dev=False):
    lib = DEV_LIB if dev else LIB
    if not force and _fresh(lib):
        return lib
    os.makedirs(LIB_DIR, exist_ok=True)
    # one builder at a time (bench.py runs one process per GPU and every rank calls build()): the others wait on the lock
    # and then find the library fresh
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _fresh(lib):
                return lib
            return _build_locked(verbose, dev)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose, dev):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = DEV_LIB if dev else LIB
    objdir = os.path.join(LIB_DIR, "dev") if dev else LIB_DIR
    os.makedirs(objdir, exist_ok=True)
    flags = FLAGS + (["-DHYPAD_DIAG=1"] if dev else ["-DHYPAD_DIAG=0"])
    flags += os.environ.get("HYPAD_FLAGS", "").split()      # (A/B experiments: scripts/ab_libs.sh)
    if dev:                                    # kernel experiments: extra -D switches for the development library only
        flags += os.environ.get("HYPAD_DEV_FLAGS", "").split()
    objs = []
    procs = []
    for src in (DEV_SOURCES if dev else SOURCES):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if src.endswith(".cpp"):               # host-only helpers (no device code): the system compiler
            cmd = [os.environ.get("CXX", "g++"), "-O3", "-fPIC", "-fvisibility=hidden", "-std=c++17", "-ffp-contract=off", "-c", os.path.join(CSRC, src), "-o", obj]
        else:
            cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = lib + ".tmp.%d" % os.getpid()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
                           "-o", tmp, *objs])
    os.replace(tmp, lib)                       # atomic: a concurrent importer never maps a half-written library
    return lib


def kernel_metadata(obj):
    """[{name, vgpr_count, vgpr_spill_count, sgpr_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size, ...}]
    of the gfx950 code object inside a host object file built above (the .hip_fatbin section, un-bundled with the ROCm LLVM tools;
    names demangled): what tests/test_cabi_and_host.py holds the compile-time instantiations to."""
    import re
    import tempfile
    llvm = os.environ.get("HYPAD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    for line in notes.splitlines():                          # (kernel entries: "  - .agpr_count: 0" then "    .key: value" lines; nested lists sit deeper)
        m = re.match(r"^  ([- ]) \.(\w+):\s*(.*)$", line)
        if not m:
            continue
        if m.group(1) == "-":
            cur = {}
            out.append(cur)
        if cur is None:
            continue
        key, val = m.group(2), m.group(3).strip().strip("'")
        if key == "name":
            cur["name"] = val
        elif key in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
                     "agpr_count", "max_flat_workgroup_size", "wavefront_size"):
            cur[key] = int(val)
    out = [k for k in out if "name" in k]
    names = [k["name"] for k in out]
    if names:
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        for k, n in zip(out, dem):
            k["mangled"], k["name"] = k["name"], n
    return [k for k in out if "vgpr_count" in k]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
