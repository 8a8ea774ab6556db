import sys, time, ctypes; sys.path.insert(0, ".")
import numpy as np
from hypad_amd import _C, host_rng
zx = np.zeros((145, 1280), np.float32); zz = np.zeros_like(zx); zg = np.zeros((29, 1280), np.float32)
def best(f, n=20):
    b = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
print("get_state ms", best(lambda: np.random.get_state()))
st = np.random.get_state()
print("set_state ms", best(lambda: np.random.set_state(st)))
key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
pos, has, cached = ctypes.c_int(int(st[2])), ctypes.c_int(0), ctypes.c_double(0.0)
ptrs = (ctypes.c_void_p * 2)(zx.ctypes.data, zz.ctypes.data)
def raw():
    _C.lib.hypad_host_mt19937_normal(key.ctypes.data, ctypes.byref(pos), ctypes.byref(has), ctypes.byref(cached), ptrs, 2, 1280, 145)
print("raw ctypes call (critic planes) ms", best(raw))
print("global_normal_into critic planes ms", best(lambda: host_rng.global_normal_into([zx, zz], 1280, 145)))
print("global_normal_into gen plane ms", best(lambda: host_rng.global_normal_into([zg], 1280, 29)))
