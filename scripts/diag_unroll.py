import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"   # development library: python -m hypad_amd.build --dev
"""Where a tile of unroll_median_kernel spends its time (dev library: shader-clock stamps of workgroup 37's first tile) and how
often the two-pivot filter decides the median (counts over the whole launch)."""
import ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hypad_amd import _C
setp = _C.lib.hypad_diag_set_unroll_stamps
setp.restype = None; setp.argtypes = [ctypes.c_void_p]
n, S = 125_000, 100
g = torch.Generator(device="cuda").manual_seed(3)
y = torch.randn(n, S, device="cuda", generator=g).contiguous()
med = torch.empty(n + S - 1, device="cuda")
st = torch.zeros(16, dtype=torch.int64, device="cuda")
for filt, tile in (("1", "128"), ("1", "64"), ("0", "128"), ("0", "64")):
    os.environ["HYPAD_UNROLL_FILTER"] = filt
    os.environ["HYPAD_UNROLL_TILE"] = tile
    setp(None)
    for _ in range(3):
        _C.lib.hypad_unroll_median(_C.ptr(y), _C.ptr(med), None, n, S, _C.stream())
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        _C.lib.hypad_unroll_median(_C.ptr(y), _C.ptr(med), None, n, S, _C.stream())
    b.record(); torch.cuda.synchronize()
    st.zero_(); setp(st.data_ptr())
    _C.lib.hypad_unroll_median(_C.ptr(y), _C.ptr(med), None, n, S, _C.stream())          # stamps only
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    st.zero_(); st[14] = 1
    _C.lib.hypad_unroll_median(_C.ptr(y), _C.ptr(med), None, n, S, _C.stream())          # counts only (contended atomics: not timed)
    torch.cuda.synchronize()
    s[8:10] = st.cpu().tolist()[8:10]
    print(f"filter={filt} tile={tile}: {a.elapsed_time(b) / 10 * 1e3:.1f} us per launch; tile of workgroup 37: stage {s[1] - s[0]} cycles, barrier {s[2] - s[1]}, "
          f"16 timesteps of wave 0 {s[3] - s[2]} ({(s[3] - s[2]) / 16:.0f} each); filter decided {s[8]} timesteps, full count {s[9]}")
