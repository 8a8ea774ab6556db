"""When does each wave (work item) of the generator's dW + Adam launch finish?  Wall clock (s_memrealtime, 100 MHz) relative to the
earliest finisher: the launch is as long as its slowest item.  Development library."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"
import ctypes, collections
import numpy as np, torch
import bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
st = torch.zeros(3 * 48 * 8 + 64 + 2 * 1024, dtype=torch.int64, device=dev)
fn = _C.lib.hypad_diag_set_gen_stamps
fn.restype = None; fn.argtypes = [ctypes.c_void_p]
fn(st.data_ptr())
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
for _ in range(5):
    eng.decoder_iteration(x, idx, None, True)
torch.cuda.synchronize()
t = st.cpu().numpy()[3 * 48 * 8 + 64:].reshape(1024, 2)
live = t[t[:, 1] >= 0]
t0 = live[:, 0].min()
kinds = {0: "weight tile", 1: "bias", 2: "decay", 3: "ball bias"}
by = collections.defaultdict(list)
for end, code in live:
    by[(int(code) // 1000, (int(code) % 1000) // 100, int(code) % 100)].append((end - t0) / 100.0)
print(f"{len(live)} items; last finisher {(live[:, 0].max() - t0) / 100.0:.2f} us after the first")
for (k, net, rr), v in sorted(by.items(), key=lambda kv: -max(kv[1])):
    print(f"  {kinds[k]:12s} net {net} reduction rows {16 * rr:4d}: {len(v):4d} items, finish {min(v):5.2f} .. {max(v):5.2f} us (median {np.median(v):5.2f})")
