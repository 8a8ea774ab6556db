"""hypad_epoch_shuffles launch time (HIP events, GPU side), 6 passes of 1 916 windows."""
import sys
sys.path.insert(0, ".")
import torch
from hypad_amd import _C
buf = torch.empty(6, 29 * 64, dtype=torch.int32, device="cuda")
cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
fn = lambda: _C.check(_C.lib.hypad_epoch_shuffles(_C.ptr(buf), 6, 29 * 64, 1916, 1234, _C.ptr(cnt), _C.stream()), "shuffles")
fn(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(5_000_000); a.record()
for _ in range(50): fn()
b.record(); torch.cuda.synchronize()
print("epoch_shuffle_kernel us per launch %.2f" % (a.elapsed_time(b) / 50 * 1e3), "checksum", int(buf.long().sum()))
