"""Two kernels on two streams handing payloads to each other through flags on ONE XCD (plain stores + drain + flag; sc1 polls and
loads): do they run beside each other (eager and inside a captured graph), is the hand-off correct, what does a round trip cost?
Development library (python -m hypad_amd.build --dev)."""
import os, sys; sys.path.insert(0, "."); os.environ["HYPAD_DEV_LIB"] = "1"
import ctypes
import torch
from hypad_amd import _C
from hypad_amd import streams as hs
dev = torch.device("cuda", 0)
fn = _C.lib.hypad_diag_pair
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
main = torch.cuda.current_stream()
side = hs.beside([main], dev)
plain = torch.cuda.Stream(device=dev)
for words in (1, 1024, 16384):
    for label, other in (("probed side stream", side), ("fresh stream", plain)):
        flags = torch.zeros(2, dtype=torch.int32, device=dev); payload = torch.zeros(2 * words, dtype=torch.int32, device=dev); out = torch.zeros(6, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        other.wait_stream(main)
        rounds = 200
        rc = fn(0, rounds, words, flags.data_ptr(), payload.data_ptr(), out.data_ptr(), main.cuda_stream, other.cuda_stream)
        torch.cuda.synchronize()
        o = out.cpu().tolist()
        print("eager  %-18s words %6d rc %d  cycles/round %.0f / %.0f  bad %d / %d  gave up %d / %d" % (label, words, rc, o[0] / rounds, o[1] / rounds, o[2], o[3], o[4], o[5]))
# inside a captured graph: fork / join through events
words, rounds = 1024, 200
flags = torch.zeros(2, dtype=torch.int32, device=dev); payload = torch.zeros(2 * words, dtype=torch.int32, device=dev); out = torch.zeros(6, dtype=torch.int64, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    side2 = torch.cuda.Stream(device=dev)
    side2.wait_stream(cur)
    fn(0, rounds, words, flags.data_ptr(), payload.data_ptr(), out.data_ptr(), cur.cuda_stream, side2.cuda_stream)
    cur.wait_stream(side2)
for rep in range(3):
    flags.zero_(); out.zero_(); payload.zero_()
    g.replay(); torch.cuda.synchronize()
    o = out.cpu().tolist()
    print("graph replay %d: cycles/round %.0f / %.0f  bad %d / %d  gave up %d / %d" % (rep, o[0] / rounds, o[1] / rounds, o[2], o[3], o[4], o[5]))
