import sys, os
if "--dev" in sys.argv: os.environ["HYPAD_DEV_LIB"] = "1"
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
B, N = bench.B, bench.N_WINDOWS
nb = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
res = {}
for name, fl in (("stepwise", 0), ("resident", _C.EPOCH_GEN_RESIDENT)):
    eng, x = bench.build_engine(1, 0, True, dev)
    eng.epoch_flags = fl
    perm = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(5))[: nb * B]]).to(torch.int32).to(dev)
    losses = torch.zeros(1, nb, 4, device=dev)
    eng.train_epoch(x, perm, nb, 0, False, losses=losses)
    torch.cuda.synchronize()
    res[name] = {net: eng.state_dict(net) for net in ("enc", "dec")}
    res[name]["m"] = {net: eng.exp_avg[net].clone() for net in ("enc", "dec")}
    if name == "resident":
        import ctypes
        o_, st_, c_ = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _C.check(_C.lib.hypad_packed_region(ctypes.byref(eng.dims), ctypes.byref(o_), ctypes.byref(st_), ctypes.byref(c_)), "packed_region")
        items_off = (o_.value + c_.value + 63) & ~63
        sync_off = items_off + 32 * 2048
        words = eng.workspace[sync_off: sync_off + 32].view(torch.int32).cpu().tolist()
        print("sync block: ready", words[0:7], "done", words[8:15], "err", hex(words[16]), "claims chain / dW", words[17], words[18])
for net in ("enc", "dec"):
    for k, v in res["stepwise"][net].items():
        d = (v - res["resident"][net][k]).abs()
        if float(d.max()) > 0:
            bad = (d > 0).nonzero()
            print(net, k, tuple(v.shape), "max", float(d.max()), "n bad", bad.shape[0], "first", bad[0].tolist(), "last", bad[-1].tolist())
    dm = (res["stepwise"]["m"][net] - res["resident"]["m"][net]).abs()
    print(net, "exp_avg max diff", float(dm.max()), "n", int((dm > 0).sum()))
