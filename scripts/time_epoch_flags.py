"""ms per captured epoch under hypad_epoch_io.flags variants (alternated): default, DW_COLOC (the dW + Adam workgroups of a model on the
XCD of its generator chains), DW_SPREAD (the records in eight chunks, one per XCD).  Usage: time_epoch_flags.py [signals_per_gpu]"""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
nsig = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variants = {"default": 0, "dw_coloc": _C.EPOCH_DW_COLOC, "dw_spread": _C.EPOCH_DW_SPREAD}
engs = {}
for name, fl in variants.items():
    eng, x = bench.build_engine(nsig, 0, True, dev)
    eng.epoch_flags = fl
    step, losses = bench.make_step(eng, x, nsig, torch.Generator(device=dev).manual_seed(1), dev)
    for _ in range(5): step()
    engs[name] = (eng, x, step)
for rep in range(4):
    for name, (eng, x, step) in engs.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): step()
        torch.cuda.synchronize()
        print(nsig, "signals", name, "epoch ms %.3f" % ((time.perf_counter() - t0) / 30 * 1e3))
