"""ms per captured configs[1] epoch under hypad_epoch_io.flags variants (alternated): 0, DW_COLOC (the dW + Adam workgroups of a model on the
XCD of its generator chains)."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
variants = {"default": 0, "dw_coloc": _C.EPOCH_DW_COLOC}
engs = {}
for name, fl in variants.items():
    eng, x = bench.build_engine(1, 0, True, dev)
    eng.epoch_flags = fl
    step, losses = bench.make_step(eng, x, 1, torch.Generator(device=dev).manual_seed(1), dev)
    for _ in range(5): step()
    engs[name] = (eng, x, step)
for rep in range(4):
    for name, (eng, x, step) in engs.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60): step()
        torch.cuda.synchronize()
        print(name, "epoch ms %.3f" % ((time.perf_counter() - t0) / 60 * 1e3))
