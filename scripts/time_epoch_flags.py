"""ms per captured configs[1] epoch under hypad_epoch_io.flags variants (alternated), and whether the variants' losses / weights agree
bit for bit with the default's after the same epochs."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
names = sys.argv[1:] or ["default", "gen_resident"]
variants = {"default": 0, "dw_coloc": _C.EPOCH_DW_COLOC, "gen_resident": _C.EPOCH_GEN_RESIDENT}
engs = {}
for name in names:
    eng, x = bench.build_engine(1, 0, True, dev)
    eng.epoch_flags = variants[name]
    step, losses = bench.make_step(eng, x, 1, torch.Generator(device=dev).manual_seed(1), dev)
    for _ in range(3): step()
    torch.cuda.synchronize()
    print(name, "status", eng.status(), "finite", bool(torch.isfinite(losses).all()))
    engs[name] = (eng, x, step, losses)
ref = engs[names[0]]
for name in names[1:]:
    e = engs[name]
    same_l = torch.equal(ref[3], e[3])
    same_w = all(torch.equal(ref[0].params[k], e[0].params[k]) for k in ("enc", "dec", "cx", "cz"))
    print(name, "vs", names[0], ": losses bit-equal", same_l, " weights bit-equal", same_w, " max |dloss|", float((ref[3] - e[3]).abs().max()))
for rep in range(4):
    for name, (eng, x, step, losses) in engs.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60): step()
        torch.cuda.synchronize()
        print(name, "epoch ms %.3f" % ((time.perf_counter() - t0) / 60 * 1e3), "status", eng.status())
