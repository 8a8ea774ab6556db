"""Generator-phase-only epochs (n_critics = 0), stepwise launches vs the two resident launches: per-step loss rows, weights, time."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
from hypad_amd import _C
dev = torch.device("cuda", 0)
B, N = bench.B, bench.N_WINDOWS
TRAIN = "--eval" not in sys.argv
REPS = 0
for nb in (1, 2, 4, 29):
    res = {}
    for name, fl in (("stepwise", 0), ("resident", _C.EPOCH_GEN_RESIDENT)):
        eng, x = bench.build_engine(1, 0, True, dev)
        eng.epoch_flags = fl
        perm = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(5))[: nb * B] for _ in range(1)]).to(torch.int32).to(dev)
        losses = torch.zeros(1, nb, 4, device=dev)
        eng.train_epoch(x, perm, nb, 0, TRAIN, losses=losses)
        torch.cuda.synchronize()
        first = losses.clone()
        t0 = time.perf_counter()
        for rep in range(REPS):
            eng.train_epoch(x, perm, nb, 0, TRAIN, losses=losses)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / max(REPS, 1)
        losses = first
        res[name] = (losses.clone(), {k: eng.params[k].clone() for k in ("enc", "dec")}, dt, eng.status(), eng.counters.cpu().tolist())
    a, b = res["stepwise"], res["resident"]
    print("nb %2d: stepwise %.1f us/epoch, resident %.1f us/epoch; status %d; counters %s vs %s" % (nb, a[2] * 1e6, b[2] * 1e6, b[3], a[4][:4], b[4][:4]))
    print("   losses equal %s  enc equal %s  dec equal %s  max|dloss| %.3g  max|denc| %.3g max|ddec| %.3g" % (
        torch.equal(a[0], b[0]), torch.equal(a[1]["enc"], b[1]["enc"]), torch.equal(a[1]["dec"], b[1]["dec"]),
        float((a[0] - b[0]).abs().max()), float((a[1]["enc"] - b[1]["enc"]).abs().max()), float((a[1]["dec"] - b[1]["dec"]).abs().max())))
    if nb <= 2:
        print("   stepwise", a[0][0].cpu().tolist()); print("   resident", b[0][0].cpu().tolist())
