"""Round 6 diagnosis: which of two queued epochs (9th, 10th) goes non-finite first, and at which loss row.  Two loss buffers = two captured graphs, alternated."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

S, N = int(sys.argv[1]), int(sys.argv[2])
sync_until = int(sys.argv[3])          # epochs run one by one (synchronised) before the queued pair
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(100)
cfg = bench.Cfg("x", S=S, B=64, n_windows=N, data="uniform")
eng, x = bench.build_engine(1, 0, True, dev, cfg)
nb = cfg.nb
L2 = [torch.empty(1, 11 * nb, 4, device=dev) for _ in range(2)]
perm_buf = torch.empty(6, nb * 64, dtype=torch.int32, device=dev)


def epoch(i):
    perm = torch.rand(6, N, device=dev, generator=gen).argsort(dim=1)[:, : nb * 64]
    perm_buf.copy_(perm)
    eng.train_epoch_graph(x, perm_buf, nb, 5, train_mode=True, losses=L2[i % 2], shuffle_windows=0)


def report(tag, l):
    l = l.cpu().numpy()[0]
    bad = np.flatnonzero(~np.isfinite(l).all(axis=1))
    if len(bad):
        b = int(bad[0])
        print(tag, "bad rows", len(bad), "first", b, "kind", "gen" if b >= 10 * nb else ("cx" if b % 2 == 0 else "cz"), "iteration", b // 2 if b < 10 * nb else b - 10 * nb,
              "rows", l[max(0, b - 1): b + 2].tolist(), flush=True)
    else:
        print(tag, "finite; counters", eng.counters.cpu().tolist()[:4], flush=True)
    return len(bad)


for e in range(sync_until):
    epoch(e)
    torch.cuda.synchronize()
    if report("epoch %d (alone)" % e, L2[e % 2]):
        sys.exit(0)
for pair in range(4):
    e = sync_until + 2 * pair
    epoch(e); epoch(e + 1)
    torch.cuda.synchronize()
    a = report("epoch %d (queued, first of pair)" % e, L2[e % 2])
    b = report("epoch %d (queued, second of pair)" % (e + 1), L2[(e + 1) % 2])
    if a or b:
        break
