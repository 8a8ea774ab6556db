"""Eager vs hipGraph replay of the configs[1] epoch (and --spg signals per GPU)."""
import sys, time
sys.path.insert(0, ".")
import argparse, torch, bench
ap = argparse.ArgumentParser(); ap.add_argument("--spg", type=int, default=1); ap.add_argument("--reps", type=int, default=30); ap.add_argument("--graph-only", action="store_true"); args = ap.parse_args()
dev = torch.device("cuda", 0)
for mode in (("graph",) if args.graph_only else ("eager", "graph")):
    eng, x = bench.build_engine(args.spg, 0, True, dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    losses = torch.empty(args.spg, 11 * bench.N_BATCHES, 4, device=dev)
    buf = torch.empty(6, bench.N_BATCHES * bench.B, dtype=torch.int32, device=dev)
    def step():
        perm = torch.rand(6, bench.N_WINDOWS, device=dev, generator=gen).argsort(dim=1)[:, : bench.N_BATCHES * bench.B]
        buf.copy_(perm)
        if mode == "eager":
            eng.train_epoch(x, buf, bench.N_BATCHES, 5, True, losses=losses)
        else:
            eng.train_epoch_graph(x, buf, bench.N_BATCHES, 5, True, losses=losses)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.reps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.reps
    c0 = time.perf_counter()
    for _ in range(30): step()
    cpu = (time.perf_counter() - c0) / 30
    torch.cuda.synchronize()
    print(mode, "epoch ms %.3f" % (dt * 1e3), "cpu enqueue ms/epoch %.3f" % (cpu * 1e3), "finite", bool(torch.isfinite(losses).all()))
