"""Graph replay vs eager launches of the epoch at --spg signals per GPU, alternating (bench.make_step on both), several rounds."""
import sys, time
sys.path.insert(0, ".")
import argparse, torch, bench
ap = argparse.ArgumentParser(); ap.add_argument("--spg", type=int, default=8); ap.add_argument("--reps", type=int, default=20); args = ap.parse_args()
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(args.spg, 0, True, dev)
gen = torch.Generator(device=dev).manual_seed(1)
steps = {m: bench.make_step(eng, x, args.spg, gen, dev, graph=m == "graph")[0] for m in ("graph", "eager")}
for rnd in range(4):
    for m in ("graph", "eager"):
        for _ in range(5): steps[m]()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.reps): steps[m]()
        torch.cuda.synchronize()
        print(rnd, m, "epoch ms %.3f" % ((time.perf_counter() - t0) / args.reps * 1e3))
