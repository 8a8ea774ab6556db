"""A/B timing of hypad_train_epoch: hoisted critic phase vs per-minibatch launch groups (configs[1] shape)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench

def main():
    dev = torch.device("cuda", 0)
    eng, x = bench.build_engine(1, 0, True, dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    nb, nc, B = bench.N_BATCHES, bench.N_CRITICS, bench.B
    for hoist in (False, True, False, True):
        ts = []
        for rep in range(6):
            perm = torch.stack([torch.randperm(bench.N_WINDOWS, device=dev, generator=gen)[: nb * B] for _ in range(nc + 1)]).to(torch.int32).contiguous()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            l = eng.train_epoch(x, perm, nb, nc, True, hoist=hoist)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            print("hoist", hoist, "rep", rep, "ms %.2f" % ts[-1], "loss", l[0, :2, 0].tolist(), flush=True)
    for kind in (4, 3, 2):
        idx = torch.arange(B, device=dev, dtype=torch.int32)
        ms = [eng.profile_iteration(kind, x, idx, True) for _ in range(30)][10:]
        print("kind", kind, [sum(m[i] for m in ms) / len(ms) * 1e3 for i in range(len(ms[0]))], "us", flush=True)

main()
