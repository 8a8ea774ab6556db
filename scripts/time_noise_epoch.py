"""Device time of one configs[1] epoch (graph replay, HIP events): device Philox draws vs injected z / alpha planes (what the drop-in
train_tadgan feeds: hypad_epoch_noise) vs host-drawn shuffles on top."""
import sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
B, S, L, NB, NC, NW = 64, 100, 20, 29, 5, 1916
eng, x = bench.build_engine(1, 0, True, dev)
ri = torch.empty(NC + 1, NB * B, dtype=torch.int32, device=dev)
eng.draw_shuffles(ri, NW)
n = NB * NC
noise = dict(z_cx=torch.randn(n, 1, B, L, device=dev), alpha_cx=torch.rand(n, 1, B, S, device=dev), z_cz=torch.randn(n, 1, B, L, device=dev),
             alpha_cz=torch.rand(n, 1, B, L, device=dev), z_gen=torch.randn(NB, 1, B, L, device=dev))
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best
print("device draws, shuffles in the graph   %.3f ms" % timed(lambda: eng.train_epoch_graph(x, ri, NB, NC, True, shuffle_windows=NW)))
print("device draws, static shuffles         %.3f ms" % timed(lambda: eng.train_epoch_graph(x, ri, NB, NC, True)))
print("injected z / alpha, static shuffles   %.3f ms" % timed(lambda: eng.train_epoch_graph(x, ri, NB, NC, True, noise=noise)))
print("status", eng.status())
