"""Several small model groups (engines) per GPU -- what plan_signal_groups makes of signals of many different lengths: their epochs
one after the other on one stream, or dealt over `lanes` streams so that latency-bound groups run beside each other."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
def run(n_groups, per, lanes, reps=8):
    engs = []
    for gidx in range(n_groups):
        eng, x = bench.build_engine(per, gidx, True, dev)
        gen = torch.Generator(device=dev).manual_seed(gidx)
        engs.append((eng, x, gen))
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    steps = []
    for i, (eng, x, gen) in enumerate(engs):
        st = streams[i % lanes]
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            step, losses = bench.make_step(eng, x, per, gen, dev)
            step(); step()
        steps.append((step, st, losses, eng))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            for step, st, _, _ in steps:
                with torch.cuda.stream(st):
                    step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    ok = all(bool(torch.isfinite(l).all()) and e.status() == 0 for _, _, l, e in steps)
    print("%2d groups x %2d models, %d lane(s): %.3f ms per epoch of all %d models = %.2f M windows/s  ok %s" % (n_groups, per, lanes, best, n_groups * per, n_groups * per * 29 * 64 / best / 1e3, ok), flush=True)
for n_groups, per in ((8, 4), (8, 2), (8, 1), (4, 8)):
    for lanes in (1, 2, 4):
        if lanes * per <= 16:
            run(n_groups, per, lanes)
