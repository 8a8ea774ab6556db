"""32 models per GPU as ONE engine (grid.y = 32) vs TWO engines of 16 on two streams (each epoch a graph replay): does the hardware
overlap one group's critic phase with the other's generator phase?"""
import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
def run(groups, per, reps=10):
    engs = [bench.build_engine(per, g, True, dev) for g in range(groups)]
    gens = [torch.Generator(device=dev).manual_seed(g) for g in range(groups)]
    streams = [torch.cuda.Stream() for _ in range(groups)]
    steps = []
    for (eng, x), gen, st in zip(engs, gens, streams):
        with torch.cuda.stream(st):
            step, losses = bench.make_step(eng, x, per, gen, dev)
            step(); step()
        steps.append((step, st, losses))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            for step, st, _ in steps:
                with torch.cuda.stream(st):
                    step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    ok = all(bool(torch.isfinite(l).all()) for _, _, l in steps)
    print("%d group(s) x %d models: %.3f ms per epoch of all %d models = %.2f M windows/s  finite %s" % (groups, per, best, groups * per, groups * per * 29 * 64 / best / 1e3, ok))
run(1, 32); run(2, 16); run(4, 8); run(1, 16); run(2, 8); run(3, 8)
