"""Epoch time (graph replay) at several models per GPU, with and without hypad_epoch_io.enc_table (encoder(x) once per window row)."""
import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
counts = [int(a) for a in sys.argv[1:]] or [8, 16, 32]
for spg in counts:
    row = []
    for table in (False, True):
        eng, x = bench.build_engine(spg, 0, True, dev)
        eng.enc_table = table
        gen = torch.Generator(device=dev).manual_seed(1)
        step, losses = bench.make_step(eng, x, spg, gen, dev)
        for _ in range(3): step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10): step()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        row.append(best)
        assert eng.status() == 0
    print("%d signals: %.3f ms per epoch, with the encoder table %.3f" % (spg, row[0], row[1]), flush=True)
