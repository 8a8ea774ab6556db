"""In-process timing of hypad_amd.host_rng: the epoch's latent draws of configs[1] (371 200 + 37 120 values) and configs[3] (4 096 000 + 409 600)
into pageable and page-locked arrays, with 0 / 1 / 3 helper threads; torch.rand natively vs torch."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from hypad_amd import host_rng
def timed(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * min(ts)
for name, chunk, rounds in (("configs[1]", 1280, 145), ("configs[3]", 5120, 400)):
    for kind in ("pageable", "pinned"):
        outs = [torch.empty(chunk * rounds, pin_memory=(kind == "pinned")).numpy() for _ in range(2)]
        for th in (0, 1, 3, 6):
            old = host_rng.PIPELINE_FROM
            host_rng.PIPELINE_FROM = 1
            try:
                ms = timed(lambda: host_rng.global_normal_into(outs, chunk, rounds, threads=th))
            finally:
                host_rng.PIPELINE_FROM = old
            print("%s %-8s %d helper thread(s): %.2f ms for %d values = %.2f ns each" % (name, kind, th, ms, 2 * chunk * rounds, 1e6 * ms / (2 * chunk * rounds)), flush=True)
buf = torch.empty(29 * (256 * 150 + 256 * 20) * 3)
print("torch.rand(%d): %.2f ms; natively %.2f ms" % (buf.numel(), timed(lambda: torch.rand(buf.shape, out=buf)), timed(lambda: host_rng.torch_rand_into(buf))))
host_rng.PIPELINE_FROM = 1
tiny = [np.empty(64, np.float32)]
for th in (0, 1, 3, 6):
    print("64 values, %d helper thread(s): %.3f ms (thread start + join)" % (th, timed(lambda: host_rng.global_normal_into(tiny, 64, 1, threads=th), reps=20)))
