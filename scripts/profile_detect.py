"""Host profile of the detector on one GPU's share of configs[4] (125 000 windows): utils.anomaly_detection_utils.univariate_anomaly_detection
(Euclidean branch with DTW errors, and hyperbolic branch) -- scoring kernels + the host-side interval extraction."""
import cProfile, io, pstats, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from types import SimpleNamespace
from hypad_amd.utils import anomaly_detection_utils as adu
n, S = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000, 100
rng = np.random.default_rng(0)
t = np.arange(n + S - 1)
series = np.sin(t / 40.0) + 0.05 * rng.standard_normal(len(t)); series[n // 2: n // 2 + 300] += 1.5
true = series[np.arange(n)[:, None] + np.arange(S)[None, :]][:, :, None]
recons = (true[:, :, 0] + 0.05 * rng.standard_normal((n, S))).astype(np.float32); recons[n // 2: n // 2 + 300] *= 0.3
critic = rng.standard_normal(n).astype(np.float32)
for hyper in (False, True):
    P = SimpleNamespace(hyperbolic=hyper, signal_shape=S, save_result=False, load=False)
    rs, ts = (np.tanh(recons) * 0.5, np.tanh(true[:, :, 0]).astype(np.float32) * 0.5) if hyper else (recons, true)
    for rep in range(2):
        pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
        out = adu.univariate_anomaly_detection(rs, ts, P, "mult", list(critic), None, None, "dtw", None, None, "s", S)
        pr.disable(); dt = time.perf_counter() - t0
    print("hyperbolic" if hyper else "euclidean", "%d windows: %.1f ms, %d intervals" % (n, 1e3 * dt, len(out["intervals"])))
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(14)
    print("\n".join(l[:170] for l in st.getvalue().splitlines() if "hypad_amd" in l or "numpy" in l or "{" in l)[:3000])
