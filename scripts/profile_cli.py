"""Where a whole `hypad_amd.main.run` (CSV -> dataset -> training -> test loop -> scoring -> intervals) spends its time, by stage (cProfile,
top cumulative entries), on the 2 016-sample NAB-style fixture: 30 epochs, hyperbolic and Euclidean."""
import cProfile, io, os, pstats, sys, tempfile, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
from types import SimpleNamespace
from hypad_amd import main as hmain
fxd = np.load(os.path.join(root, "tests", "golden", "dataloader.npz"), allow_pickle=True)
for hyper in (True, False):
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "sig.csv"), "w") as f:
            f.write(str(fxd["dl_nab600_csv"]))
        ts = fxd["dl_nab600_index"]
        with open(os.path.join(d, "anomalies.csv"), "w") as f:
            f.write('signal,events\nsig,"[[%d, %d]]"\n' % (ts[200], ts[260]))
        os.chdir(d)
        torch.manual_seed(5)
        P = SimpleNamespace(dataset="NAB", signal="sig", epochs=30, hyperbolic=hyper, signal_shape=100, lr=5e-4, batch_size=64, save_result=False, filename="",
                            rec_error="dtw", combination="mult", interval=600, unique_dataset=True, resume=False, resume_epoch=0, load=False)
        for rep in range(2):                       # (the second run: libraries loaded, kernels resident)
            pr = cProfile.Profile()
            t0 = time.perf_counter()
            pr.enable()
            sys.stdout = io.StringIO()
            try:
                out = hmain.run(P, None, d, log=lambda s: None)
            finally:
                sys.stdout = sys.__stdout__
            pr.disable()
            dt = time.perf_counter() - t0
        print("hyperbolic" if hyper else "euclidean", "whole run %.1f ms (30 epochs of training ~ %.0f ms of launches)" % (1e3 * dt, 30 * 2.85))
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45)
        keep = [l for l in st.getvalue().splitlines() if "hypad_amd" in l or "pandas" in l.lower() or "torch/serialization" in l or "scipy" in l]
        print("\n".join(l[:190] for l in keep[:40]))
        os.chdir("/tmp")
