"""dW + Adam / generator kernel launch times at --spg signals per GPU (HIP events, hypad_profile_iteration kind 2)."""
import sys
sys.path.insert(0, ".")
import argparse, numpy as np, torch, bench
ap = argparse.ArgumentParser(); ap.add_argument("--spg", type=int, default=32); args = ap.parse_args()
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(args.spg, 0, True, dev)
idx = torch.arange(bench.B, device=dev, dtype=torch.int32)
ms = [eng.profile_iteration(2, x, idx, True) for _ in range(40)][10:]
print("spg", args.spg, "gen us %.1f" % (1e3 * np.mean([m[0] for m in ms])), "dw us %.1f" % (1e3 * np.mean([m[1] for m in ms])))
