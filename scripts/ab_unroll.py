"""unroll_median_kernel timing at 125 000 windows (HYPAD_LIB_PATH selects the build)."""
import sys
import torch
sys.path.insert(0, ".")
import bench
_, _, rs = bench.bench_scoring(torch.device("cuda", 0), reps=10)
print(" ".join("%s %.3f ms" % (k.split("_kernel")[0], rs[k]["ms"]) for k in ("unroll_median_kernel", "kde_mode_kernel", "score_forward_packed_kernel")))
