"""The row-wise ball kernels at 2 000 000 rows of 100 (bench.py roofline_hbm), HIP events."""
import sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
_, hbm, _ = bench.bench_scoring(dev, cpu_sample=0)
for k, v in hbm["kernels"].items():
    print("%-36s %.1f us  %.0f GB/s (%.1f %% of 8 TB/s)" % (k, v["ms"] * 1e3, v["achieved"], 100 * v["frac"]))
