"""bench.py's drop_in section alone."""
import sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for _ in range(2):
    d = bench.bench_drop_in(True, dev)
    print("host samples us/iteration %.1f, device samples %.1f" % (d["host_samples"]["us_per_iteration"], d["device_samples"]["us_per_iteration"]))
