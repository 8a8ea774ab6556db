"""score_forward_packed_kernel timing at 125 000 windows (HYPAD_LIB_PATH selects the build): python scripts/ab_scoring.py"""
import sys
import torch
sys.path.insert(0, ".")
import bench
_, _, rs = bench.bench_scoring(torch.device("cuda", 0), reps=10)
e = rs["score_forward_packed_kernel"]
print("score_forward_packed_kernel %.3f ms  %.1f TFLOP/s  frac %.3f" % (e["ms"], e["achieved"], e["frac"]))
