"""The stand-alone LSTM layer timings of bench.py's roofline_lstm section alone (A/B runs: scripts/ab_variants.sh)."""
import sys
sys.path.insert(0, ".")
import torch, bench
r = bench.bench_lstm_layers(torch.device("cuda", 0))
print(" | ".join("%s %.1f us" % (k, v["us"]) for k, v in r["layers"].items()))
