"""bench.py's scoring sections alone (no training, no CPU baselines): scoring value and the two sharded scorers."""
import json, sys
sys.path.insert(0, ".")
import torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
sc = bench.bench_scoring(dev, cpu_sample=0)[0]
sh = bench.bench_scoring_sharded(dev, 1, 0)
out = (json.dumps({"value": sc["value"], "without_kde": sc["without_kde_value"], "smoothing": sc["critic_smoothing_windows_per_s"],
                  "sharded_hyper": sh["hyperbolic"], "sharded_dtw": sh["euclidean_dtw"]}))
open("gpurun_out/scoring_sections.json", "a").write(out + "\n")
