import sys, json
sys.path.insert(0, ".")
import torch, bench
sc = bench.bench_scoring(torch.device("cuda", 0), cpu_sample=0)[0]
print(json.dumps({"value": sc["value"], "graph_replay_value": sc["graph_replay_value"], **sc["pcie_inclusive"]}))
