"""GPU time of the per-iteration entry points' kernels (hypad_profile_iteration) at configs[1]."""
import sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
eng, x = bench.build_engine(1, 0, True, dev)
for kind, name in ((0, "critic_x"), (1, "critic_z"), (2, "decoder"), (3, "critic pair")):
    for _ in range(3): ms = eng.profile_iteration(kind, x[:, :64])
    print(name, ["%.1f us" % (1e3 * v) for v in ms], "sum %.1f us" % (1e3 * sum(ms)))
