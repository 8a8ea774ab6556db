"""BASELINE.json configs[3] stand-in: rows U(-1, 1), signal_shape 150, batch 256, 20 480 windows,
hyperbolic=True, one epoch = 80 x (5 critic_x + 5 critic_z + 1 decoder) iterations.  Prints epoch-windows/s."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from hypad_amd.engine import Engine
from hypad_amd.models import tadgan

S, L, B, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 150), 20, (int(sys.argv[2]) if len(sys.argv) > 2 else 256), 20480
NB, NC = N // B, 5
dev = torch.device("cuda", 0)
eng = Engine(S, L, B, True, n_signals=1, device=dev, lr=5e-4, seed=1234)
torch.manual_seed(0)
for k, m in dict(enc=tadgan.Encoder(S, L), dec=tadgan.Decoder(S, L, True), cx=tadgan.CriticX(S, L), cz=tadgan.CriticZ(L)).items():
    eng.load_state_dict(k, m.state_dict(), 0)
x = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (1, N, S))).to(dev, torch.float32).contiguous()
gen = torch.Generator(device=dev).manual_seed(100)
losses = torch.empty(1, (2 * NC + 1) * NB, 4, device=dev)

def step():
    perm = torch.rand(NC + 1, N, device=dev, generator=gen).argsort(dim=1)[:, : NB * B]
    eng.train_epoch(x, perm.to(torch.int32).contiguous(), NB, NC, train_mode=True, losses=losses)

for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
last = losses.float().mean(dim=(0, 1)).cpu().tolist()
print(json.dumps({"workload": f"S={S} B={B} N={N} hyperbolic", "ms_per_epoch": dt * 1e3, "epoch_windows_per_s": NB * B / dt, "losses": last}))
