"""cProfile of the drop-in epoch loop (hypad_amd.train iteration functions, host RNG): where the host time per iteration goes."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
import bench
from hypad_amd import train as ht
from hypad_amd.models import tadgan
S, L, B = bench.S, bench.L, bench.B
dev = torch.device("cuda", 0)
P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True)
torch.manual_seed(0)
enc, dec, cx, cz = [m.to(dev) for m in (tadgan.Encoder(S, L), tadgan.Decoder(S, L, True), tadgan.CriticX(S, L), tadgan.CriticZ(L))]
opt = ht.make_optimizers(enc, dec, cx, cz, P)
data = torch.from_numpy(bench.synth_windows(bench.N_WINDOWS, S, 0)[: 29 * B, :, None])
samples = [data[b * B:(b + 1) * B].to(dev) for b in range(29)]
def epoch():
    for _ in range(5):
        for b in range(29):
            ht.critic_x_iteration(samples[b], dec, cx, opt[0], P)
            ht.critic_z_iteration(samples[b], enc, cz, opt[1], P)
    for b in range(29):
        ht.decoder_iteration(samples[b], enc, dec, cx, cz, opt[2], P)
epoch(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); epoch(); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
