"""What `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ...` is pointed at (scripts/profile_r03.sh; row d' of VERDICT r2): the kernels whose
bound is the fp32 matrix unit, each launched a few times on its BASELINE-sized input --
  score_forward_packed_kernel   125 000 windows (anomaly_detection.py:67-113 fused)
  lstm_bidir (hypad_lstm_bidir_fwd) 200 000 rows, 100 -> 2 x 50 (encoder, models/tadgan.py:15-21) and 128 -> 2 x 64 (decoder layer 1, :35-38)
  one training epoch            critic_persistent_kernel (with its producers), gen_kernel, dw_adam_kernel (configs[1])
Run directly after `--` (no wrapper process).  Prints nothing but a short summary."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hypad_amd import _C  # noqa: E402
from hypad_amd.models import tadgan  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
S, L = 100, 20
reps = int(os.environ.get("REPS", "5"))

# ---- fused scoring forward
torch.manual_seed(0)
enc, dec, cx = tadgan.Encoder(S, L).to(dev).eval(), tadgan.Decoder(S, L, True).to(dev).eval(), tadgan.CriticX(S, L).to(dev).eval()
n = 125_000
g = torch.Generator(device=dev).manual_seed(3)
x = (torch.rand(n, S, device=dev, generator=g) * 2 - 1).contiguous()
new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
for _ in range(reps):
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, _C.ptr(hyper), _C.ptr(eucl),
                                               _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist), n, S, L, 1, ws.data_ptr(), ws_bytes, _C.stream()), "score_forward")

# ---- stand-alone bidirectional LSTM layer (T = 1), the two shapes of the reference
rows = 200_000
for in_dim, hidden in ((100, 50), (128, 64)):
    lstm = torch.nn.LSTM(input_size=in_dim, hidden_size=hidden, num_layers=1, bidirectional=True).to(dev)
    xin = torch.randn(rows, in_dim, device=dev)
    out, gates = new(rows, 2 * hidden), new(rows, 8 * hidden)
    p = lambda name: _C.ptr(getattr(lstm, name).detach().contiguous())
    for _ in range(reps):
        _C.check(_C.lib.hypad_lstm_bidir_fwd(_C.ptr(xin), p("weight_ih_l0"), p("bias_ih_l0"), p("bias_hh_l0"), p("weight_ih_l0_reverse"),
                                             p("bias_ih_l0_reverse"), p("bias_hh_l0_reverse"), _C.ptr(out), _C.ptr(gates), rows, in_dim, hidden,
                                             _C.stream()), "lstm_bidir_fwd")

# ---- training epochs (eager launches: every kernel is a dispatch the profiler sees by name)
eng, xw = bench.build_engine(1, 0, True, dev)
gen = torch.Generator(device=dev).manual_seed(100)
step, losses = bench.make_step(eng, xw, 1, gen, dev, graph=False)
for _ in range(reps):
    step()
torch.cuda.synchronize()
eng.check_status()
print("profile target done:", reps, "repetitions")
