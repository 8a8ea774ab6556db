"""hypad_lstm_bidir_fwd (T = 1) at 200 000 rows, the reference's two layer shapes: launch time (HIP events), MFMA FLOP rate of the three gate
products it issues (2 x 3 x H x K MAC per row and direction) against the fp32 matrix peak.  HYPAD_LSTM_LDS=0: the streamed-weights form."""
import os, sys
sys.path.insert(0, ".")
import torch
from hypad_amd import _C
rows = 200_000
torch.manual_seed(0)
for in_dim, hidden in ((100, 50), (128, 64)):
    lstm = torch.nn.LSTM(input_size=in_dim, hidden_size=hidden, num_layers=1, bidirectional=True).cuda()
    x = torch.randn(rows, in_dim, device="cuda")
    out, gates = torch.empty(rows, 2 * hidden, device="cuda"), torch.zeros(rows, 8 * hidden, device="cuda")
    p = lambda n: _C.ptr(getattr(lstm, n).detach().contiguous())
    ps = [p(n) for n in ("weight_ih_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse")]
    for gs in (gates, None):
        fn = lambda: _C.check(_C.lib.hypad_lstm_bidir_fwd(_C.ptr(x), *ps, _C.ptr(out), _C.ptr(gs) if gs is not None else None, rows, in_dim, hidden, _C.stream()), "lstm")
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        flop = 2.0 * 2 * 3 * hidden * in_dim * rows
        print("LDS form" if os.environ.get("HYPAD_LSTM_LDS", "1") != "0" else "streamed", "%d -> 2 x %d" % (in_dim, hidden), "gates saved" if gs is not None else "no gates",
              "%.1f us  %.1f TFLOP/s = %.1f %% of 157.3" % (us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100), "checksum %.4f" % float(out.double().sum()), ("gates %.4f" % float(gs[:, : 8 * hidden].double().sum())) if gs is not None else "")
