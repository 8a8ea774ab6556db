"""General-T bidirectional LSTM layer (csrc/lstm_seq.hip: one MFMA GEMM for the input projections + a persistent recurrence
kernel with W_hh in LDS) against torch.nn.LSTM on the CPU -- the module models/tadgan.py:15-20, :35-38 builds; the reference
itself only ever drives it with T = 1, where it must also agree with the T = 1 kernel of the training path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,rows,K,H", [(1, 64, 100, 50), (2, 5, 20, 16), (7, 37, 50, 64), (30, 64, 100, 50), (30, 100, 128, 64), (150, 16, 5, 33)])
@pytest.mark.parametrize("with_state", [False, True])
def test_lstm_sequence_matches_torch(T, rows, K, H, with_state):
    from hypad_amd import autograd as hag
    torch.manual_seed(T * 1000 + rows + H)
    ref = torch.nn.LSTM(input_size=K, hidden_size=H, num_layers=1, bidirectional=True)
    x = torch.randn(T, rows, K)
    hx = (0.5 * torch.randn(2, rows, H), 0.5 * torch.randn(2, rows, H)) if with_state else None
    with torch.no_grad():
        want, (hn, cn) = ref(x, hx)
        dev = ref.__class__(input_size=K, hidden_size=H, num_layers=1, bidirectional=True)
        dev.load_state_dict(ref.state_dict())
        dev = dev.cuda()
        got, (ghn, gcn) = hag.lstm_seq_forward(x.cuda(), dev, 0, None if hx is None else (hx[0].cuda(), hx[1].cuda()))
    tol = 2e-5 if T <= 30 else 1e-4                         # fp32 recurrences: rounding differences compound over the steps
    assert got.shape == want.shape
    assert float((got.cpu() - want).abs().max()) < tol
    assert float((ghn.cpu() - hn).abs().max()) < tol and float((gcn.cpu() - cn).abs().max()) < tol


def test_one_step_equals_the_training_paths_lstm_kernel():
    """At T = 1 with zero initial state W_hh cannot matter: the sequence kernel and hypad_lstm_bidir_fwd -- the layer the
    fused iterations are built from -- must agree to rounding."""
    from hypad_amd import autograd as hag
    torch.manual_seed(3)
    lstm = torch.nn.LSTM(input_size=100, hidden_size=50, num_layers=1, bidirectional=True).cuda()
    x = torch.randn(64, 100, device="cuda")
    with torch.no_grad():
        seq, _ = hag.lstm_seq_forward(x.view(1, 64, 100), lstm)
        one = hag.lstm_layer(x, lstm, 0)
    assert float((seq[0] - one).abs().max()) < 2e-6
    with pytest.raises(Exception):
        hag.lstm_seq_forward(x.view(1, 64, 100).requires_grad_(True), lstm)      # the inference form (lstm_seq is the differentiable one)


def test_lstm_sequence_entry_point_validates_arguments():
    import ctypes
    from hypad_amd import _C
    f = _C.lib.hypad_lstm_bidir_seq_fwd
    x = torch.zeros(4, device="cuda")
    p = _C.ptr(x)
    assert _C.lib.hypad_lstm_seq_workspace_bytes(3, 10, 50) == 2 * 3 * 10 * 200 * 4
    assert f(None, p, p, p, p, p, p, p, p, None, None, p, None, None, 1, 1, 1, 1, None, 0, _C.stream()) == -1
    assert f(p, p, p, p, p, p, p, p, p, None, None, p, None, None, 1, 1, 1, 65, p, 1 << 20, _C.stream()) == -3        # hidden > 64
    assert f(p, p, p, p, p, p, p, p, p, None, None, p, None, None, 1, 1, 1, 4, None, 0, _C.stream()) == -2            # no workspace


@pytest.mark.parametrize("rows,K,H", [(2048 + 37, 100, 50), (4096, 128, 64), (3000, 51, 7), (2500, 20, 64), (2049, 17, 33)])
def test_weights_stationary_layer_matches_torch_and_feeds_the_backward(rows, K, H):
    """hypad_lstm_bidir_fwd takes its weights-stationary form from 2 048 rows on (one direction's W_ih in LDS per workgroup, x rows
    straight into the MFMA A layout, the cell on the accumulators): output against torch.nn.LSTM (T = 1, zero state;
    models/tadgan.py:15-25,35-38,59-60) and the saved gates through hypad_lstm_bidir_bwd against autograd."""
    from hypad_amd import _C
    torch.manual_seed(rows + K)
    lstm = torch.nn.LSTM(K, H, 1, bidirectional=True)
    x = torch.randn(1, rows, K, requires_grad=True)
    out, _ = lstm(x)
    go = torch.randn(1, rows, 2 * H)
    names = ["weight_ih_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "bias_ih_l0_reverse", "bias_hh_l0_reverse"]
    ps = [getattr(lstm, n) for n in names]
    grads = torch.autograd.grad(out, [x] + ps, go)
    d = [p.detach().cuda().contiguous() for p in ps]
    dx = x.detach().view(rows, K).cuda().contiguous()
    o = torch.empty(rows, 2 * H, device="cuda")
    gs = torch.full((rows, 8 * H), float("nan"), device="cuda")
    _C.check(_C.lib.hypad_lstm_bidir_fwd(_C.ptr(dx), *[_C.ptr(t) for t in d], _C.ptr(o), _C.ptr(gs), rows, K, H, _C.stream()))
    assert float((o.cpu() - out.detach().view(rows, 2 * H)).abs().max()) < 1e-5
    gsv = gs.view(rows, 2, 4, H)
    assert bool(torch.isfinite(gsv).all())                                 # every (row, direction, gate, unit) was written
    o2 = torch.empty_like(o)                                               # without the saved gates
    _C.check(_C.lib.hypad_lstm_bidir_fwd(_C.ptr(dx), *[_C.ptr(t) for t in d], _C.ptr(o2), None, rows, K, H, _C.stream()))
    assert torch.equal(o, o2)
    gg = torch.empty(rows, 8 * H, device="cuda")
    gx = torch.empty(rows, K, device="cuda")
    _C.check(_C.lib.hypad_lstm_bidir_bwd(_C.ptr(d[0]), _C.ptr(d[3]), _C.ptr(gs), _C.ptr(go.view(rows, 2 * H).cuda().contiguous()), _C.ptr(gg),
                                         _C.ptr(gx), rows, K, H, _C.stream()))
    assert float((gx.cpu() - grads[0].view(rows, K)).abs().max()) < 1e-5
    ggc = gg.cpu().view(rows, 2, 4 * H)
    scale = max(1.0, float(grads[1].abs().max()))
    assert float((ggc[:, 0].t() @ x.detach().view(rows, K) - grads[1]).abs().max()) < 2e-4 * scale      # weight_ih_l0 (a sum over `rows` terms)
