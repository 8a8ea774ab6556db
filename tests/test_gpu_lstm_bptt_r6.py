"""Back-propagation through time for the general-T bidirectional LSTM layer: hypad_lstm_bidir_seq_fwd_train +
hypad_lstm_bidir_seq_bwd through hypad_amd.autograd.lstm_seq against torch.nn.LSTM's CPU autograd -- the modules /root/reference/models/tadgan.py:15-27,
35-38 build, whose autograd covers any sequence length (the reference drives T = 1).  The twelve forward cases of tests/test_gpu_lstm_r3.py plus
T in {1, 7, 100}: gradients of the input, of the initial states and of all eight parameters, with upstream gradients on out, h_n and c_n."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [(1, 64, 100, 50), (2, 5, 20, 16), (7, 37, 50, 64), (30, 64, 100, 50), (30, 100, 128, 64), (150, 16, 5, 33), (1, 16, 128, 64), (7, 64, 100, 50), (100, 64, 128, 64),
         (100, 33, 100, 50)]


def _loss(out, hn, cn, w):
    return (out * w[0]).sum() + (hn * w[1]).sum() + (cn * w[2]).sum()


@pytest.mark.parametrize("T,rows,K,H", CASES)
@pytest.mark.parametrize("with_state", [False, True])
def test_bptt_matches_torch_autograd(T, rows, K, H, with_state):
    from hypad_amd import autograd as hag
    torch.manual_seed(T * 1000 + rows + H)
    ref = torch.nn.LSTM(input_size=K, hidden_size=H, num_layers=1, bidirectional=True)
    dev = torch.nn.LSTM(input_size=K, hidden_size=H, num_layers=1, bidirectional=True)
    dev.load_state_dict(ref.state_dict())
    dev = dev.cuda()
    x = torch.randn(T, rows, K)
    hx = (0.5 * torch.randn(2, rows, H), 0.5 * torch.randn(2, rows, H)) if with_state else None
    w = [torch.randn(T, rows, 2 * H), torch.randn(2, rows, H), torch.randn(2, rows, H)]
    xr = x.clone().requires_grad_(True)
    hr = None if hx is None else tuple(t.clone().requires_grad_(True) for t in hx)
    out, (hn, cn) = ref(xr, hr)
    _loss(out, hn, cn, w).backward()
    xd = x.cuda().requires_grad_(True)
    hd = None if hx is None else tuple(t.cuda().requires_grad_(True) for t in hx)
    got, (ghn, gcn) = hag.lstm_seq(xd, dev, 0, hd)
    tol = 2e-5 if T <= 30 else 1e-4                         # fp32 recurrences: rounding differences compound over the steps
    assert float((got.detach().cpu() - out.detach()).abs().max()) < tol
    assert float((ghn.detach().cpu() - hn.detach()).abs().max()) < tol and float((gcn.detach().cpu() - cn.detach()).abs().max()) < tol
    _loss(got, ghn, gcn, [t.cuda() for t in w]).backward()
    torch.cuda.synchronize()

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach().double()
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max())
        assert np.isfinite(err) and err < (5e-5 if T <= 30 else 4e-4) * scale, (what, err, scale)     # (sums over T x rows terms)
    close(xd.grad, xr.grad, "x")
    if with_state:
        close(hd[0].grad, hr[0].grad, "h0"); close(hd[1].grad, hr[1].grad, "c0")
    for (n, pd), (_, pr) in zip(dev.named_parameters(), ref.named_parameters()):
        close(pd.grad, pr.grad, n)


def test_bptt_upstream_gradient_on_one_output_only_and_argument_checks():
    """Only h_n carries a gradient (the encoder pattern: the last state feeds a dense layer); NULL-argument and size errors come back as codes."""
    from hypad_amd import _C
    from hypad_amd import autograd as hag
    torch.manual_seed(1)
    ref = torch.nn.LSTM(input_size=12, hidden_size=20, num_layers=1, bidirectional=True)
    dev = torch.nn.LSTM(input_size=12, hidden_size=20, num_layers=1, bidirectional=True)
    dev.load_state_dict(ref.state_dict()); dev = dev.cuda()
    x = torch.randn(9, 21, 12)
    xr = x.clone().requires_grad_(True)
    _, (hn, _) = ref(xr)
    hn.square().sum().backward()
    xd = x.cuda().requires_grad_(True)
    _, (ghn, _) = hag.lstm_seq(xd, dev)
    ghn.square().sum().backward()
    assert float((xd.grad.cpu() - xr.grad).abs().max()) < 5e-5 * max(1.0, float(xr.grad.abs().max()))
    for (n, pd), (_, pr) in zip(dev.named_parameters(), ref.named_parameters()):
        assert float((pd.grad.cpu() - pr.grad).abs().max()) < 5e-5 * max(1.0, float(pr.grad.abs().max())), n
    with pytest.raises(_C.HypadError):                       # double backward is refused, like every layer function
        xd2 = x.cuda().requires_grad_(True)
        o, _ = hag.lstm_seq(xd2, dev)
        (g,) = torch.autograd.grad(o.sum(), xd2, create_graph=True)
    f = _C.lib.hypad_lstm_bidir_seq_bwd
    p = _C.ptr(torch.zeros(4, device="cuda"))
    assert _C.lib.hypad_lstm_seq_bwd_workspace_bytes(3, 10, 7, 50) == (3 * 10 * (13 * 50 + 7) + 200 + 64) * 4
    args = [p] * 21
    assert f(*args, 1, 1, 1, 65, p, 1 << 20, _C.stream()) == -3                     # hidden > 64
    assert f(*args, 1, 1, 1, 4, None, 0, _C.stream()) == -2                         # no workspace
    bad = list(args); bad[9] = bad[10] = bad[11] = None                             # no upstream gradient at all
    assert f(*bad, 1, 1, 1, 4, p, 1 << 20, _C.stream()) == -1
