"""Round 5: the fused scoring forward's two tile forms and the critic launch beside it (anomaly_detection.py:67-113).

hypad_score_forward_packed runs 32 windows per workgroup from 65 536 windows on (16 below) and takes the critic value of the windows from
critic_rows_kernel.  The reference-pinned checks are on the 16-window form (tests/test_gpu_parity.py: fixtures and the streamed kernel);
here: the 32-window form gives the same bits as the 16-window form, row for row, ragged tail and series view included, and the critic
launch agrees with hypad_critic_x_fwd (the fixture-checked entry point)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets(hyper):
    from hypad_amd.models import tadgan
    torch.manual_seed(5)
    S, L = 100, 20
    enc, dec, cx = tadgan.Encoder(S, L).cuda().eval(), tadgan.Decoder(S, L, hyper).cuda().eval(), tadgan.CriticX(S, L).cuda().eval()
    if hyper:
        with torch.no_grad():
            dec.hyperbolic_linear.weight.mul_(30)
    return S, L, enc, dec, cx, hyper


@pytest.fixture(scope="module")
def nets():
    return _nets(True)


def _forward(nets, src, stride, n, outs):
    from hypad_amd import _C
    S, L, enc, dec, cx, hyper = nets
    if not hyper:                        # (Euclidean decoder: only the reconstruction and the critic value exist)
        outs = [None, outs[1], None, outs[3], None]
    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, int(hyper))
    ws = torch.empty(ws_bytes // 4, device="cuda")
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(src), stride, *[_C.ptr(o) for o in outs],
                                               n, S, L, int(hyper), ws.data_ptr(), ws_bytes, _C.stream()), "score_forward_packed")
    torch.cuda.synchronize()


@pytest.mark.parametrize("view,hyper", [("rows", True), ("series", True), ("rows", False)])
def test_32_window_form_equals_16_window_form(view, hyper):
    nets = _nets(hyper)
    S = nets[0]
    n = 65_536 + 16 + 5                 # the 32-window form, last workgroup: 21 valid rows of 32
    series = (torch.rand(n + S - 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 2 - 1).contiguous()
    x = series.unfold(0, S, 1).contiguous()
    new = lambda *s: torch.full(s, float("nan"), device="cuda")
    big = [new(n, S), new(n, S), new(n, S), new(n), new(n)]
    _forward(nets, series if view == "series" else x, 1 if view == "series" else 0, n, big)
    small = [new(n, S), new(n, S), new(n, S), new(n), new(n)]
    for lo in range(0, n, 30_000):      # every piece below the switch: the 16-window form
        m = min(30_000, n - lo)
        piece = [new(m, S), new(m, S), new(m, S), new(m), new(m)]
        _forward(nets, x[lo:lo + m].contiguous(), 0, m, piece)
        for dst, p in zip(small, piece):
            dst[lo:lo + m] = p
    for k, (a, b) in enumerate(zip(big, small)):
        if not hyper and k in (0, 2, 4):
            continue
        assert bool(torch.isfinite(a).all()), k
        assert torch.equal(a, b), (k, float((a - b).abs().max()))


def test_critic_launch_matches_the_entry_point(nets):
    from hypad_amd import _C
    S, L, enc, dec, cx, _ = nets
    for n in (1, 37, 128 * 16 + 3, 70_001):
        x = (torch.rand(n, S, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n)) * 2 - 1).contiguous()
        got, dist = torch.full((n,), float("nan"), device="cuda"), torch.empty(n, device="cuda")
        _forward(nets, x, 0, n, [None, None, None, got, dist])
        ref = torch.empty(n, device="cuda")
        _C.check(_C.lib.hypad_critic_x_fwd(_C.ptr(cx.arena()), _C.ptr(x), _C.ptr(ref), n, S, L, None, _C.stream()), "critic_x_fwd")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(got).all()), n
        assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (n, float((got - ref).abs().max()))


@pytest.mark.parametrize("S,L", [(256, 32), (123, 20), (51, 7)])
def test_critic_launch_other_windows(S, L):
    """The run-time-shape build of the critic launch (rows streamed, no register prefetch), incl. the widest window the library takes --
    whose wave-private tiles leave room for fewer waves per workgroup."""
    from hypad_amd import _C
    from hypad_amd.models import tadgan
    torch.manual_seed(S)
    enc, dec, cx = tadgan.Encoder(S, L).cuda().eval(), tadgan.Decoder(S, L, True).cuda().eval(), tadgan.CriticX(S, L).cuda().eval()
    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
    ws = torch.empty(ws_bytes // 4, device="cuda")
    for n in (5, 16 * 9 + 1, 3000):
        x = (torch.rand(n, S, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n)) * 2 - 1).contiguous()
        got, dist, ref = torch.full((n,), float("nan"), device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
        _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, None, None, None, _C.ptr(got), _C.ptr(dist),
                                                   n, S, L, 1, ws.data_ptr(), ws_bytes, _C.stream()), "score_forward_packed")
        _C.check(_C.lib.hypad_critic_x_fwd(_C.ptr(cx.arena()), _C.ptr(x), _C.ptr(ref), n, S, L, None, _C.stream()), "critic_x_fwd")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(got).all()) and bool(torch.isfinite(dist).all()), (S, n)
        assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())), (S, n, float((got - ref).abs().max()))
