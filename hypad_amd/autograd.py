"""Differentiable forwards of the TadGAN networks, for callers that train OUTSIDE the three fused iteration functions
(``loss.backward(); optimizer.step()`` with their own loss, as any reference user may: models/tadgan.py:23-27,58-67,91-106,123-132).

Each layer is a ``torch.autograd.Function`` over the library's forward / backward building blocks (include/hypad.h:
``hypad_linear_act_{fwd,bwd}``, ``hypad_lstm_bidir_{fwd,bwd}``, ``hypad_mobius_linear_{fwd,bwd}``), so the graph's arithmetic
is the HIP kernels'; torch only chains them (and draws the dropout masks of train mode).  The fused iteration functions
of ``hypad_amd.train`` remain the fast path: one launch group per iteration instead of ~20 layer launches.

First-order only: every backward is guarded by ``_C.first_order_only`` -- a reference-style gradient penalty taken through
these forwards with ``torch.autograd.grad(..., create_graph=True)`` (train.py:72-93) raises instead of silently returning a
penalty whose second-order gradient is zero.  The WGAN-GP iterations, second-order chain included, are
``hypad_amd.train.critic_{x,z}_iteration``.
"""
import torch

from . import _C


def _f32c(t, name):
    return _C.require_cuda(t.to(torch.float32).contiguous(), name)


class _LinearAct(torch.autograd.Function):
    """y = act(x W^T + b): nn.Linear + nn.Tanh / nn.LeakyReLU(0.2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, w, b = _f32c(x, "input"), _f32c(weight, "weight"), _f32c(bias, "bias")
        rows, k, n = x.shape[0], x.shape[1], w.shape[0]
        y = torch.empty(rows, n, device=x.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_linear_act_fwd(_C.ptr(x), _C.ptr(w), _C.ptr(b), _C.ptr(y), rows, k, n, int(act), _C.stream()), "linear_act_fwd")
        ctx.save_for_backward(x, w, y)
        ctx.act = int(act)
        return y

    @staticmethod
    @_C.first_order_only
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        rows, k, n = x.shape[0], x.shape[1], w.shape[0]
        gy = _f32c(gy, "grad")
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        gb = torch.empty(n, device=x.device, dtype=torch.float32)
        scratch = torch.empty(rows, n, device=x.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_linear_act_bwd(_C.ptr(x), _C.ptr(w), _C.ptr(y), _C.ptr(gy), _C.ptr(gx), _C.ptr(gw), _C.ptr(gb), _C.ptr(scratch),
                                             rows, k, n, ctx.act, _C.stream()), "linear_act_bwd")
        return gx, gw, gb, None


class _LstmBidirT1(torch.autograd.Function):
    """One bidirectional LSTM layer at sequence length 1 with h0 = c0 = 0: out = [h_fwd | h_rev].
    weight_hh receives an all-zero gradient, exactly as autograd gives the reference."""

    @staticmethod
    def forward(ctx, x, wf, whf, bif, bhf, wr, whr, bir, bhr):
        x = _f32c(x, "input")
        ps = [_f32c(t, "lstm parameter") for t in (wf, bif, bhf, wr, bir, bhr)]
        rows, k, h = x.shape[0], x.shape[1], wf.shape[0] // 4
        out = torch.empty(rows, 2 * h, device=x.device, dtype=torch.float32)
        gates = torch.empty(rows, 8 * h, device=x.device, dtype=torch.float32)
        _C.check(_C.lib.hypad_lstm_bidir_fwd(_C.ptr(x), *[_C.ptr(t) for t in ps], _C.ptr(out), _C.ptr(gates), rows, k, h, _C.stream()),
                 "lstm_bidir_fwd")
        ctx.save_for_backward(x, ps[0], ps[3], gates)
        ctx.hh = (whf.shape, whr.shape)
        return out

    @staticmethod
    @_C.first_order_only
    def backward(ctx, go):
        x, wf, wr, gates = ctx.saved_tensors
        rows, k, h = x.shape[0], x.shape[1], wf.shape[0] // 4
        go = _f32c(go, "grad")
        gg = torch.empty(rows, 8 * h, device=x.device, dtype=torch.float32)        # (rows, 2, 4h) pre-activation gradients
        gx = torch.empty_like(x)
        _C.check(_C.lib.hypad_lstm_bidir_bwd(_C.ptr(wf), _C.ptr(wr), _C.ptr(gates), _C.ptr(go), _C.ptr(gg), _C.ptr(gx), rows, k, h, _C.stream()),
                 "lstm_bidir_bwd")
        grads = []
        gg3 = gg.view(rows, 2, 4 * h)
        for d, w in enumerate((wf, wr)):
            # parameter gradients of one direction = those of a bias-carrying linear layer fed by x whose output gradient is
            # the direction's gate gradient (hypad_linear_act_bwd with no activation); its input gradient is discarded
            ggd = gg3[:, d].contiguous()
            gw, gb = torch.empty_like(w), torch.empty(4 * h, device=x.device, dtype=torch.float32)
            dump, scratch = torch.empty_like(x), torch.empty(rows, 4 * h, device=x.device, dtype=torch.float32)
            _C.check(_C.lib.hypad_linear_act_bwd(_C.ptr(x), _C.ptr(w), _C.ptr(ggd), _C.ptr(ggd), _C.ptr(dump), _C.ptr(gw), _C.ptr(gb),
                                                 _C.ptr(scratch), rows, k, 4 * h, _C.ACT_NONE, _C.stream()), "lstm weight gradient")
            grads.append((gw, gb))
        zf = torch.zeros(ctx.hh[0], device=x.device, dtype=torch.float32)
        zr = torch.zeros(ctx.hh[1], device=x.device, dtype=torch.float32)
        (gwf, gbf), (gwr, gbr) = grads
        return gx, gwf, zf, gbf, gbf.clone(), gwr, zr, gbr, gbr.clone()


def linear_act(x, weight, bias, act=_C.ACT_NONE):
    return _LinearAct.apply(x, weight, bias, act)


def lstm_layer(x, group, layer):
    g = lambda n: getattr(group, n)
    sfx = f"_l{layer}"
    return _LstmBidirT1.apply(x, g("weight_ih" + sfx), g("weight_hh" + sfx), g("bias_ih" + sfx), g("bias_hh" + sfx),
                              g("weight_ih" + sfx + "_reverse"), g("weight_hh" + sfx + "_reverse"), g("bias_ih" + sfx + "_reverse"),
                              g("bias_hh" + sfx + "_reverse"))


def wants_graph(module, *inputs):
    """True when the caller can differentiate through this forward: autograd is recording and an input or a parameter of
    the module requires a gradient."""
    if not torch.is_grad_enabled():
        return False
    return any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs) or any(p.requires_grad for p in module.parameters())


def encoder_forward(m, x):                                   # models/tadgan.py:23-27
    h = lstm_layer(x, m.lstm, 0)
    return linear_act(h, m.dense.weight, m.dense.bias)


def decoder_forward(m, z):                                   # models/tadgan.py:58-67
    a = linear_act(z, m.dense1.weight, m.dense1.bias)
    h0 = lstm_layer(a, m.lstm, 0)
    h0 = torch.nn.functional.dropout(h0, p=0.2, training=m.training)          # nn.LSTM(dropout=0.2): between the two layers
    h1 = lstm_layer(h0, m.lstm, 1)
    e = linear_act(h1, m.dense2.weight, m.dense2.bias, _C.ACT_TANH)
    if m.hyperbolic:
        return m.hyperbolic_linear(e), e
    return e


def critic_forward(m, x, n_hidden, p_drop):                  # models/tadgan.py:91-106, :123-132
    h = x
    for i in range(1, n_hidden + 1):
        d = getattr(m, f"dense{i}")
        h = linear_act(h, d.weight, d.bias, _C.ACT_LEAKY02)
        h = torch.nn.functional.dropout(h, p=p_drop, training=m.training)
    d = getattr(m, f"dense{n_hidden + 1}")
    return linear_act(h, d.weight, d.bias)


class _LstmBidirSeq(torch.autograd.Function):
    """One bidirectional LSTM layer over a whole sequence with back-propagation through time (csrc/lstm_seq.hip: hypad_lstm_bidir_seq_fwd_train /
    hypad_lstm_bidir_seq_bwd) -- what autograd gives ``nn.LSTM(in, H, bidirectional=True)`` (models/tadgan.py:15-27, 35-38) at any T.
    Inputs: x (T, rows, in), h0 / c0 (2, rows, H) or None, then the eight parameters in nn.LSTM's order
    (weight_ih, weight_hh, bias_ih, bias_hh of the forward direction, then of the reverse one).  Returns out, h_n, c_n."""

    @staticmethod
    def forward(ctx, x, h0, c0, wif, whf, bif, bhf, wir, whr, bir, bhr):
        x = _f32c(x, "input")
        ps = [_f32c(t, "lstm parameter") for t in (wif, whf, bif, bhf, wir, whr, bir, bhr)]
        T, rows, k = x.shape
        H = ps[1].shape[1]
        h0c = None if h0 is None else _f32c(h0, "h0")
        c0c = None if c0 is None else _f32c(c0, "c0")
        new = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
        out, hn, cn, saved = new(T, rows, 2 * H), new(2, rows, H), new(2, rows, H), new(T, rows, 2, 5, H)
        nbytes = _C.lib.hypad_lstm_seq_workspace_bytes(T, rows, H)
        ws = new(max(nbytes // 4, 1))
        _C.check(_C.lib.hypad_lstm_bidir_seq_fwd_train(_C.ptr(x), *[_C.ptr(p) for p in ps], _C.ptr(h0c), _C.ptr(c0c), _C.ptr(out), _C.ptr(hn), _C.ptr(cn),
                                                       _C.ptr(saved), T, rows, k, H, ws.data_ptr(), nbytes, _C.stream()), "lstm_bidir_seq_fwd_train")
        ctx.save_for_backward(x, ps[0], ps[1], ps[4], ps[5], out, saved, *(t for t in (h0c, c0c) if t is not None))
        ctx.has_h0, ctx.has_c0 = h0c is not None, c0c is not None
        return out, hn, cn

    @staticmethod
    @_C.first_order_only
    def backward(ctx, g_out, g_hn, g_cn):
        x, wif, whf, wir, whr, out, saved, *rest = ctx.saved_tensors
        h0 = rest.pop(0) if ctx.has_h0 else None
        c0 = rest.pop(0) if ctx.has_c0 else None
        T, rows, k = x.shape
        H = whf.shape[1]
        new = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
        g_out, g_hn, g_cn = (None if g is None else _f32c(g, "grad") for g in (g_out, g_hn, g_cn))
        if g_out is None and g_hn is None and g_cn is None:
            return (None,) * 11
        gx, gwif, gwhf, gbf, gwir, gwhr, gbr = new(T, rows, k), new(4 * H, k), new(4 * H, H), new(4 * H), new(4 * H, k), new(4 * H, H), new(4 * H)
        gh0, gc0 = (new(2, rows, H) if ctx.has_h0 else None), (new(2, rows, H) if ctx.has_c0 else None)
        nbytes = _C.lib.hypad_lstm_seq_bwd_workspace_bytes(T, rows, k, H)
        ws = new(max(nbytes // 4, 1))
        _C.check(_C.lib.hypad_lstm_bidir_seq_bwd(_C.ptr(x), _C.ptr(wif), _C.ptr(whf), _C.ptr(wir), _C.ptr(whr), _C.ptr(h0), _C.ptr(c0), _C.ptr(out), _C.ptr(saved),
                                                 _C.ptr(g_out), _C.ptr(g_hn), _C.ptr(g_cn), _C.ptr(gx), _C.ptr(gwif), _C.ptr(gwhf), _C.ptr(gbf), _C.ptr(gwir),
                                                 _C.ptr(gwhr), _C.ptr(gbr), _C.ptr(gh0), _C.ptr(gc0), T, rows, k, H, ws.data_ptr(), nbytes, _C.stream()),
                 "lstm_bidir_seq_bwd")
        return gx, gh0, gc0, gwif, gwhf, gbf, gbf.clone(), gwir, gwhr, gbr, gbr.clone()


def lstm_seq(x, lstm, layer=0, hx=None):
    """The differentiable form of lstm_seq_forward: ``(out, (h_n, c_n))`` of one bidirectional layer of a ``torch.nn.LSTM``-shaped parameter holder
    over (T, rows, in), with gradients for the input, the initial states and all eight parameters (back-propagation through time on the device)."""
    names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
    ps = [getattr(lstm, n + f"_l{layer}") for n in names] + [getattr(lstm, n + f"_l{layer}_reverse") for n in names]
    h0, c0 = (None, None) if hx is None else hx
    out, hn, cn = _LstmBidirSeq.apply(x, h0, c0, *ps)
    return out, (hn, cn)


def lstm_seq_forward(x, lstm, layer=0, hx=None):
    """One bidirectional layer of a ``torch.nn.LSTM``-shaped parameter holder over a whole sequence (inference): ``x`` (T, rows, in)
    -> ``(out (T, rows, 2H), (h_n, c_n) each (2, rows, H))`` as ``nn.LSTM(in, H, bidirectional=True)(x, hx)`` returns them.  The
    reference's modules (models/tadgan.py:15-20, :35-38) are such layers driven with T = 1; this is the general-T
    form: one MFMA GEMM for the input projections of all steps + a persistent recurrence kernel (csrc/lstm_seq.hip)."""
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in lstm.parameters())):
        raise _C.HypadError("lstm_seq_forward is the inference form: wrap the call in torch.no_grad(), or use lstm_seq (back-propagation through time)")
    x = _f32c(x, "input")
    T, rows, k = x.shape
    g = lambda n: _f32c(getattr(lstm, n + f"_l{layer}").detach(), n)
    r = lambda n: _f32c(getattr(lstm, n + f"_l{layer}_reverse").detach(), n)
    H = g("weight_hh").shape[1]
    out = torch.empty(T, rows, 2 * H, device=x.device, dtype=torch.float32)
    hn, cn = torch.empty(2, rows, H, device=x.device), torch.empty(2, rows, H, device=x.device)
    h0 = c0 = None
    if hx is not None:
        h0, c0 = _f32c(hx[0], "h0"), _f32c(hx[1], "c0")
    nbytes = _C.lib.hypad_lstm_seq_workspace_bytes(T, rows, H)
    ws = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
    _C.check(_C.lib.hypad_lstm_bidir_seq_fwd(_C.ptr(x), _C.ptr(g("weight_ih")), _C.ptr(g("weight_hh")), _C.ptr(g("bias_ih")), _C.ptr(g("bias_hh")),
                                             _C.ptr(r("weight_ih")), _C.ptr(r("weight_hh")), _C.ptr(r("bias_ih")), _C.ptr(r("bias_hh")),
                                             _C.ptr(h0), _C.ptr(c0), _C.ptr(out), _C.ptr(hn), _C.ptr(cn), T, rows, k, H, ws.data_ptr(), nbytes,
                                             _C.stream()), "lstm_bidir_seq_fwd")
    return out, (hn, cn)
