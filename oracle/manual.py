"""Closed-form forward / hand-derived backward of the whole hot path (oracle; test infrastructure).

This file is the *derivation sheet* the HIP kernels transcribe: every formula
here is explicit (matmuls + elementwise, no autograd) and is checked on CPU
against autograd of the reference-pinned oracle (tests/test_manual_derivations.py)
before a single kernel is trusted.  It also accepts injected dropout masks, which
the nn.Module oracle cannot, so train-mode parity of the HIP path is checked
against this file.

Conventions: all tensors are 2-D ``(rows, features)``; weights keep PyTorch's
``(out, in)`` layout and ``state_dict`` names; LSTM gate blocks
are ``[i, f, g, o]``.  With T=1 and h0=c0=0 the f gate and
``W_hh`` never influence any output and receive exactly zero data gradient, so
they are not evaluated.
"""
import torch

LEAK = 0.2
MAXNORM_F32 = 1.0 - 4e-3


# ------------------------------------------------------------------------------------ LSTM (T = 1)
def lstm_dir_fwd(a, w_ih, b_ih, b_hh):
    """One direction of one layer, closed form.  Returns h and what backward needs."""
    H = w_ih.shape[0] // 4
    G = a @ w_ih.t() + b_ih + b_hh
    i, g, o = torch.sigmoid(G[:, :H]), torch.tanh(G[:, 2 * H:3 * H]), torch.sigmoid(G[:, 3 * H:])
    tc = torch.tanh(i * g)
    return o * tc, (i, g, o, tc)


def lstm_dir_bwd(dh, saved):
    """d(pre-activations) as (rows, 4H) with the f block identically zero."""
    i, g, o, tc = saved
    do = dh * tc * o * (1 - o)
    dc = dh * o * (1 - tc * tc)
    di = dc * g * i * (1 - i)
    dg = dc * i * (1 - g * g)
    return torch.cat([di, torch.zeros_like(di), dg, do], dim=1)


def bilstm_layer_fwd(a, sd, prefix, layer):
    hf, sf = lstm_dir_fwd(a, sd[f"{prefix}weight_ih_l{layer}"], sd[f"{prefix}bias_ih_l{layer}"], sd[f"{prefix}bias_hh_l{layer}"])
    hr, sr = lstm_dir_fwd(a, sd[f"{prefix}weight_ih_l{layer}_reverse"], sd[f"{prefix}bias_ih_l{layer}_reverse"],
                          sd[f"{prefix}bias_hh_l{layer}_reverse"])
    return torch.cat([hf, hr], dim=1), (sf, sr)


def bilstm_layer_bwd(dh, a, saved, sd, prefix, layer, grads):
    """Accumulates parameter grads into ``grads`` and returns d(a)."""
    H = dh.shape[1] // 2
    da = torch.zeros_like(a)
    for d, (sl, sfx) in enumerate(((slice(0, H), ""), (slice(H, 2 * H), "_reverse"))):
        dG = lstm_dir_bwd(dh[:, sl], saved[d])
        w = sd[f"{prefix}weight_ih_l{layer}{sfx}"]
        _acc(grads, f"{prefix}weight_ih_l{layer}{sfx}", dG.t() @ a)
        _acc(grads, f"{prefix}bias_ih_l{layer}{sfx}", dG.sum(0))
        _acc(grads, f"{prefix}bias_hh_l{layer}{sfx}", dG.sum(0))
        da = da + dG @ w
    return da


def _acc(grads, key, val):
    grads[key] = grads[key] + val if key in grads else val


# ------------------------------------------------------------------------------------ Moebius head
def head_epilogue_fwd(u, b, maxnorm=MAXNORM_F32):
    """expmap0 -> mobius_add(bias) -> project on rows of ``u`` (= e @ W_h^T).  SURVEY.md A.2."""
    n = u.norm(dim=1, keepdim=True).clamp_min(1e-15)
    t = torch.tanh(n.clamp(max=15.0))
    p = t * u / n
    x2, y2, xy = (p * p).sum(1, keepdim=True), (b * b).sum(), (p * b).sum(1, keepdim=True)
    A, Bc = 1 + 2 * xy + y2, 1 - x2
    D = (1 + 2 * xy + x2 * y2).clamp_min(1e-15)
    m = (A * p + Bc * b) / D
    nm = m.norm(dim=1, keepdim=True).clamp_min(1e-15)
    return torch.where(nm > maxnorm, m / nm * maxnorm, m)


def head_epilogue_bwd(u, b, dr, maxnorm=MAXNORM_F32):
    """Returns (du, db_rows) with db_rows the per-row bias-gradient contributions."""
    raw = u.norm(dim=1, keepdim=True)
    n = raw.clamp_min(1e-15)
    nc = n.clamp(max=15.0)
    t = torch.tanh(nc)
    f = t / n
    p = f * u
    x2, y2, xy = (p * p).sum(1, keepdim=True), (b * b).sum(), (p * b).sum(1, keepdim=True)
    A, Bc = 1 + 2 * xy + y2, 1 - x2
    Draw = 1 + 2 * xy + x2 * y2
    D = Draw.clamp_min(1e-15)
    m = (A * p + Bc * b) / D
    mraw = m.norm(dim=1, keepdim=True)
    nm = mraw.clamp_min(1e-15)
    clipped = nm > maxnorm
    # project
    dm_clip = (maxnorm / nm) * (dr - m * (m * dr).sum(1, keepdim=True) / (nm * nm) * (mraw >= 1e-15))
    dm = torch.where(clipped, dm_clip, dr)
    # mobius_add
    dN = dm / D
    dD = -(dm * m).sum(1, keepdim=True) / D * (Draw >= 1e-15)
    dA, dBc = (dN * p).sum(1, keepdim=True), (dN * b).sum(1, keepdim=True)
    dxy = 2 * dA + 2 * dD
    dx2 = -dBc + y2 * dD
    dy2 = dA + x2 * dD
    dp = A * dN + 2 * dx2 * p + dxy * b
    db_rows = Bc * dN + 2 * dy2 * b + dxy * p
    # expmap0
    tprime = (1 - t * t) * (n <= 15.0)
    dfdn = (tprime * n - t) / (n * n)
    du = f * dp + (dp * u).sum(1, keepdim=True) * dfdn * (u / n) * (raw >= 1e-15)
    return du, db_rows


# ------------------------------------------------------------------------------------ row distance
def rowdist_fwd(u, v):
    sq = ((u - v) ** 2).sum(1)
    un, vn = (u * u).sum(1), (v * v).sum(1)
    return torch.acosh(1 + 2 * sq / ((1 - un) * (1 - vn)) + 1e-7)


def rowdist_bwd(u, v, gd):
    """gd: (rows,) upstream gradient of the distances."""
    diff = u - v
    sq, un, vn = (diff * diff).sum(1), (u * u).sum(1), (v * v).sum(1)
    den = (1 - un) * (1 - vn)
    xt = 1 + 2 * sq / den + 1e-7
    gx = gd / torch.sqrt(xt * xt - 1)
    dsq = (gx * 2 / den).unsqueeze(1)
    dden = -gx * 2 * sq / (den * den)
    dun, dvn = (-dden * (1 - vn)).unsqueeze(1), (-dden * (1 - un)).unsqueeze(1)
    return dsq * 2 * diff + dun * 2 * u, -dsq * 2 * diff + dvn * 2 * v


# ------------------------------------------------------------------------------------ critics
def critic_layers(sd, prefix):
    names = sorted({k.split(".")[-2] for k in sd if k.startswith(prefix + "dense")}, key=lambda s: int(s[5:]))
    return [(sd[f"{prefix}{n}.weight"], sd[f"{prefix}{n}.bias"], n) for n in names]


def critic_fwd(x, layers, masks=None):
    """masks: list (one per hidden layer) of (rows, 20) tensors holding 0 or 1/(1-p); None = eval mode.
    Returns (out, acts, ds): acts[i] is the input of layer i, ds[i] = leaky'(pre_i) * mask_i."""
    acts, ds, a = [], [], x
    for li, (w, b, _) in enumerate(layers[:-1]):
        acts.append(a)
        pre = a @ w.t() + b
        d = torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, LEAK))
        if masks is not None:
            d = d * masks[li]
        ds.append(d)
        a = pre * d
    acts.append(a)
    w, b, _ = layers[-1]
    return a @ w.t() + b, acts, ds


def critic_bwd(dout, layers, ds):
    """First-order backward.  Returns (deltas, dinput): deltas[i] is d(loss)/d(pre_i) (post-mask chain)."""
    deltas = [None] * len(layers)
    deltas[-1] = dout
    for li in range(len(layers) - 2, -1, -1):
        deltas[li] = (deltas[li + 1] @ layers[li + 1][0]) * ds[li]
    return deltas, deltas[0] @ layers[0][0]


def critic_param_grads(grads, prefix, layers, deltas, acts):
    for (w, b, name), dl, a in zip(layers, deltas, acts):
        _acc(grads, f"{prefix}{name}.weight", dl.t() @ a)
        _acc(grads, f"{prefix}{name}.bias", dl.sum(0))


def critic_gp_pairs(layers, deltas, ds, ugrad):
    """Second-order part of the gradient penalty.  ``deltas`` is the first backward chain started from
    ones (train.py:75-81), ``ugrad`` = d(10*gp)/d(gradients).  Returns per-layer (left, right) pairs with
    dW_i += left_i^T @ right_i; biases get nothing (LeakyReLU'' = 0)."""
    pairs = [(deltas[0], ugrad)]
    e = ugrad @ layers[0][0].t()
    for li in range(1, len(layers)):
        ep = e * ds[li - 1]
        pairs.append((deltas[li], ep))
        e = ep @ layers[li][0].t()
    return pairs


def gradient_penalty(x, layers, masks):
    """Returns (gp, pairs) for 10*gp's contribution: whole-batch norm."""
    out, acts, ds = critic_fwd(x, layers, masks)
    deltas, g = critic_bwd(torch.ones_like(out), layers, ds)
    nrm = torch.sqrt((g * g).sum() + 1e-12)
    gp = (nrm - 1) ** 2
    ugrad = 10 * 2 * (nrm - 1) / nrm * g
    return gp, critic_gp_pairs(layers, deltas, ds, ugrad)


# ------------------------------------------------------------------------------------ encoder / decoder
def encoder_fwd(x, sd, p="enc."):
    h, saved = bilstm_layer_fwd(x, sd, p + "lstm.", 0)
    return h @ sd[p + "dense.weight"].t() + sd[p + "dense.bias"], (x, h, saved)


def encoder_bwd(dz, ctx, sd, grads, p="enc."):
    x, h, saved = ctx
    _acc(grads, p + "dense.weight", dz.t() @ h)
    _acc(grads, p + "dense.bias", dz.sum(0))
    dh = dz @ sd[p + "dense.weight"]
    bilstm_layer_bwd(dh, x, saved, sd, p + "lstm.", 0, grads)


def decoder_fwd(z, sd, hyperbolic, drop_mask=None, p="dec."):
    """drop_mask: (rows,128) inter-layer dropout mask holding 0 or 1/0.8 (models/tadgan.py:35-38); None = eval."""
    a0 = z @ sd[p + "dense1.weight"].t() + sd[p + "dense1.bias"]
    h0, s0 = bilstm_layer_fwd(a0, sd, p + "lstm.", 0)
    h0d = h0 if drop_mask is None else h0 * drop_mask
    h1, s1 = bilstm_layer_fwd(h0d, sd, p + "lstm.", 1)
    e = torch.tanh(h1 @ sd[p + "dense2.weight"].t() + sd[p + "dense2.bias"])
    ctx = dict(z=z, a0=a0, s0=s0, h0d=h0d, s1=s1, h1=h1, e=e, mask=drop_mask)
    if not hyperbolic:
        return e, None, ctx
    u = e @ sd[p + "hyperbolic_linear.weight"].t()
    ctx["u"] = u
    return head_epilogue_fwd(u, sd[p + "hyperbolic_linear.bias"]), e, ctx


def head_fwd(x, sd, p="dec."):
    u = x @ sd[p + "hyperbolic_linear.weight"].t()
    return head_epilogue_fwd(u, sd[p + "hyperbolic_linear.bias"]), u


def head_bwd(dr, x, u, sd, grads, p="dec."):
    du, db = head_epilogue_bwd(u, sd[p + "hyperbolic_linear.bias"], dr)
    _acc(grads, p + "hyperbolic_linear.weight", du.t() @ x)
    _acc(grads, p + "hyperbolic_linear.bias", db.sum(0))
    return du @ sd[p + "hyperbolic_linear.weight"]


def decoder_bwd(dout, ctx, sd, grads, hyperbolic, p="dec."):
    """dout: gradient of the decoder's (hyperbolic or tanh) output.  Returns dz."""
    de = head_bwd(dout, ctx["e"], ctx["u"], sd, grads, p) if hyperbolic else dout
    dpre = de * (1 - ctx["e"] ** 2)
    _acc(grads, p + "dense2.weight", dpre.t() @ ctx["h1"])
    _acc(grads, p + "dense2.bias", dpre.sum(0))
    dh1 = dpre @ sd[p + "dense2.weight"]
    dh0d = bilstm_layer_bwd(dh1, ctx["h0d"], ctx["s1"], sd, p + "lstm.", 1, grads)
    dh0 = dh0d if ctx["mask"] is None else dh0d * ctx["mask"]
    da0 = bilstm_layer_bwd(dh0, ctx["a0"], ctx["s0"], sd, p + "lstm.", 0, grads)
    _acc(grads, p + "dense1.weight", da0.t() @ ctx["z"])
    _acc(grads, p + "dense1.bias", da0.sum(0))
    return da0 @ sd[p + "dense1.weight"]


# ------------------------------------------------------------------------------------ iterations
def cx_iteration(sd, x, z, alpha, hyperbolic, masks=None):
    """train.py:18-104.  masks: dict(valid=[4], fake=[4], inter=[4], dec=(B,128)) or None.
    Returns (loss, grads of critic_x)."""
    masks = masks or {}
    B = x.shape[0]
    layers = critic_layers(sd, "cx.")
    grads = {}
    valid, acts_v, ds_v = critic_fwd(x, layers, masks.get("valid"))
    gen, _, _ = decoder_fwd(z, sd, hyperbolic, masks.get("dec"))
    fake, acts_f, ds_f = critic_fwd(gen, layers, masks.get("fake"))
    dl_v, _ = critic_bwd(torch.full_like(valid, -1.0 / B), layers, ds_v)
    dl_f, _ = critic_bwd(torch.full_like(fake, 1.0 / B), layers, ds_f)
    critic_param_grads(grads, "cx.", layers, dl_v, acts_v)
    critic_param_grads(grads, "cx.", layers, dl_f, acts_f)
    inter = alpha * x + (1 - alpha) * gen
    gp, pairs = gradient_penalty(inter, layers, masks.get("inter"))
    for (w, b, name), (left, right) in zip(layers, pairs):
        _acc(grads, f"cx.{name}.weight", left.t() @ right)
    return fake.mean() - valid.mean() + 10 * gp, grads


def cz_iteration(sd, x, z, alpha, masks=None):
    """train.py:107-186.  masks: dict(fake=[2], valid=[2], inter=[2])."""
    masks = masks or {}
    B = x.shape[0]
    layers = critic_layers(sd, "cz.")
    grads = {}
    z_enc, _ = encoder_fwd(x, sd)
    fake, acts_f, ds_f = critic_fwd(z_enc, layers, masks.get("fake"))
    valid, acts_v, ds_v = critic_fwd(z, layers, masks.get("valid"))
    dl_f, _ = critic_bwd(torch.full_like(fake, 1.0 / B), layers, ds_f)
    dl_v, _ = critic_bwd(torch.full_like(valid, -1.0 / B), layers, ds_v)
    critic_param_grads(grads, "cz.", layers, dl_f, acts_f)
    critic_param_grads(grads, "cz.", layers, dl_v, acts_v)
    inter = alpha * z + (1 - alpha) * z_enc
    gp, pairs = gradient_penalty(inter, layers, masks.get("inter"))
    for (w, b, name), (left, right) in zip(layers, pairs):
        _acc(grads, f"cz.{name}.weight", left.t() @ right)
    return fake.mean() - valid.mean() + 10 * gp, grads


def dec_iteration(sd, x, z, hyperbolic, masks=None):
    """train.py:189-249.  masks: dict(cz=[2], cx=[4], dec_gen=(B,128), dec_rec=(B,128)).
    Returns (loss, hyper_loss_or_mse, grads of decoder + encoder)."""
    masks = masks or {}
    B, S = x.shape
    grads = {}
    lz, lx = critic_layers(sd, "cz."), critic_layers(sd, "cx.")
    z_enc, ectx = encoder_fwd(x, sd)
    fake_z, _, ds_z = critic_fwd(z_enc, lz, masks.get("cz"))
    gen, _, gctx = decoder_fwd(z, sd, hyperbolic, masks.get("dec_gen"))
    fake_x, _, ds_x = critic_fwd(gen, lx, masks.get("cx"))
    rec, _, rctx = decoder_fwd(z_enc, sd, hyperbolic, masks.get("dec_rec"))
    _, dgen = critic_bwd(torch.full_like(fake_x, -1.0 / B), lx, ds_x)
    _, dz_enc = critic_bwd(torch.full_like(fake_z, -1.0 / B), lz, ds_z)
    if hyperbolic:
        hx, ux = head_fwd(x, sd)
        dist = rowdist_fwd(rec, hx)
        aux = dist.sum() / B
        drec, dhx = rowdist_bwd(rec, hx, torch.full_like(dist, 10.0 / B))
        head_bwd(dhx, x, ux, sd, grads)
    else:
        aux = ((rec - x) ** 2).mean()
        drec = 10 * 2 * (rec - x) / (B * S)
    decoder_bwd(dgen, gctx, sd, grads, hyperbolic)
    dz_enc = dz_enc + decoder_bwd(drec, rctx, sd, grads, hyperbolic)
    encoder_bwd(dz_enc, ectx, sd, grads)
    return 10 * aux - fake_x.mean() - fake_z.mean(), aux, grads


# ------------------------------------------------------------------------------------ optimizers
def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.Adam single-tensor rule (train.py:274-281); wd = L2-into-gradient."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = v.sqrt() / (1 - b2 ** t) ** 0.5 + eps
    return p - (lr / (1 - b1 ** t)) * m / denom, m, v


def radam_euclid_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5):
    """Euclidean branch of oracle/radam.py (rounding order of geoopt: sqrt(v/bc2) + eps)."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    den = (v / (1 - b2 ** t)).sqrt() + eps
    return p - lr * (m / (1 - b1 ** t)) / den, m, v


def radam_ball_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5, stabilize=10, maxnorm=MAXNORM_F32):
    """Ball branch of oracle/radam.py for one vector (``hyperbolic_linear.bias``)."""
    def lam(x):
        return 2 / (1 - (x * x).sum()).clamp_min(1e-15)

    def proj(x):
        n = x.norm().clamp_min(1e-15)
        return x / n * maxnorm if n > maxnorm else x

    g = g + wd * p
    lp = lam(p)
    rg = g / (lp * lp)
    m = b1 * m + (1 - b1) * rg
    v = b2 * v + (1 - b2) * (lp * lp * (rg * rg).sum())
    den = (v / (1 - b2 ** t)).sqrt() + eps
    newp = proj(p - lr * (m / (1 - b1 ** t)) / den)
    # parallel transport p -> newp:  gyr[newp, -p] m * lambda_p / lambda_newp   (math_.py:1738-1746, 656-676)
    a_, b_ = newp, -p
    u2, v2, uv = (a_ * a_).sum(), (b_ * b_).sum(), (a_ * b_).sum()
    uw, vw = (a_ * m).sum(), (b_ * m).sum()
    ca = -uw * v2 + vw + 2 * uv * vw
    cb = -vw * u2 - uw
    d = (1 + 2 * uv + u2 * v2).clamp_min(1e-15)
    m = (m + 2 * (ca * a_ + cb * b_) / d) * lp / lam(newp)
    if stabilize is not None and t % stabilize == 0:
        newp = proj(newp)
    return newp, m, v
