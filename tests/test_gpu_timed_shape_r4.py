"""The workload bench.py times, tested as it is timed: BASELINE configs[1]'s epoch -- 1 916 windows, 29 minibatches x (5 + 5 + 1)
iterations = 145 resident critic iterations fed by 1 160 producer workgroups, device shuffles, replayed as a captured hipGraph --
(a) graph replay == eager launches == the per-iteration form of the phase, whole epoch, bit for bit; (b) the oracle
(oracle.train_iters, train.py:18-249 on CPU autograd) teacher-forced at critic iterations 0, 17, 18, 72 and 144 and at generator
launches 0 and 28 of that same captured epoch; (c) a phase that crosses the 512-iteration slice boundary (520 iterations)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import params_ns

pytestmark = pytest.mark.gpu
S, L, B, N, NB, NC = 100, 20, 64, 1916, 29, 5
TOL = 1e-4


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda", dtype).contiguous()


def windows(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n + S - 1)
    series = np.clip(np.sin(2 * np.pi * t / 288.0) + 0.05 * rng.standard_normal(len(t)), -1, 1)
    series[n // 2: n // 2 + 40] = np.clip(series[n // 2: n // 2 + 40] + 0.8, -1, 1)
    return series[np.arange(n)[:, None] + np.arange(S)[None, :]]


def oracle_modules(hyper, seed):
    from oracle import tadgan as ot
    torch.manual_seed(seed)
    mods = dict(enc=ot.Encoder(S, L).eval(), dec=ot.Decoder(S, L, hyper).eval(), cx=ot.CriticX(S, L).eval(), cz=ot.CriticZ(L).eval())
    if hyper:
        with torch.no_grad():
            mods["dec"].hyperbolic_linear.weight.mul_(50)
    return mods


def engine(mods, hyper=True, seed=1234, flags=0):
    from hypad_amd.engine import Engine
    eng = Engine(S, L, B, hyper, n_signals=1, lr=5e-4, seed=seed)
    for k, m in mods.items():
        eng.load_state_dict(k, m.state_dict())
    eng.epoch_flags = flags
    return eng


def snapshot(eng):
    return {(d, k): getattr(eng, d)[k].clone() for d in ("params", "exp_avg", "exp_avg_sq") for k in ("enc", "dec", "cx", "cz")}


def test_timed_epoch_graph_equals_eager_equals_per_iteration_form():
    """Exactly bench.py's step (make_step): train mode, device Philox noise and dropout, shuffles drawn inside the captured sequence,
    three replays.  The eager launch sequence gives the same bits (losses, weights, moments, counters -- every one of the 319
    iterations); the per-iteration form of the critic phase (HYPAD_EPOCH_PER_ITERATION: what a recovered epoch runs) sums the
    chunks' gradient shares in another order, so it tracks the resident form to rounding: first-iteration losses equal, the epoch's
    losses within 1e-4."""
    from hypad_amd import _C
    mods = oracle_modules(True, 3)
    x = cu(windows(N)).reshape(1, N, S)
    runs = {}
    for form in ("graph", "eager", "per_iteration"):
        eng = engine(mods, flags=_C.EPOCH_PER_ITERATION if form == "per_iteration" else 0)
        assert eng.critic_phase_persistent() == eng.critic_phase_producers(NB * NC) == (form != "per_iteration")
        perm = torch.empty(NC + 1, NB * B, dtype=torch.int32, device="cuda")
        out = []
        for _ in range(3):
            if form == "graph":
                l = eng.train_epoch_graph(x, perm, NB, NC, True, shuffle_windows=N)
            else:
                eng.draw_shuffles(perm, N)
                l = eng.train_epoch(x, perm, NB, NC, True)
            torch.cuda.synchronize()
            assert eng.status() == 0
            out.append((l.clone(), perm.clone()))
        runs[form] = (out, snapshot(eng), eng.counters[:4].cpu().tolist())
    for e in range(3):
        lg, pg = runs["graph"][0][e]
        assert torch.isfinite(lg).all() and lg.shape == (1, (2 * NC + 1) * NB, 4)
        head = pg[0].cpu().tolist()
        assert len(set(head)) == len(head) and int(pg.max()) < N and int(pg.min()) >= 0        # the head of a permutation
        le, pe = runs["eager"][0][e]
        assert torch.equal(pg, pe), e
        bad = (lg != le).any(dim=2).nonzero()
        assert bad.numel() == 0, ("eager", e, "first differing loss row", bad[0].tolist())
    assert not torch.equal(runs["graph"][0][0][1], runs["graph"][0][1][1])          # shuffled afresh at every replay
    assert runs["eager"][2] == runs["graph"][2] == runs["per_iteration"][2] == [3 * NB * NC, 3 * NB * NC, 3 * NB, 3 * (NB * NC + NB)]
    for key, t in runs["graph"][1].items():
        assert torch.equal(t, runs["eager"][1][key]), key
    # the per-iteration form: same shuffles and random streams, another summation order
    lg, pg = runs["graph"][0][0]
    lp, pp = runs["per_iteration"][0][0]
    assert torch.equal(pg, pp)
    assert torch.equal(lg[0, :2], lp[0, :2])                                          # iteration 0: no gradient has been summed yet
    rel = ((lg - lp).abs() / lp.abs().clamp_min(1.0)).max()
    assert float(rel) < 1e-3, float(rel)


@pytest.mark.parametrize("hyper", [True, False])
def test_captured_epoch_of_the_timed_shape_against_the_oracle(hyper):
    """The captured configs[1] epoch (device shuffles, graph replay, resident critic launch with its own producers), eval mode with
    injected z / alpha planes so that the oracle can follow: teacher-forced at critic iterations 0, 17, 18, 72, 144 (weights read
    out of m-iteration prefix runs, asserted bit-identical to the long run through every loss up to m) and at generator launches 0
    and 28 (generator state after 28 steps read out of a generator-only prefix)."""
    from oracle import train_iters as oi
    mods = oracle_modules(hyper, 11)
    w0 = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in mods.items()}
    P = params_ns(B, S, hyper)
    xw = windows(N, seed=1)
    x = cu(xw).reshape(1, N, S)
    rng = np.random.default_rng(5)
    nit = NB * NC
    planes = dict(z_cx=rng.standard_normal((nit, 1, B, L)).astype(np.float32), alpha_cx=rng.uniform(size=(nit, 1, B, S)).astype(np.float32),
                  z_cz=rng.standard_normal((nit, 1, B, L)).astype(np.float32), alpha_cz=rng.uniform(size=(nit, 1, B, L)).astype(np.float32),
                  z_gen=rng.standard_normal((NB, 1, B, L)).astype(np.float32))
    dpl = {k: cu(v) for k, v in planes.items()}
    full_eng = engine(mods, hyper)
    perm = torch.empty(NC + 1, NB * B, dtype=torch.int32, device="cuda")
    full = full_eng.train_epoch_graph(x, perm, NB, NC, False, shuffle_windows=N, noise=dpl)[0].cpu().numpy()
    torch.cuda.synchronize()
    assert full_eng.status() == 0 and np.isfinite(full).all()
    ri = perm.cpu().numpy()                                            # the shuffles the captured epoch drew for itself
    crit_rows = ri[:NC].reshape(nit, B)
    gen_rows = ri[NC].reshape(NB, B)

    def critics_after(m):
        """critic weights after the first m iterations: an epoch of ONE pass of m minibatches (+ m generator steps that do not touch them)"""
        eng = engine(mods_init, hyper)
        rows = np.stack([crit_rows[:m].reshape(-1), np.tile(gen_rows, (m // NB + 1, 1))[:m].reshape(-1)]).astype(np.int32)
        nz = {k: v[:m].contiguous() for k, v in dpl.items() if k != "z_gen"}
        nz["z_gen"] = dpl["z_gen"][[b % NB for b in range(m)]].contiguous()
        l = eng.train_epoch(x, cu(rows, torch.int32), m, 1, False, noise=nz)[0].cpu().numpy()
        assert np.array_equal(l[: 2 * m], full[: 2 * m]), m          # the prefix IS the long run's beginning, bit for bit
        return eng

    mods_init = oracle_modules(hyper, 11)
    for m in (0, 17, 18, 72, 144):
        for k in ("enc", "dec"):
            mods[k].load_state_dict(w0[k])
        src = None if m == 0 else critics_after(m)
        for k in ("cx", "cz"):
            mods[k].load_state_dict(w0[k] if src is None else {n: v.cpu() for n, v in src.state_dict(k).items()})
        o = oi.make_optimizers(mods["enc"], mods["dec"], mods["cx"], mods["cz"], P)
        sample = torch.from_numpy(xw[crit_rows[m]][:, :, None])
        ref_x = float(oi.critic_x_iteration(sample, mods["dec"], mods["cx"], o[0], P, z=planes["z_cx"][m, 0], alpha=planes["alpha_cx"][m, 0]))
        ref_z = float(oi.critic_z_iteration(sample, mods["enc"], mods["cz"], o[1], P, z=planes["z_cz"][m, 0], alpha=planes["alpha_cz"][m, 0]))
        assert abs(float(full[2 * m, 0]) - ref_x) < TOL * max(1, abs(ref_x)), ("critic_x", m, full[2 * m, 0], ref_x)
        assert abs(float(full[2 * m + 1, 0]) - ref_z) < TOL * max(1, abs(ref_z)), ("critic_z", m, full[2 * m + 1, 0], ref_z)
    # generator launches: critics as the phase left them (the generator launches read, never write them)
    final_critics = {k: {n: v.cpu() for n, v in full_eng.state_dict(k).items()} for k in ("cx", "cz")}
    for g in (0, 28):
        for k in ("cx", "cz"):
            mods[k].load_state_dict(final_critics[k])
        if g == 0:
            gen_state = w0
        else:
            eng = engine(mods_init, hyper)
            for k in ("cx", "cz"):
                eng.load_state_dict(k, final_critics[k])
            l = eng.train_epoch(x, cu(gen_rows[:g].reshape(1, -1), torch.int32), g, 0, False, noise={"z_gen": dpl["z_gen"][:g].contiguous()})[0].cpu().numpy()
            assert np.array_equal(l, full[2 * nit: 2 * nit + g]), "generator prefix"
            gen_state = {k: {n: v.cpu() for n, v in eng.state_dict(k).items()} for k in ("enc", "dec")}
        for k in ("enc", "dec"):
            mods[k].load_state_dict(gen_state[k])
        o = oi.make_optimizers(mods["enc"], mods["dec"], mods["cx"], mods["cz"], P)
        sample = torch.from_numpy(xw[gen_rows[g]][:, :, None])
        r = oi.decoder_iteration(sample, mods["enc"], mods["dec"], mods["cx"], mods["cz"], o[2], P, z=planes["z_gen"][g, 0])
        row = full[2 * nit + g]
        assert abs(float(row[0]) - float(r[0])) < 2 * TOL * max(1, abs(float(r[0]))), ("generator", g, row[0], float(r[0]))
        assert abs(float(row[1]) - float(r[1] if hyper else r[2])) < TOL, ("aux", g)


def test_phase_across_the_512_iteration_slice_boundary():
    """nb * nc = 104 * 5 = 520 critic iterations: the library processes them as slices of 512 + 8 (hypad_epoch_workspace_bytes
    sizes the record area for 512).  Same bits as the phase in ONE piece (a workspace sized for 520 by hand) and as slices of 200;
    in the resident form and in the per-iteration form."""
    from hypad_amd import _C
    nb, nc = 104, 5
    n = nb * B + 11
    mods = oracle_modules(True, 21)
    x = cu(windows(n, seed=2)).reshape(1, n, S)
    g = torch.Generator().manual_seed(0)
    perm = torch.stack([torch.randperm(n, generator=g)[: nb * B] for _ in range(nc + 1)]).to(torch.int32).cuda()
    for flags in (0, _C.EPOCH_PER_ITERATION):
        ref = None
        for ws in ("default", "one_piece", 200):
            eng = engine(mods, flags=flags)
            f = lambda k: _C.lib.hypad_epoch_workspace_bytes(ctypes.byref(eng.dims), k, 1)
            kw = {}
            if ws == "one_piece":
                eng._grow_workspace(f(512) + 8 * (f(2) - f(1)))          # room for 520 iterations' records
            elif ws != "default":
                kw["workspace_iters"] = ws
            l = eng.train_epoch(x, perm, nb, nc, True, **kw)
            torch.cuda.synchronize()
            assert eng.status() == 0 and torch.isfinite(l).all()
            got = (l.clone(), snapshot(eng), eng.counters[:4].cpu().tolist())
            if ref is None:
                ref = got
                continue
            bad = (got[0] != ref[0]).any(dim=2).nonzero()
            assert bad.numel() == 0, (flags, ws, "first differing loss row", bad[0].tolist())
            assert got[2] == ref[2] == [520, 520, nb, 520 + nb]
            for key, t in ref[1].items():
                assert torch.equal(t, got[1][key]), (flags, ws, key)
