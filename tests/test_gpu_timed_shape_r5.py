"""The OTHER shapes bench.py times, tested against the oracle at full length:
* BASELINE configs[3] as timed (`multivariate` section: window 150, batch 256, 20 480 windows U(-1, 1) -> 80 minibatches x 5 passes = 400
  resident critic iterations, /root/reference/configs/multivariate.yaml:1-19): the captured epoch teacher-forced against
  oracle.train_iters (train.py:18-249 on CPU autograd) at critic iterations 0, 199, 399 and generator launches 0 and 79;
* the 32-model epoch (`signals32` section: 32 models per GPU, 145 iterations, record precompute launch in front of the resident launch,
  encoder table): one teacher-forced critic iteration and one generator launch of three of the 32 models.
Same method as tests/test_gpu_timed_shape_r4.py: eval mode with injected z / alpha planes so that the oracle can follow; the weights an
iteration starts from are read out of a prefix run that is asserted bit-identical to the long run."""
import numpy as np
import pytest
import torch

from helpers import params_ns

pytestmark = pytest.mark.gpu
L, NC = 20, 5
TOL = 1e-4


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda", dtype).contiguous()


def oracle_modules(S, hyper, seed):
    from oracle import tadgan as ot
    torch.manual_seed(seed)
    mods = dict(enc=ot.Encoder(S, L).eval(), dec=ot.Decoder(S, L, hyper).eval(), cx=ot.CriticX(S, L).eval(), cz=ot.CriticZ(L).eval())
    if hyper:
        with torch.no_grad():
            mods["dec"].hyperbolic_linear.weight.mul_(50)
    return mods


def make_engine(S, B, per_signal_mods, hyper=True, seed=1234):
    from hypad_amd.engine import Engine
    eng = Engine(S, L, B, hyper, n_signals=len(per_signal_mods), lr=5e-4, seed=seed)
    for slot, mods in enumerate(per_signal_mods):
        for k, m in mods.items():
            eng.load_state_dict(k, m.state_dict() if hasattr(m, "state_dict") else m, slot)
    return eng


def states(mods):
    return {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in mods.items()}


def check_against_oracle(S, B, n_windows, nb, xw, w0_all, planes, full, ri, critic_its, gen_launches, slots, hyper=True):
    """xw: (k, N, S) windows on the host; w0_all: per model the initial state dicts; planes: host noise planes (iterations, k, B, .);
    full: (k, iterations, 4) losses of the long captured run; ri: the (NC + 1, nb * B) shuffles it drew (shared by the models)."""
    from oracle import train_iters as oi
    k = len(w0_all)
    nit = nb * NC
    P = params_ns(B, S, hyper)
    x = cu(xw)
    dpl = {n: cu(v) for n, v in planes.items()}
    crit_rows = ri[:NC].reshape(nit, B)
    gen_rows = ri[NC].reshape(nb, B)

    def critics_after(m):
        """critic weights of every model after the first m iterations: an epoch of ONE pass of m minibatches (+ m generator steps that
        do not touch the critics); its losses must be the long run's first 2 m rows, bit for bit, for every model"""
        eng = make_engine(S, B, w0_all, hyper)
        rows = np.stack([crit_rows[:m].reshape(-1), np.tile(gen_rows, (m // nb + 1, 1))[:m].reshape(-1)]).astype(np.int32)
        nz = {n: v[:m].contiguous() for n, v in dpl.items() if n != "z_gen"}
        nz["z_gen"] = dpl["z_gen"][[b % nb for b in range(m)]].contiguous()
        l = eng.train_epoch(x, cu(rows, torch.int32), m, 1, False, noise=nz).cpu().numpy()
        assert eng.status() == 0
        assert np.array_equal(l[:, : 2 * m], full[:, : 2 * m]), ("critic prefix", m)
        return eng

    for m in critic_its:
        src = None if m == 0 else critics_after(m)
        for s in slots:
            mods = oracle_modules(S, hyper, 0)
            for net in ("enc", "dec"):
                mods[net].load_state_dict(w0_all[s][net])
            for net in ("cx", "cz"):
                mods[net].load_state_dict(w0_all[s][net] if src is None else {n: v.cpu() for n, v in src.state_dict(net, s).items()})
            o = oi.make_optimizers(mods["enc"], mods["dec"], mods["cx"], mods["cz"], P)
            sample = torch.from_numpy(xw[s][crit_rows[m]][:, :, None])
            ref_x = float(oi.critic_x_iteration(sample, mods["dec"], mods["cx"], o[0], P, z=planes["z_cx"][m, s], alpha=planes["alpha_cx"][m, s]))
            ref_z = float(oi.critic_z_iteration(sample, mods["enc"], mods["cz"], o[1], P, z=planes["z_cz"][m, s], alpha=planes["alpha_cz"][m, s]))
            assert abs(float(full[s, 2 * m, 0]) - ref_x) < TOL * max(1, abs(ref_x)), ("critic_x", s, m, full[s, 2 * m, 0], ref_x)
            assert abs(float(full[s, 2 * m + 1, 0]) - ref_z) < TOL * max(1, abs(ref_z)), ("critic_z", s, m, full[s, 2 * m + 1, 0], ref_z)
    return crit_rows, gen_rows, dpl, x


def generator_against_oracle(S, B, nb, xw, w0_all, planes, full, gen_rows, dpl, x, final_critics, launches, slots, hyper=True):
    from oracle import train_iters as oi
    nit = nb * NC
    P = params_ns(B, S, hyper)
    for g in launches:
        eng = None
        if g > 0:                                    # generator state after g steps: a generator-only prefix with the phase's final critics
            eng = make_engine(S, B, w0_all, hyper)
            for s in range(len(w0_all)):
                for net in ("cx", "cz"):
                    eng.load_state_dict(net, final_critics[s][net], s)
            l = eng.train_epoch(x, cu(gen_rows[:g].reshape(1, -1), torch.int32), g, 0, False, noise={"z_gen": dpl["z_gen"][:g].contiguous()}).cpu().numpy()
            assert np.array_equal(l, full[:, 2 * nit: 2 * nit + g]), ("generator prefix", g)
        for s in slots:
            mods = oracle_modules(S, hyper, 0)
            for net in ("cx", "cz"):
                mods[net].load_state_dict(final_critics[s][net])
            for net in ("enc", "dec"):
                mods[net].load_state_dict(w0_all[s][net] if eng is None else {n: v.cpu() for n, v in eng.state_dict(net, s).items()})
            o = oi.make_optimizers(mods["enc"], mods["dec"], mods["cx"], mods["cz"], P)
            sample = torch.from_numpy(xw[s][gen_rows[g]][:, :, None])
            r = oi.decoder_iteration(sample, mods["enc"], mods["dec"], mods["cx"], mods["cz"], o[2], P, z=planes["z_gen"][g, s])
            row = full[s, 2 * nit + g]
            assert abs(float(row[0]) - float(r[0])) < 2 * TOL * max(1, abs(float(r[0]))), ("generator", s, g, row[0], float(r[0]))
            assert abs(float(row[1]) - float(r[1] if hyper else r[2])) < TOL, ("aux", s, g)


def test_configs3_epoch_as_timed_against_the_oracle():
    """bench.py `multivariate`: Cfg("configs[3]", S=150, B=256, n_windows=20480, data="uniform") -- 80 x (5 + 5 + 1) iterations, device
    shuffles (torch-drawn: 20 480 windows exceed the in-graph sort), graph replay."""
    S, B, N = 150, 256, 20480
    nb = N // B
    nit = nb * NC
    assert nit == 400
    mods = oracle_modules(S, True, 11)
    w0 = states(mods)
    xw = np.random.default_rng(0).uniform(-1, 1, (1, N, S))
    rng = np.random.default_rng(5)
    planes = dict(z_cx=rng.standard_normal((nit, 1, B, L)).astype(np.float32), alpha_cx=rng.uniform(size=(nit, 1, B, S)).astype(np.float32),
                  z_cz=rng.standard_normal((nit, 1, B, L)).astype(np.float32), alpha_cz=rng.uniform(size=(nit, 1, B, L)).astype(np.float32),
                  z_gen=rng.standard_normal((nb, 1, B, L)).astype(np.float32))
    eng = make_engine(S, B, [w0])
    assert eng.critic_phase_persistent()                        # the timed form: ONE resident critic launch (critic_persistent_kernel<150, 20, ...>)
    g = torch.Generator(device="cuda").manual_seed(3)
    perm = torch.rand(NC + 1, N, device="cuda", generator=g).argsort(dim=1)[:, : nb * B].to(torch.int32).contiguous()      # bench.make_step's host_shuffle branch
    dpl = {n: cu(v) for n, v in planes.items()}
    full = eng.train_epoch_graph(cu(xw), perm, nb, NC, False, noise=dpl).cpu().numpy()
    torch.cuda.synchronize()
    assert eng.status() == 0 and np.isfinite(full).all() and full.shape == (1, 11 * nb, 4)
    ri = perm.cpu().numpy()
    crit_rows, gen_rows, dpl, x = check_against_oracle(S, B, N, nb, xw, [w0], planes, full, ri, (0, 199, 399), None, (0,))
    final_critics = [{k: {n: v.cpu() for n, v in eng.state_dict(k, 0).items()} for k in ("cx", "cz")}]
    generator_against_oracle(S, B, nb, xw, [w0], planes, full, gen_rows, dpl, x, final_critics, (0, 79), (0,))


def test_32_model_epoch_one_teacher_forced_iteration_per_phase():
    """bench.py `signals32`: 32 models per GPU at configs[1]'s shape (1 916 windows, 29 x 11 iterations): critic iterations 0 and 72 and
    generator launches 0 and 14 of models 0, 13 and 31 against the oracle."""
    S, B, N, k = 100, 64, 1916, 32
    nb = N // B
    nit = nb * NC
    w0_all, xs = [], []
    for s in range(k):
        w0_all.append(states(oracle_modules(S, True, 100 + s)))
        r = np.random.default_rng(s)
        t = np.arange(N + S - 1)
        series = np.clip(np.sin(2 * np.pi * (t + 17 * s) / (200.0 + 5 * s)) + 0.05 * r.standard_normal(len(t)), -1, 1)
        xs.append(series[np.arange(N)[:, None] + np.arange(S)[None, :]])
    xw = np.stack(xs)
    rng = np.random.default_rng(6)
    f32 = lambda a: a.astype(np.float32)
    planes = dict(z_cx=f32(rng.standard_normal((nit, k, B, L))), alpha_cx=rng.random((nit, k, B, S), dtype=np.float32),
                  z_cz=f32(rng.standard_normal((nit, k, B, L))), alpha_cz=rng.random((nit, k, B, L), dtype=np.float32),
                  z_gen=f32(rng.standard_normal((nb, k, B, L))))
    eng = make_engine(S, B, w0_all)
    assert eng.critic_phase_persistent() and not eng.critic_phase_producers(nit)       # 32 models: every CU holds a critic -> precompute launch in front
    perm = torch.empty(NC + 1, nb * B, dtype=torch.int32, device="cuda")
    dpl = {n: cu(v) for n, v in planes.items()}
    full = eng.train_epoch_graph(cu(xw), perm, nb, NC, False, shuffle_windows=N, noise=dpl).cpu().numpy()
    torch.cuda.synchronize()
    assert eng.status() == 0 and np.isfinite(full).all() and full.shape == (k, 11 * nb, 4)
    ri = perm.cpu().numpy()
    slots = (0, 13, 31)
    crit_rows, gen_rows, dpl, x = check_against_oracle(S, B, N, nb, xw, w0_all, planes, full, ri, (0, 72), None, slots)
    final_critics = [{net: {n: v.cpu() for n, v in eng.state_dict(net, s).items()} for net in ("cx", "cz")} for s in range(k)]
    generator_against_oracle(S, B, nb, xw, w0_all, planes, full, gen_rows, dpl, x, final_critics, (0, 14), slots)
    assert not np.array_equal(full[0], full[13])                # the models really are different models
