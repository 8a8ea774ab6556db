"""Round 3: the status channel of the resident critic launch (include/hypad.h: counters[4], hypad_epoch_status,
hypad_epoch_restore, HYPAD_EPOCH_PER_ITERATION) and the engine's graph cache.

The resident launch (critic_persistent_kernel) needs all its critic workgroups co-resident and bounds every wait.  A wait
that gives up must (a) reach the host, (b) stop the epoch's remaining launches (the generator must not be stepped against
half-updated critics), and (c) be recoverable: critics + counters restored, the epoch repeated with one launch per critic
iteration -- bit for bit what a healthy epoch in that form produces (same random streams; the two forms of the phase differ
only in floating-point summation order; train.py:299-356 is the schedule either way)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda", dtype).contiguous()


def _setup(ns=1, seed=5, S=100, B=64, nb=3, nc=2, hyper=True):
    from hypad_amd.engine import Engine
    from oracle import tadgan as ot
    rng = np.random.default_rng(seed)
    N = nb * B
    x = cu(np.clip(np.sin(np.arange(N)[None, :, None] / (9.0 + np.arange(ns)[:, None, None]) + np.arange(S)[None, None, :] / 7.0)
                   + 0.05 * rng.standard_normal((ns, N, S)), -1, 1))
    perms = [cu(np.stack([rng.permutation(N) for _ in range(nc + 1)]), torch.int32) for _ in range(3)]

    def engine():
        eng = Engine(S, 20, B, hyper, n_signals=ns, lr=5e-4, seed=99)
        for s in range(ns):
            torch.manual_seed(seed + s)
            mods = dict(enc=ot.Encoder(S, 20), dec=ot.Decoder(S, 20, hyper), cx=ot.CriticX(S, 20), cz=ot.CriticZ(20))
            for k, m in mods.items():
                eng.load_state_dict(k, m.state_dict(), s)
        return eng
    return engine, x, perms, nb, nc


def _snapshot(eng):
    torch.cuda.synchronize()
    return ({k: eng.params[k].clone() for k in eng.params}, {k: eng.exp_avg[k].clone() for k in eng.params},
            {k: eng.exp_avg_sq[k].clone() for k in eng.params}, eng.counters.clone())


def _same(a, b):
    return all(torch.equal(a[i][k], b[i][k]) for i in range(3) for k in a[i]) and torch.equal(a[3], b[3])


@pytest.mark.parametrize("ns,graph,aux", [(1, False, 0), (3, False, 0), (1, True, 0), (3, False, 2), (3, True, 1)])
def test_resident_launch_that_gives_up_is_reported_stops_the_epoch_and_is_recovered(ns, graph, aux, monkeypatch):
    from hypad_amd import _C
    monkeypatch.setenv("HYPAD_AUX_STREAMS", str(aux))          # (aux > 0: the generator phase in model groups must stop and recover alike)
    engine, x, perms, nb, nc = _setup(ns)
    good = engine()
    assert good.critic_phase_persistent(), "this test is about the resident form"
    run = (lambda e, p, **kw: e.train_epoch_graph(x, p, nb, nc, True, **kw)) if graph else (lambda e, p, **kw: e.train_epoch(x, p, nb, nc, True, **kw))
    perm_buf = perms[0].clone()                    # (graph replays read the shuffles from one static buffer)
    resident = run(good, perm_buf).clone()         # a healthy epoch, resident form
    assert good.status() == 0
    good = engine()                                # ... and three healthy epochs with one launch per critic iteration: the reference run
    good.epoch_flags = _C.EPOCH_PER_ITERATION
    l_good = [run(good, perm_buf).clone()]
    assert good.status() == 0 and float((l_good[0] - resident).abs().max()) < 1e-3
    s_good = [_snapshot(good)]
    l_good.append(run(good, perm_buf).clone())     # (the same shuffles again: the epoch queued behind the failing one, below)
    s_good.append(_snapshot(good))
    perm_buf.copy_(perms[1])
    l_good.append(run(good, perm_buf).clone())
    s_good.append(_snapshot(good))

    bad = engine()
    start = _snapshot(bad)
    bad.epoch_flags = 2 << _C.EPOCH_TEST_GIVE_UP_SHIFT          # critic_x chunk 0 of signal 0 "times out" at critic iteration 2
    perm_buf.copy_(perms[0])
    l_bad = run(bad, perm_buf).clone()
    code = bad.status()
    assert code == 0x100 + 2, hex(code)
    after = _snapshot(bad)
    # fail-stop: no generator step was taken (parameters, moments, step counter), the loss row of the failing iteration is NaN
    for k in ("enc", "dec"):
        assert torch.equal(after[0][k], start[0][k]) and torch.equal(after[1][k], start[1][k]), k
    assert int(after[3][2]) == 0 and int(after[3][4]) == code
    assert bool(torch.isnan(l_bad[0, 2 * 1, 0])) or bool(torch.isnan(l_bad[0, 2 * 2, 0]))
    # a second epoch on the failed state changes nothing at all (its launches are no-ops; the snapshot is not overwritten)
    run(bad, perm_buf)
    assert _same(_snapshot(bad), after)
    # without recovery the host gets an exception
    with pytest.raises(_C.HypadError):
        bad.check_status(recover=False)
    # recovery: restore + EVERY epoch queued since -- the failed one and the one behind it -- repeated with one launch per critic
    # iteration == the healthy per-iteration epochs, bit for bit
    assert bad.check_status() == code
    assert bad.status() == 0 and bad.epoch_flags == _C.EPOCH_PER_ITERATION
    assert torch.equal(bad._last_epoch["losses"], l_good[1])
    assert _same(_snapshot(bad), s_good[1])
    # ... and the engine goes on in that form
    perm_buf.copy_(perms[1])
    l2 = run(bad, perm_buf)
    assert bad.check_status() == 0
    assert torch.equal(l2, l_good[2]) and _same(_snapshot(bad), s_good[2])


@pytest.mark.parametrize("graph", [False, True])
def test_epochs_queued_around_a_failure_are_all_repaired(graph):
    """Four epochs queued without a check in between (bench.py queues 20): the first completes in the resident form, the second's
    resident launch gives up, the third and fourth are no-ops behind it.  ONE check_status finds the failure, keeps the first epoch,
    and repeats the other three in order == a run whose epochs 2..4 used the per-iteration form, bit for bit.  With the shuffles
    drawn inside the captured sequence (graph) the repeats draw the same permutations again (the rng tick is restored)."""
    from hypad_amd import _C
    engine, x, perms, nb, nc = _setup(1)
    n_windows = x.shape[1]

    def run(e, flags=None):
        buf = e.__dict__.setdefault("_test_perm", perms[0].clone())
        if graph:
            if flags is not None:
                e.epoch_flags = flags
            return e.train_epoch_graph(x, buf, nb, nc, True, shuffle_windows=n_windows).clone()
        return e.train_epoch(x, buf, nb, nc, True, flags=flags).clone()

    ref = engine()
    l_ref = [run(ref)]
    assert ref.check_status() == 0
    ref.epoch_flags = _C.EPOCH_PER_ITERATION
    l_ref += [run(ref, _C.EPOCH_PER_ITERATION if not graph else None) for _ in range(3)]
    assert ref.check_status() == 0
    bad = engine()
    l_bad = [run(bad)]
    give_up = 3 << _C.EPOCH_TEST_GIVE_UP_SHIFT
    l_bad.append(run(bad, give_up))
    if graph:
        bad.epoch_flags = give_up            # (same captured graph for the epochs behind it: they are no-ops anyway)
    l_bad += [run(bad, 0 if not graph else None) for _ in range(2)]
    code = bad.check_status()
    assert code == 0x100 + 3 and bad.status() == 0
    assert torch.equal(l_bad[0], l_ref[0])
    assert torch.equal(bad._last_epoch["losses"], l_ref[3])
    assert _same(_snapshot(bad), _snapshot(ref))


def test_epoch_status_entry_points_validate_their_arguments():
    import ctypes
    from hypad_amd import _C
    engine, x, perms, nb, nc = _setup()
    eng = engine()
    st = eng._state()
    out = ctypes.c_int(-1)
    assert _C.lib.hypad_epoch_status(None, ctypes.byref(out), _C.stream()) == -1
    assert _C.lib.hypad_epoch_status(ctypes.byref(st), None, _C.stream()) == -1
    assert _C.lib.hypad_epoch_status(ctypes.byref(st), ctypes.byref(out), _C.stream()) == 0 and out.value == 0
    assert _C.lib.hypad_epoch_restore(ctypes.byref(eng.dims), ctypes.byref(st), None, 0, _C.stream()) == -2
    assert _C.lib.hypad_epoch_restore(ctypes.byref(eng.dims), ctypes.byref(st), eng.workspace.data_ptr(), 64, _C.stream()) == -2


def test_captured_epoch_is_recaptured_when_what_it_froze_changes():
    """the graph cache key must cover every address and by-value scalar the capture froze -- workspace, arenas,
    moments, counters, lr / betas / eps / weight decay -- not only x / row_index / losses."""
    engine, x, perms, nb, nc = _setup()
    a, b = engine(), engine()
    pa, pb = perms[0].clone(), perms[0].clone()
    a.train_epoch_graph(x, pa, nb, nc, True)
    b.train_epoch_graph(x, pb, nb, nc, True)
    assert len(a._graphs) == 1
    # a larger workspace (what profile_iteration or a longer epoch asks for) drops the captured epochs
    a._grow_workspace(a._ws_bytes + (1 << 20))
    assert "_graphs" not in a.__dict__
    # a changed learning rate must take effect in the next replayed epoch: compare with an eager engine given the same change
    for e in (a, b):
        e.lr = 1e-3
    la = a.train_epoch_graph(x, pa, nb, nc, True).clone()
    lb = b.train_epoch(x, pb, nb, nc, True)
    torch.cuda.synchronize()
    assert torch.equal(la, lb)
    for k in a.params:
        assert torch.equal(a.params[k], b.params[k]), k
    # adopting other arenas (hypad_amd.train binds module views this way) drops them too
    a.adopt({k: v.clone().view(-1) for k, v in a.params.items()})
    assert "_graphs" not in a.__dict__


# ------------------------------------------------------------------------------------------------ XCD placement of the resident launch
@pytest.mark.parametrize("ns,S,B", [(1, 100, 64), (5, 100, 64), (8, 100, 64), (1, 150, 256)])
def test_chunks_of_a_critic_share_an_xcd_and_the_epoch_keeps_its_bits(ns, S, B, monkeypatch):
    """critic_persistent_kernel deals the chunk workgroups of one critic to one XCD (ids stretched by 8) and, having read from
    the hardware that they really share it, keeps their exchange -- gradient shares, scalar granules, epoch words -- in that
    XCD's L2 (stores without the write-through bit).  HYPAD_EPOCH_ID_ORDER deals them in id order (the round-2 placement: a
    critic's chunks on different XCDs -> the write-through forms).  Placement and store flavour are speed matters only: both
    epochs must agree bit for bit -- losses, all four networks, moments, counters -- also with the chip busy on a side stream;
    and the census word (counters[5]) must say which form ran."""
    nb, nc = 4, 3
    engine, x, perms, _, _ = _setup(ns, seed=7, S=S, B=B, nb=nb, nc=nc)
    outs = {}
    from hypad_amd import _C
    for mode in ("1", "0"):
        e = engine()
        e.epoch_flags = 0 if mode == "1" else _C.EPOCH_ID_ORDER
        assert e.critic_phase_persistent()
        l = e.train_epoch(x, perms[0], nb, nc, True).clone()
        l2 = e.train_epoch(x, perms[1], nb, nc, True).clone()
        assert e.status() == 0
        outs[mode] = (l, l2, _snapshot(e))
    census = {m: int(outs[m][2][3][5]) for m in outs}
    print("critics whose chunks share an XCD: stretched ids", census["1"], "of", 2 * ns, "; id order", census["0"])
    assert census["1"] == 2 * ns, census            # (what the dispatcher is observed to do; the kernel would be correct without it)
    assert census["0"] == 0 or B // 16 == 1, census
    a, b = outs["1"], outs["0"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(a[2][i][k], b[2][i][k]) for i in range(3) for k in a[2][i]) and torch.equal(a[2][3][:5], b[2][3][:5])
    # round 3 also stopped zeroing the activation / delta tiles after every iteration (every element read is written first):
    # HYPAD_EPOCH_CLEAR_TILES brings the sweep back -- same bits
    e = engine()
    e.epoch_flags = _C.EPOCH_CLEAR_TILES
    l = e.train_epoch(x, perms[0], nb, nc, True).clone()
    l2 = e.train_epoch(x, perms[1], nb, nc, True).clone()
    snap = _snapshot(e)
    assert torch.equal(l, a[0]) and torch.equal(l2, a[1]) and all(torch.equal(snap[i][k], a[2][i][k]) for i in range(3) for k in snap[i])
    # the same under an uneven background load, several times
    side = torch.cuda.Stream()
    src = torch.empty(32 << 20, dtype=torch.float32, device="cuda")
    dst = torch.empty_like(src)
    for rep in range(6):
        e = engine()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):
                dst.copy_(src)
        l = e.train_epoch(x, perms[0], nb, nc, True)
        l2 = e.train_epoch(x, perms[1], nb, nc, True)
        torch.cuda.synchronize()
        assert torch.equal(l, a[0]) and torch.equal(l2, a[1]), rep
        snap = _snapshot(e)
        assert all(torch.equal(snap[i][k], a[2][i][k]) for i in range(3) for k in snap[i]), rep


# ------------------------------------------------------------------------------------------------ device-side shuffles
def test_epoch_shuffles_are_uniform_permutations_and_replay_afresh():
    """hypad_epoch_shuffles stands in for the DataLoader's shuffle=True, drop_last=True (main.py:38): every pass the head of a
    fresh permutation of the windows; captured into the epoch's graph it must draw NEW permutations at every replay (the key
    includes the device rng tick) and the epoch must equal the one that is handed the same permutations explicitly."""
    from hypad_amd import _C
    engine, x, perms, nb, nc = _setup(ns=1, nb=3, nc=2)
    n_windows, take = x.shape[1], nb * 64
    e = engine()
    buf = torch.empty(nc + 1, take, dtype=torch.int32, device="cuda")
    draws = []
    for tick in range(40):
        e.counters[3] = tick
        draws.append(e.draw_shuffles(buf, n_windows).clone())
    d = torch.stack(draws).cpu().numpy()                       # (40, passes, take)
    assert d.min() >= 0 and d.max() < n_windows
    for t in range(40):
        for p in range(nc + 1):
            assert len(set(d[t, p].tolist())) == take          # no index twice
    assert not np.array_equal(d[0, 0], d[0, 1]) and not np.array_equal(d[0], d[1])
    e.counters[3] = 0
    assert torch.equal(e.draw_shuffles(buf, n_windows), draws[0])           # keyed, not stateful
    # uniformity: over 120 permutation heads every window should be drawn ~ take / n of the time, at a mean position ~ take / 2
    big = Engine_draws(e, 2000, 1536, 400)
    freq = np.bincount(big.reshape(-1), minlength=2000) / big.shape[0]
    assert abs(freq.mean() - 1536 / 2000) < 1e-9 and freq.std() < 3.5 * np.sqrt(0.768 * 0.232 / big.shape[0])
    assert 0.35 < (big[:, 0] < 1000).mean() < 0.65
    # in-graph shuffles == the same permutations handed over explicitly, and a second replay uses other permutations
    a, b = engine(), engine()
    pa = torch.empty(nc + 1, take, dtype=torch.int32, device="cuda")
    la = a.train_epoch_graph(x, pa, nb, nc, True, shuffle_windows=n_windows).clone()
    first = pa.clone()
    lb = b.train_epoch(x, b.draw_shuffles(torch.empty_like(pa), n_windows), nb, nc, True)
    torch.cuda.synchronize()
    assert torch.equal(la, lb) and all(torch.equal(a.params[k], b.params[k]) for k in a.params)
    a.train_epoch_graph(x, pa, nb, nc, True, shuffle_windows=n_windows)
    torch.cuda.synchronize()
    assert not torch.equal(pa, first)
    assert _C.lib.hypad_epoch_shuffles(pa.data_ptr(), 3, take, 5000, 1, None, _C.stream()) == -3       # > 4096 windows: caller's generator


def Engine_draws(e, n_windows, take, reps):
    buf = torch.empty(1, take, dtype=torch.int32, device="cuda")
    out = []
    for t in range(reps):
        e.counters[3] = 1000 + t
        out.append(e.draw_shuffles(buf, n_windows)[0].clone())
    e.counters[3] = 0
    return torch.stack(out).cpu().numpy()


# ------------------------------------------------------------------------------------------------ generator phase in model groups
@pytest.mark.parametrize("ns,graph", [(2, False), (5, False), (8, True), (11, False)])
def test_generator_phase_in_model_groups_keeps_the_bits(ns, graph):
    """hypad_epoch_io.aux_streams (ABI 4): with several models per GPU the generator phase (train.py:347-352) runs them in groups,
    each group's chain of launches on a stream of its own; every launch carries its step number and rng tick, the counters advance
    once after the groups joined.  Two epochs on 0, 1, 3 and 7 auxiliary streams -- eagerly and as a replayed graph -- must agree
    bit for bit: losses, all four networks, moments, counters; and the injected-noise planes must reach the right model."""
    nb, nc = 3, 2
    engine, x, perms, _, _ = _setup(ns, seed=21, nb=nb, nc=nc)
    outs = {}
    for aux in (0, 1, 3, 7):
        e = engine()
        e.aux_streams = aux
        run = e.train_epoch_graph if graph else e.train_epoch
        l = run(x, perms[0], nb, nc, True).clone()
        l2 = run(x, perms[1], nb, nc, True).clone()
        assert e.status() == 0
        outs[aux] = (l, l2, _snapshot(e))
    ref = outs[0]
    assert int(ref[2][3][2]) == 2 * nb                       # generator steps counted once per iteration
    for aux in (1, 3, 7):
        got = outs[aux]
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), aux
        assert _same(got[2], ref[2]), aux
    # the models differ from each other (a group reading another group's rows would still be "consistent" across aux counts
    # only if it did so in every run: compare with single-model engines)
    e1 = engine()
    e1.aux_streams = 7
    l_all = e1.train_epoch(x, perms[0], nb, nc, True).clone()
    assert not torch.equal(l_all[0], l_all[ns - 1])
