"""world_size-2 gloo tests of the sharding logic: CPU only, no HIP calls."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hypad_amd import parallel as par


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_windows, window, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import scoring
        rng = np.random.default_rng(0)
        y_hat = rng.standard_normal((n_windows, window)).astype(np.float32)       # same on every rank
        full, _ = scoring.unroll_predictions(y_hat, False)
        # --- scoring: each rank un-rolls only its timesteps from its window range + halo
        hb, he = par.window_range_with_halo(n_windows, world, rank, window)
        tb, te = par.timestep_range(n_windows, world, rank, window)
        local, _ = scoring.unroll_predictions(y_hat[hb:he], False)               # local timestep k == global hb + k
        mine = local[tb - hb: te - hb]
        assert np.array_equal(mine, full[tb:te]), "sharded un-roll differs from the unsharded one"
        # --- global z-score from one all-reduce
        x = np.abs(full.astype(np.float64))
        mean, std = par.global_zscore_stats(float(x[tb:te].sum()), float((x[tb:te] ** 2).sum()), te - tb)
        assert abs(mean - x.mean()) < 1e-12 and abs(std - x.std()) < 1e-12
        # --- training: signals are owned round-robin, metrics gathered at the end only
        mine_s = par.signals_of_rank(7, world, rank)
        merged = par.gather_signal_metrics({s: {"loss": float(s) * 0.5} for s in mine_s})
        assert sorted(merged) == list(range(7)) and merged[5]["loss"] == 2.5
        ret[rank] = (tb, te, len(mine_s))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_matches_unsharded():
    world, n_windows, window = 2, 257, 100
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_windows, window, ret), nprocs=world, join=True)
    assert ret[0][0] == 0 and ret[0][1] == ret[1][0] and ret[1][1] == n_windows + window - 1
    assert ret[0][2] + ret[1][2] == 7


def test_partition_helpers_cover_everything_once():
    for n, w in ((1916, 8), (1_000_000, 8), (5, 8), (64, 3)):
        ranges = [par.window_range(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
        assert max(e - b for b, e in ranges) - min(e - b for b, e in ranges) <= 1
        owned = sorted(s for r in range(w) for s in par.signals_of_rank(n if n < 100 else 64, w, r))
        assert owned == list(range(n if n < 100 else 64))


def _score_worker(rank, world, port, n_windows, window, combination, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import math
        from oracle import scoring
        rng = np.random.default_rng(1)                                     # the same "model outputs" on every rank
        rowdist_all, critic_all = rng.random(n_windows), rng.standard_normal(n_windows)
        recons_all = rng.standard_normal((n_windows, window))
        calls = []

        def evaluate(lo, hi):
            calls.append((lo, hi))
            return {"rowdist": torch.from_numpy(rowdist_all[lo:hi]), "critic": torch.from_numpy(critic_all[lo:hi]),
                    "norms": torch.from_numpy(np.linalg.norm(recons_all[lo:hi], axis=1))}

        def kde_modes(critic, w):
            ext = np.repeat(critic.numpy().reshape(-1, 1), w, axis=1)
            return torch.tensor([scoring.kde_mode(scoring.antidiagonal(ext, i)) for i in range(len(ext) + w - 1)], dtype=torch.float64)

        def finish(rowdist, modes, norms):
            crit = scoring.compute_critic_score(modes.numpy(), math.trunc(n_windows * 0.01))[:n_windows]
            fake = np.zeros((n_windows, 1)) if norms is None else norms.numpy().reshape(-1, 1)     # ||row|| == |value|
            return scoring.combine_scores(combination, crit, rowdist.numpy(), np.abs(fake))

        got = par.sharded_hyperbolic_scores(n_windows, window, evaluate, kde_modes, finish, "uncertainty" in combination)
        want = scoring.combine_scores(combination, scoring.final_critic_scores(critic_all, n_windows, window)[:n_windows], rowdist_all,
                                      recons_all)
        assert np.array_equal(got, want), "sharded scores differ from the unsharded ones"
        b, e = par.window_range(n_windows, world, rank)
        assert calls == [(max(0, b - window + 1), e)]                      # own range + halo, evaluated once
        ret[rank] = float(got.sum())
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_hyperbolic_scores_equal_unsharded():
    """BASELINE.json configs[4] partitioning: window ranges + re-computed halo + all-gather; bit-equal to one rank."""
    for combination in ("mult", "sum_uncertainty"):
        port = _free_port()
        ret = mp.Manager().dict()
        mp.spawn(_score_worker, args=(2, port, 157, 20, combination, ret), nprocs=2, join=True)
        assert ret[0] == ret[1]


# ---------------------------------------------------------------------------------------------- sharded Euclidean scoring
def _direct_rolling_mean(x, w):
    """Centred rolling mean with each window summed on its own (position-deterministic, like hypad_rolling_mean; pandas'
    online add/remove update makes the last bits depend on where the series starts)."""
    x = np.asarray(x, dtype=np.float64)
    if w == 0:
        return np.full_like(x, np.nan)
    out = np.full_like(x, np.nan)
    for i in range(len(x)):
        lo, hi = max(0, i - w // 2), min(len(x), i + (w - 1) // 2 + 1)
        v = x[lo:hi]
        v = v[~np.isnan(v)]
        if len(v) >= max(w // 2, 1):
            out[i] = v.sum() / len(v)
    return out


def _eucl_ops(kind, y, y_hat, critic, n_windows, window, calls=None):
    import math
    from oracle import scoring
    w = math.trunc(n_windows * 0.01)

    def evaluate(lo, hi):
        if calls is not None:
            calls.append((lo, hi))
        return {"recon": torch.from_numpy(y_hat[lo:hi]), "critic": torch.from_numpy(critic[lo:hi]),
                "true": torch.from_numpy(scoring.unroll_true(y[lo:hi]))}

    unroll = lambda r: torch.from_numpy(scoring.unroll_predictions(r.numpy(), False)[0])
    err = {"point": scoring.point_error, "area": scoring.area_error, "dtw": scoring.dtw_error}[kind]
    error_fn = lambda t, p: torch.from_numpy(np.asarray(err(t.numpy(), p.numpy().astype(np.float64)), dtype=np.float64))
    rolling = lambda e, ww, origin=0: torch.from_numpy(_direct_rolling_mean(e.numpy(), ww))

    def kde_modes(c, ww):
        ext = np.repeat(c.numpy().astype(np.float64).reshape(-1, 1), ww, axis=1)
        return torch.tensor([scoring.kde_mode(scoring.antidiagonal(ext, i)) for i in range(len(ext) + ww - 1)], dtype=torch.float64)

    def finish(e, modes):
        rec = scoring.zscore_clip(e.numpy())
        crit = scoring.compute_critic_score(modes.numpy(), w)
        return scoring.combine_euclidean("mult", crit, rec)

    return w, evaluate, unroll, error_fn, rolling, kde_modes, finish


def _eucl_inputs(n_windows, window):
    rng = np.random.default_rng(3)
    series = np.sin(np.arange(n_windows + window - 1) / 13.0) + 0.1 * rng.standard_normal(n_windows + window - 1)
    idx = np.arange(n_windows)[:, None] + np.arange(window)[None, :]
    y = series[idx]
    y_hat = (y + 0.1 * rng.standard_normal(y.shape)).astype(np.float32)
    critic = rng.standard_normal(n_windows).astype(np.float32)
    return y, y_hat, critic


def _eucl_worker(rank, world, port, n_windows, window, kind, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        y, y_hat, critic = _eucl_inputs(n_windows, window)
        calls = []
        w, *ops = _eucl_ops(kind, y, y_hat, critic, n_windows, window, calls)
        got = par.sharded_euclidean_scores(n_windows, window, w, *ops)
        a, b = par.extended_timestep_range(n_windows, world, rank, window, w)
        assert calls == [(max(0, a - window + 1), min(n_windows, b))]        # own range + halos, evaluated once
        if w > 0 and not np.isnan(got).any():                                 # one-all-reduce variant: same scores to rounding
            alt = par.sharded_euclidean_scores(n_windows, window, w, *ops[:-1],
                                               lambda e, m: ops[-1].__globals__["np"].clip(e.numpy(), 0, None) + 1, zscore="allreduce")
            from oracle import scoring
            want_rec = scoring.zscore_clip(scoring.reconstruction_errors(y, y_hat, 10, w, True, kind, with_summary=False)[0])
            assert np.allclose(alt, want_rec, rtol=0, atol=1e-9)
        ret[rank] = got
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_euclidean_scores_equal_unsharded():
    """The DTW leg of BASELINE.json configs[4]: timestep ranges + error / rolling-mean halos + the S-1 window halo, two
    all-gathers.  Two ranks == one rank bit for bit (position-deterministic ops), and == oracle.scoring.score_anomalies
    (pandas' rolling mean differs from a direct mean in the last bits: 1e-12)."""
    from oracle import scoring
    for kind, n_windows, window in (("point", 257, 40), ("dtw", 331, 25), ("area", 212, 30), ("dtw", 57, 20)):
        port = _free_port()
        ret = mp.Manager().dict()
        mp.spawn(_eucl_worker, args=(2, port, n_windows, window, kind, ret), nprocs=2, join=True)
        y, y_hat, critic = _eucl_inputs(n_windows, window)
        w, *ops = _eucl_ops(kind, y, y_hat, critic, n_windows, window)
        one = par.sharded_euclidean_scores(n_windows, window, w, *ops)       # no process group: world 1
        assert np.array_equal(ret[0], ret[1], equal_nan=True) and np.array_equal(ret[0], one, equal_nan=True), kind
        want, _, _ = scoring.score_anomalies(y, y_hat, critic, kind, "mult")
        assert np.allclose(one, want, rtol=0, atol=1e-10, equal_nan=True), kind
        assert np.isnan(one).all() == (w == 0)                                # fewer than 100 windows: pandas' window-0 NaNs


def _bench_aggregate_worker(rank, world, port, ret):
    """bench.py's multi-rank accounting (bench_signals / bench_signals_sharded): per-rank durations are max-reduced over the group, the
    job's rate = ALL ranks' windows / the slowest rank's time; plan_signal_groups gives every rank its share of the 8 x world signals."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from hypad_amd import train as ht
        mine_ms = 3.0 + 2.0 * rank                                      # rank 1 is the slow one
        job_ms = bench.RankGuard(dist, torch.device("cpu"), world).max(mine_ms)
        n = 8 * world
        plan, stream = ht.plan_signal_groups([bench.N_WINDOWS] * n, bench.B, world, rank)
        members = [i for _, ms in plan for i in ms]
        every = [None] * world
        dist.all_gather_object(every, members)
        ret[rank] = (job_ms, len(members), sorted(i for m in every for i in m), n * bench.N_BATCHES * bench.B / job_ms * 1e3)
    finally:
        dist.destroy_process_group()


def test_bench_aggregate_is_all_ranks_windows_over_the_slowest_rank():
    import bench
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_bench_aggregate_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        job_ms, mine, union, value = ret[r]
        assert job_ms == 5.0 and mine == 8 and union == list(range(16))
        assert abs(value - 16 * 29 * 64 / 5.0e-3) < 1e-6
    assert bench.RankGuard().max(1.25) == 1.25                                  # no group: unchanged
