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
