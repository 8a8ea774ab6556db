"""RCCL for real on the one GPU this pool offers: a fresh process creates a world-size-1 `nccl` (= RCCL on ROCm) process
group bound to cuda:0, and the sharded scorers run their collectives through it -- `broadcast` of the parameter arenas,
`all_gather_into_tensor` of the per-window / per-timestep vectors, `all_reduce` of the z-score statistics
(hypad_amd/parallel.py executes them whenever a process group exists, world size 1 included) -- and must equal the un-sharded
pipeline bit for bit (the reference itself is single-GPU: train.py:415-426, main.py:32-70)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _windows(n, S):
    rng = np.random.default_rng(5)
    series = np.clip(np.sin(np.arange(n + S - 1) / 21.0) + 0.1 * rng.standard_normal(n + S - 1), -1, 1)
    series[n // 2: n // 2 + 30] += 0.5
    return series


def _worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from hypad_amd import parallel as par
        from hypad_amd.anomaly_detection import score_batches
        from hypad_amd.models import tadgan
        from hypad_amd.utils import anomaly_detection_utils as adu
        calls = {}
        for name in ("broadcast", "all_gather_into_tensor", "all_reduce"):         # count what really goes through torch.distributed
            def wrap(fn, name=name):
                def inner(*a, **k):
                    calls[name] = calls.get(name, 0) + 1
                    return fn(*a, **k)
                return inner
            setattr(dist, name, wrap(getattr(dist, name)))
        S, n = 100, 433
        series = _windows(n, S)
        y = series[np.arange(n)[:, None] + np.arange(S)[None, :]]
        yd = torch.from_numpy(y).cuda()
        out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
        # ---- Euclidean branch (un-roll median + DTW / point), z-score by gather and by all-reduce
        torch.manual_seed(3)
        enc, dec, cx = [m.cuda().eval() for m in (tadgan.Encoder(S, 20), tadgan.Decoder(S, 20, False), tadgan.CriticX(S, 20))]
        before = [m.arena().clone() for m in (enc, dec, cx)]
        nbytes = par.broadcast_weights([enc, dec, cx], src=0)
        assert nbytes == sum(b.numel() * 4 for b in before) and all(torch.equal(m.arena(), b) for m, b in zip((enc, dec, cx), before))
        res = score_batches([yd[: n // 2], yd[n // 2:]], enc, dec, cx, S)
        for kind in ("dtw", "point"):
            want, _, _, _ = adu.score_anomalies(y, res["recons"], res["critic"], None, rec_error_type=kind, comb="mult")
            got = par.score_anomalies_sharded(yd, enc, dec, cx, S, rec_error_type=kind, comb="mult")
            assert np.array_equal(got, want, equal_nan=True), kind
            got_t = par.score_anomalies_sharded(yd, enc, dec, cx, S, rec_error_type=kind, comb="mult", as_tensor=True)
            assert got_t.is_cuda and np.array_equal(got_t.cpu().numpy(), want, equal_nan=True)
            red = par.score_anomalies_sharded(yd, enc, dec, cx, S, rec_error_type=kind, comb="mult", zscore="allreduce")
            assert np.allclose(red, want, rtol=1e-10, atol=1e-10, equal_nan=True), kind
        # ---- hyperbolic branch (row-wise Poincare distance + KDE critic modes), window matrix and series view
        torch.manual_seed(4)
        enc, dec, cx = [m.cuda().eval() for m in (tadgan.Encoder(S, 20), tadgan.Decoder(S, 20, True), tadgan.CriticX(S, 20))]
        with torch.no_grad():
            dec.hyperbolic_linear.weight.mul_(50)
        par.broadcast_weights([enc, dec, cx], src=0)
        x32 = yd.to(torch.float32).contiguous()
        res = score_batches([yd[: n // 3], yd[n // 3:]], enc, dec, cx, S)
        for comb in ("mult", "sum_uncertainty"):
            want = adu.hyperbolic_scores(res["recons"].cpu().numpy(), res["hyper_real"].cpu().numpy(), res["critic"].cpu().numpy(), S, comb)
            got_m = par.score_windows_sharded(x32, enc, dec, cx, S, comb)
            got_s = par.score_windows_sharded(torch.from_numpy(series).cuda().float().contiguous(), enc, dec, cx, S, comb, x_row_stride=1)
            # (the sharded scorer takes the distance the fused forward computed; hyperbolic_scores re-computes it from the written
            # reconstructions with the stand-alone kernel: same formula, possibly another summation order)
            assert np.array_equal(got_m, got_s), comb
            np.testing.assert_allclose(got_m, want, rtol=1e-6, atol=1e-9)
        # ---- the scorer call replayed as ONE hipGraph (collectives captured with it): same scores as the eager call, also after the
        # input was refilled in place
        series_d = torch.from_numpy(series).cuda().float().contiguous()
        eager = par.score_windows_sharded(series_d, enc, dec, cx, S, "mult", x_row_stride=1, as_tensor=True).clone()
        call = lambda: par.score_windows_sharded(series_d, enc, dec, cx, S, "mult", x_row_stride=1, as_tensor=True)
        rep = lambda: par.replay_scorer(call, series_d, enc.arena(), dec.arena(), cx.arena(), key="test")
        assert torch.equal(rep(), eager) and torch.equal(rep(), eager)
        series_d.copy_(torch.from_numpy(_windows(n, S)[::-1].copy()).cuda().float())
        eager2 = call().clone()
        assert not torch.equal(eager2, eager) and torch.equal(rep(), eager2)
        out["graph_replays"] = 3
        out["calls"] = dict(calls)
        ret[0] = out
    finally:
        dist.destroy_process_group()


def test_sharded_scorers_through_a_one_rank_rccl_group_equal_the_unsharded_pipeline():
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(_free_port(), ret), nprocs=1, join=True)
    out = ret[0]
    print("RCCL:", out)
    assert out["backend"] == "nccl" and out["world"] == 1 and out["graph_replays"] == 3
    assert out["calls"]["broadcast"] >= 6 and out["calls"]["all_gather_into_tensor"] >= 10 and out["calls"]["all_reduce"] >= 2
