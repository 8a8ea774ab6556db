"""Sharded Euclidean scoring on the device kernels (`parallel.score_anomalies_sharded`, the DTW leg of BASELINE.json
configs[4]): one rank == the un-sharded `score_anomalies` pipeline bit for bit; two ranks (two processes sharing the GPU,
gloo for the collectives) == one rank bit for bit; the weight broadcast."""
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


def _models(S, seed):
    from hypad_amd.models import tadgan
    torch.manual_seed(seed)
    enc, dec, cx = tadgan.Encoder(S, 20), tadgan.Decoder(S, 20, False), tadgan.CriticX(S, 20)
    return [m.cuda().eval() for m in (enc, dec, cx)]


def _windows(n, S):
    rng = np.random.default_rng(5)
    series = np.clip(np.sin(np.arange(n + S - 1) / 21.0) + 0.1 * rng.standard_normal(n + S - 1), -1, 1)
    series[n // 2: n // 2 + 30] += 0.5
    return series[np.arange(n)[:, None] + np.arange(S)[None, :]]


@pytest.mark.parametrize("kind,n,S", [("point", 700, 100), ("dtw", 433, 100), ("area", 350, 51), ("dtw", 64, 100)])
def test_one_rank_equals_unsharded_pipeline(kind, n, S):
    from hypad_amd import parallel as par
    from hypad_amd.utils import anomaly_detection_utils as adu
    enc, dec, cx = _models(S, 3)
    y = _windows(n, S)                                       # float64 windows, as the reference's dataset yields them
    yd = torch.from_numpy(y).cuda()
    from hypad_amd.anomaly_detection import score_batches       # the un-sharded test loop (anomaly_detection.py:67-113), 3 batches
    res = score_batches([yd[: n // 3], yd[n // 3: n // 2], yd[n // 2:]], enc, dec, cx, S)
    recon, critic = res["recons"], res["critic"]
    for comb in ("mult", "sum", "rec"):
        want, _, _, _ = adu.score_anomalies(y, recon, critic, None, rec_error_type=kind, comb=comb)
        got = par.score_anomalies_sharded(yd, enc, dec, cx, S, rec_error_type=kind, comb=comb)
        assert got.shape == (n + S - 1,)
        assert np.array_equal(got, want, equal_nan=True), (kind, comb)
    assert np.isnan(got).all() == (n < 100)


def _rank_worker(rank, world, port, kind, n, S, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hypad_amd import parallel as par
        enc, dec, cx = _models(S, 3 if rank == 0 else 99)      # rank 1 starts from OTHER weights: the broadcast must fix that
        nbytes = par.broadcast_weights([enc, dec, cx], src=0)
        assert nbytes == sum(m.arena().numel() * 4 for m in (enc, dec, cx))
        yd = torch.from_numpy(_windows(n, S)).cuda()
        out = {}
        for z in ("gather", "allreduce"):
            out[z] = par.score_anomalies_sharded(yd, enc, dec, cx, S, rec_error_type=kind, comb="mult", zscore=z)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,n,S", [("dtw", 433, 100), ("point", 1000, 100), ("dtw", 4_100, 100), ("point", 30_000, 100)])
def test_two_ranks_sharing_the_gpu_equal_one_rank(kind, n, S):
    import torch.multiprocessing as mp
    from hypad_amd import parallel as par
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_rank_worker, args=(2, port, kind, n, S, ret), nprocs=2, join=True)
    enc, dec, cx = _models(S, 3)
    one = par.score_anomalies_sharded(torch.from_numpy(_windows(n, S)).cuda(), enc, dec, cx, S, rec_error_type=kind, comb="mult")
    assert np.array_equal(ret[0]["gather"], one) and np.array_equal(ret[1]["gather"], one)
    assert np.array_equal(ret[0]["allreduce"], ret[1]["allreduce"])
    assert np.allclose(ret[0]["allreduce"], one, rtol=1e-10, atol=1e-10)
