"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL.

The reference is single-process; what shards is the *unit of work* it already treats independently:

* training -- one model per signal (train.py:428-437): rank r owns signals {s : s mod world == r}; nothing is
  exchanged while training (no gradient all-reduce exists, each signal has its own weights).  Only end-of-run
  metrics are gathered.
* scoring -- windows are independent through the networks; the un-roll over anti-diagonals needs a halo of
  (window - 1) windows at each boundary, which is re-computed locally instead of exchanged; the global z-score
  needs one all-reduce of (sum, sum of squares, count).
"""
import torch
import torch.distributed as dist


def signals_of_rank(n_signals, world, rank):
    """Round-robin ownership of signals."""
    return list(range(rank, n_signals, world))


def window_range(n_windows, world, rank):
    """Contiguous, balanced [begin, end) window range of a rank."""
    base, extra = divmod(n_windows, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def window_range_with_halo(n_windows, world, rank, window):
    """Range a rank must *evaluate* so that every un-rolled timestep it owns sees all its contributing windows
    (timestep t gathers y_hat[t - j, j], j < window: utils/anomaly_detection_utils.py:918-921)."""
    b, e = window_range(n_windows, world, rank)
    return max(0, b - (window - 1)), e


def timestep_range(n_windows, world, rank, window):
    """Un-rolled timesteps owned by a rank: those whose newest contributing window is in its window range
    (the last rank also owns the tail of the final window)."""
    b, e = window_range(n_windows, world, rank)
    return b, (e if rank < world - 1 else n_windows + window - 1)


def global_zscore_stats(local_sum, local_sumsq, local_count, group=None):
    """Mean and population std over all ranks from one all-reduce of (sum, sum sq, count) (stats.zscore, ddof=0)."""
    t = torch.tensor([local_sum, local_sumsq, local_count], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t = t.cpu()
    s, ss, n = t.tolist()
    mean = s / n
    var = max(ss / n - mean * mean, 0.0)
    return mean, var ** 0.5


def gather_signal_metrics(local, group=None):
    """{signal_id: metrics} from every rank -> merged dict on every rank (end-of-run only; KB-sized)."""
    if not (dist.is_available() and dist.is_initialized()):
        return dict(local)
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, dict(local), group=group)
    merged = {}
    for d in out:
        merged.update(d)
    return merged


# ---------------------------------------------------------------------------------------------- sharded scoring (configs[4])
def _all_gather_ranges(local, total, ranges, group=None):
    """Every rank contributes the 1-D tensor `local` = its [begin, end) slice of a vector of `total` elements
    (`ranges[r]` = rank r's slice); returns the whole vector on every rank.  One all-gather of equal-sized (padded) pieces."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    width = max(e - b for b, e in ranges)
    piece = torch.zeros(width, dtype=local.dtype, device=local.device)
    piece[: local.numel()] = local
    out = torch.empty(world * width, dtype=local.dtype, device=local.device)
    if local.is_cuda:
        dist.all_gather_into_tensor(out, piece, group=group)
    else:                                                                   # gloo (the CPU tests)
        parts = [torch.empty_like(piece) for _ in range(world)]
        dist.all_gather(parts, piece, group=group)
        out = torch.cat(parts)
    full = torch.cat([out[r * width: r * width + (e - b)] for r, (b, e) in enumerate(ranges)])
    assert full.numel() == total
    return full


def sharded_hyperbolic_scores(n_windows, window, evaluate, kde_modes, finish, need_norms=False, group=None):
    """Window scores of the hyperbolic branch (utils/anomaly_detection_utils.py:54-86) with the windows split over the ranks
    (BASELINE.json configs[4]).  Every rank gets the full result.

    * rank r evaluates windows [b - (window - 1), e): its own range plus the halo whose critic values reach its un-rolled
      timesteps -- re-computed locally, not exchanged.  ``evaluate(lo, hi)`` -> dict of 1-D tensors over those windows:
      ``rowdist`` (row-wise Poincare distance), ``critic``, and ``norms`` (||recons||_2, only if ``need_norms``);
    * ``kde_modes(critic, window)`` -> the KDE mode of every un-rolled timestep of the evaluated windows
      (:374-400); the rank keeps the timesteps it owns;
    * one all-gather each of the (N,) distances [and norms] and the (N + window - 1,) modes -- 4-8 MB at 10^6 windows;
    * ``finish(rowdist, modes, norms)`` -> final scores: the global steps (quantile-trimmed z-score of the modes, rolling
      mean, combination) on the full vectors, exactly the unsharded code, so the result does not depend on the world size.
    """
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    b, e = window_range(n_windows, world, rank)
    hb, he = window_range_with_halo(n_windows, world, rank, window)
    tb, te = timestep_range(n_windows, world, rank, window)
    if e > b or rank == world - 1:
        got = evaluate(hb, he)
        modes = kde_modes(got["critic"], window)[tb - hb: te - hb]          # local timestep k == global hb + k
        rowdist = got["rowdist"][b - hb:]
        norms = got["norms"][b - hb:] if need_norms else None
    else:                                                                   # more ranks than windows: nothing owned
        ref = evaluate(0, min(1, n_windows))
        modes, rowdist = ref["critic"][:0].to(torch.float64), ref["rowdist"][:0]
        norms = ref["norms"][:0] if need_norms else None
    wr = [window_range(n_windows, world, r) for r in range(world)]
    tr = [timestep_range(n_windows, world, r, window) for r in range(world)]
    rowdist = _all_gather_ranges(rowdist.contiguous(), n_windows, wr, group)
    modes = _all_gather_ranges(modes.contiguous(), n_windows + window - 1, tr, group)
    if need_norms:
        norms = _all_gather_ranges(norms.contiguous(), n_windows, wr, group)
    return finish(rowdist, modes, norms)


def score_windows_sharded(x, encoder, decoder, critic_x, signal_shape, combination="mult", group=None, x_row_stride=0,
                          n_windows=None):
    """``sharded_hyperbolic_scores`` on the device kernels: ``x`` is the (N, S) fp32 window matrix -- or, with
    ``x_row_stride=1``, the scaled series whose window n is x[n : n + S] -- resident on every rank, like the weights
    (broadcast them once with ``torch.distributed.broadcast(module.arena(), 0)`` if the ranks did not load the same
    checkpoint).  Returns the final scores, (N,) float64 NumPy, on every rank."""
    import math
    from . import _C
    from .hyperspace import gmath
    from .utils import anomaly_detection_utils as adu
    if not decoder.hyperbolic:
        raise ValueError("score_windows_sharded: the Euclidean branch un-rolls reconstructions; shard it with timestep_range()")
    S, L = signal_shape, encoder.latent_space_dim
    N = n_windows if n_windows is not None else (x.shape[0] if x_row_stride == 0 else x.numel() - S + 1)
    need_norms = "uncertainty" in combination
    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device=x.device)
    encoder.eval(); decoder.eval(); critic_x.eval()

    def evaluate(lo, hi):
        n = hi - lo
        new = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
        hyper, eucl, hreal, critic, dist_ = new(n, S), new(n, S), new(n, S), new(n), new(n)
        xs = x[lo:hi] if x_row_stride == 0 else x[lo * x_row_stride:]
        _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(encoder.arena()), _C.ptr(decoder.arena()), _C.ptr(critic_x.arena()), _C.ptr(xs),
                                                   x_row_stride, _C.ptr(hyper), _C.ptr(eucl), _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist_),
                                                   n, S, L, 1, ws.data_ptr(), ws_bytes, _C.stream()), "score_forward_packed")
        out = {"rowdist": gmath.poincare_rowdist(hreal, hyper), "critic": critic}
        if need_norms:
            out["norms"] = adu.row_norms(hyper)
        return out

    def finish(rowdist, modes, norms):
        critic_scores = []
        if combination in ("mult", "uncertainty", "sum", "sum_uncertainty", "critic", "critic_uncertainty"):
            critic_scores = adu._compute_critic_score(modes, math.trunc(N * 0.01))[:N]
        return adu.combine_scores(combination, critic_scores, rowdist, norms=norms)

    return sharded_hyperbolic_scores(N, S, evaluate, adu.kde_modes, finish, need_norms, group)
