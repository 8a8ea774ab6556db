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
