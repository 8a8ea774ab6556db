"""Test-time driver (reference: anomaly_detection.py:20-155): eval-mode batch loop fused into one kernel per
batch -- encoder -> decoder -> hyperbolic_linear(sample) -> critic_x(sample) [-> row-wise Poincare distance]."""
import numpy as np
import torch

from . import _C


def score_batches(test_loader, encoder, decoder, critic_x, signal_shape):
    """The loop of anomaly_detection.py:67-113 as ONE launch: the loader is iterated exactly as the reference iterates it, but its
    batches are only collected; the windows then go to the device in one copy and through the fused forward in one call (one
    weight-pack launch + one forward launch for the whole test set instead of a pack, a 4-workgroup forward and five allocations
    per batch of 64).  The rows of a fused forward do not depend on each other, so the results are those of the batch-by-batch
    loop bit for bit (tests/test_gpu_dropin_r4.py); the reference's special case for a last batch of one window
    (anomaly_detection.py:76-88, 98-104: it only works around squeeze()) needs no counterpart.
    Returns a dict of device tensors: recons (N,S) [hyperbolic output or tanh output], eucl (N,S), hyper_real (N,S), critic (N,),
    rowdist (N,), true (N,S[,1])."""
    encoder.eval(); decoder.eval(); critic_x.eval()
    hyp = bool(decoder.hyperbolic)
    S, L = signal_shape, encoder.latent_space_dim
    from .epoch_feed import _index_matrix, loader_batches
    matrix = _index_matrix(test_loader, test=True)
    if matrix is not None:
        # a plain DataLoader over one of hypad_amd's datasets (or a tensor): its batches are rows of the dataset's window matrix --
        # take the rows, in the sampler's order, without fetching and collating 64 item tuples per batch (a test item carries the
        # signal's whole index and target arrays, utils/dataloader.py:229-231: ~4 MB of collation per batch of 64)
        n_seq = _sequential_rows(test_loader, len(matrix))
        if n_seq is not None:                       # DataLoader(shuffle=False): rows 0 .. n in order, no need to enumerate them
            next(loader_batches(test_loader), None)                          # (the iterator's base-seed draw)
            idx, ordered = range(n_seq), n_seq == len(matrix)
            true = matrix[:n_seq]
        else:
            idx = [i for b in loader_batches(test_loader) for i in b]
            ordered = idx == list(range(len(matrix)))
            true = matrix if ordered else matrix[torch.as_tensor(idx, dtype=torch.long)]
        if not len(idx):
            return {k: None for k in ("recons", "eucl", "hyper_real", "critic", "rowdist", "true")}
        ds = test_loader.dataset
        shape = tuple(np.asarray(ds.X).shape[1:]) if hasattr(ds, "X") else tuple(ds.shape[1:])
        true = true.reshape((len(idx),) + shape)
        view = ds.series_windows() if ordered and hasattr(ds, "series_windows") else None
        if view is not None and view[1] == len(idx) and shape == (S, 1):
            # the windows of a univariate signal, in order, are overlapping rows of its scaled series: the T + S - 1 values go to
            # the device instead of the T x S float64 matrix (0.5 MB instead of 100 MB at 125 000 windows) and the forward reads
            # window n at series + n -- the same float32 values in the same places, so the same results bit for bit
            return score_windows(true, encoder, decoder, critic_x, S, L, hyp, series=view[0])
        return score_windows(true, encoder, decoder, critic_x, S, L, hyp)
    samples = []
    for batch in test_loader:
        samples.append(batch[0] if isinstance(batch, (list, tuple)) else batch)
    if not samples:
        return {k: None for k in ("recons", "eucl", "hyper_real", "critic", "rowdist", "true")}
    true = torch.cat(samples) if len(samples) > 1 else samples[0]
    return score_windows(true, encoder, decoder, critic_x, S, L, hyp)


def _sequential_rows(loader, n):
    """Rows one pass over `loader` yields when they are 0 .. rows-1 in order -- torch's own BatchSampler over its SequentialSampler,
    what DataLoader(shuffle=False) builds -- else None."""
    from torch.utils.data import BatchSampler, SequentialSampler
    bs = loader.batch_sampler
    if type(bs) is not BatchSampler or type(bs.sampler) is not SequentialSampler or len(bs.sampler) != n:
        return None
    return n - n % bs.batch_size if bs.drop_last else n


def score_windows(true, encoder, decoder, critic_x, S, L, hyp, series=None):
    """One fused test-loop forward over all windows of `true` ((N, S[, 1]) host or device tensor).  ``series``: a (N + S - 1,) device
    tensor whose overlapping rows ARE those windows (x_row_stride = 1 of hypad_score_forward_packed) -- `true` is then not uploaded."""
    if series is not None:
        x, n, stride = series, true.shape[0], 1
        assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() >= n + S - 1
    else:
        x, stride = true.reshape(-1, S).to("cuda", torch.float32, non_blocking=True).contiguous(), 0
        n = x.shape[0]
    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, int(hyp))
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device="cuda")           # packed weight copies (built by the call)
    new = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
    eucl, critic = new(n, S), new(n)
    hyper, hreal, dist = (new(n, S), new(n, S), new(n)) if hyp else (None, None, None)
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(encoder.arena()), _C.ptr(decoder.arena()), _C.ptr(critic_x.arena()), _C.ptr(x), stride,
                                               _C.ptr(hyper), _C.ptr(eucl), _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist), n, S, L,
                                               int(hyp), ws.data_ptr(), ws_bytes, _C.stream()), "score_forward_packed")
    return {"recons": hyper if hyp else eucl, "eucl": eucl, "hyper_real": hreal, "critic": critic, "rowdist": dist, "true": true}


def score_batches_per_batch(test_loader, encoder, decoder, critic_x, signal_shape):
    """The batch-by-batch form of score_batches (one pack + one forward launch per loader batch, as anomaly_detection.py:67-113 is
    written): the reference point of the one-call form's test and of bench.py's drop_in.scoring."""
    encoder.eval(); decoder.eval(); critic_x.eval()
    hyp = bool(decoder.hyperbolic)
    S, L = signal_shape, encoder.latent_space_dim
    keys = ("recons", "eucl", "hyper_real", "critic", "rowdist", "true")
    outs = {k: [] for k in keys}
    for batch in test_loader:
        sample = batch[0] if isinstance(batch, (list, tuple)) else batch
        r = score_windows(sample, encoder, decoder, critic_x, S, L, hyp)
        for k in keys:
            if r[k] is not None:
                outs[k].append(r[k])
    return {k: (torch.cat(v) if v else None) for k, v in outs.items()}


def test_tadgan(test_loader, encoder, decoder, critic_x, read_path="", signal="", path="", signal_shape=100, params=[]):
    """anomaly_detection.py:20-155 up to the hand-off to the scoring utilities.  Writes the same cache files
    (recons_signal.pt, gt_signal.pt, critic_score.pt [, eucl_recons.pt, real_hyper.pt]) and returns
    (recons_signal, true_signal, critic_score) as the reference passes them on."""
    path = path + "/" if path else ""
    res = score_batches(test_loader, encoder, decoder, critic_x, signal_shape)
    # results back through page-locked buffers, all copies queued before the one wait (a pageable .cpu() of an (N, S) array is
    # staged by the driver in pieces: 4-5 ms per array at 125 000 windows); the arrays returned are views of those buffers
    want = {"recons": res["recons"], "critic": res["critic"]}
    if isinstance(res["true"], torch.Tensor) and res["true"].is_cuda:
        want["true"] = res["true"]
    if decoder.hyperbolic:
        want["hyper_real"] = res["hyper_real"]
        if path:
            want["eucl"] = res["eucl"]
    host = _to_host(want)
    recons_signal = host["recons"]
    gt_signal = host["true"] if "true" in host else (res["true"].numpy() if isinstance(res["true"], torch.Tensor) else np.concatenate(res["true"]))
    critic_score = list(host["critic"])
    true_signal = gt_signal
    if path:
        torch.save(recons_signal, path + "recons_signal.pt")
        torch.save(gt_signal, path + "gt_signal.pt")
        torch.save(critic_score, path + "critic_score.pt")
    if decoder.hyperbolic:
        true_signal = host["hyper_real"]
        if path:
            torch.save(host["eucl"], path + "eucl_recons.pt")
            torch.save(true_signal, path + "real_hyper.pt")
    return recons_signal, true_signal, critic_score


def _to_host(tensors):
    """{name: device tensor} -> {name: NumPy array}: one page-locked buffer per tensor, the copies queued back to back, one wait."""
    out = {k: torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for k, t in tensors.items()}
    for k, t in tensors.items():
        out[k].copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return {k: v.numpy() for k, v in out.items()}
