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
        idx = [i for b in loader_batches(test_loader) for i in b]
        ordered = idx == list(range(len(matrix)))
        true = matrix if ordered else matrix[torch.as_tensor(idx, dtype=torch.long)]
        if not len(idx):
            return {k: None for k in ("recons", "eucl", "hyper_real", "critic", "rowdist", "true")}
        ds = test_loader.dataset
        shape = tuple(np.asarray(ds.X).shape[1:]) if hasattr(ds, "X") else tuple(ds.shape[1:])
        return score_windows(true.reshape((len(idx),) + shape), encoder, decoder, critic_x, S, L, hyp)
    samples = []
    for batch in test_loader:
        samples.append(batch[0] if isinstance(batch, (list, tuple)) else batch)
    if not samples:
        return {k: None for k in ("recons", "eucl", "hyper_real", "critic", "rowdist", "true")}
    true = torch.cat(samples) if len(samples) > 1 else samples[0]
    return score_windows(true, encoder, decoder, critic_x, S, L, hyp)


def score_windows(true, encoder, decoder, critic_x, S, L, hyp):
    """One fused test-loop forward over all windows of `true` ((N, S[, 1]) host or device tensor)."""
    x = true.reshape(-1, S).to("cuda", torch.float32, non_blocking=True).contiguous()
    n = x.shape[0]
    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, int(hyp))
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device="cuda")           # packed weight copies (built by the call)
    new = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
    eucl, critic = new(n, S), new(n)
    hyper, hreal, dist = (new(n, S), new(n, S), new(n)) if hyp else (None, None, None)
    _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(encoder.arena()), _C.ptr(decoder.arena()), _C.ptr(critic_x.arena()), _C.ptr(x), 0,
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
    recons_signal = res["recons"].cpu().numpy()
    gt_signal = res["true"].cpu().numpy() if isinstance(res["true"], torch.Tensor) else np.concatenate(res["true"])
    critic_score = list(res["critic"].cpu().numpy())
    true_signal = gt_signal
    if path:
        torch.save(recons_signal, path + "recons_signal.pt")
        torch.save(gt_signal, path + "gt_signal.pt")
        torch.save(critic_score, path + "critic_score.pt")
    if decoder.hyperbolic:
        true_signal = res["hyper_real"].cpu().numpy()
        if path:
            torch.save(res["eucl"].cpu().numpy(), path + "eucl_recons.pt")
            torch.save(true_signal, path + "real_hyper.pt")
    return recons_signal, true_signal, critic_score
