"""Test-time driver (reference: anomaly_detection.py:20-155): eval-mode batch loop fused into one kernel per
batch -- encoder -> decoder -> hyperbolic_linear(sample) -> critic_x(sample) [-> row-wise Poincare distance]."""
import numpy as np
import torch

from . import _C


def score_batches(test_loader, encoder, decoder, critic_x, signal_shape):
    """The loop body of anomaly_detection.py:67-113.  Returns a dict of device tensors:
    recons (N,S) [hyperbolic output or tanh output], eucl (N,S), hyper_real (N,S), critic (N,), rowdist (N,), true (N,S[,1])."""
    encoder.eval(); decoder.eval(); critic_x.eval()
    hyp = bool(decoder.hyperbolic)
    S, L = signal_shape, encoder.latent_space_dim
    outs = {k: [] for k in ("recons", "eucl", "hyper_real", "critic", "rowdist", "true")}
    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, int(hyp))
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=torch.float32, device="cuda")           # packed weight copies (built by the call)
    for batch in test_loader:
        sample = batch[0] if isinstance(batch, (list, tuple)) else batch
        x = sample.reshape(-1, S).to("cuda", torch.float32).contiguous()
        n = x.shape[0]
        new = lambda *s: torch.empty(*s, device=x.device, dtype=torch.float32)
        eucl, critic = new(n, S), new(n)
        hyper, hreal, dist = (new(n, S), new(n, S), new(n)) if hyp else (None, None, None)
        _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(encoder.arena()), _C.ptr(decoder.arena()), _C.ptr(critic_x.arena()), _C.ptr(x), 0,
                                                   _C.ptr(hyper), _C.ptr(eucl), _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist), n, S, L,
                                                   int(hyp), ws.data_ptr(), ws_bytes, _C.stream()), "score_forward_packed")
        outs["recons"].append(hyper if hyp else eucl)
        outs["eucl"].append(eucl)
        outs["critic"].append(critic)
        outs["true"].append(sample)
        if hyp:
            outs["hyper_real"].append(hreal)
            outs["rowdist"].append(dist)
    res = {k: (torch.cat(v) if v else None) for k, v in outs.items()}
    return res


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
