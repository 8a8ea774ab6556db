"""Streams that really run beside each other.

The runtime deals streams onto a handful of hardware queues; two streams that share a queue run one after the other whatever the
program says (measured: a scoring pass whose side stream shared the main stream's queue, 1.05 ms instead of 0.95; two lanes of model
groups that shared one, no faster than one lane).  ``beside(streams, device)`` therefore tries candidates: a ~0.5 ms spin on each
stream it must not share a queue with, a trivial fill on the candidate -- the candidate is taken if its fill is done while the spin
still runs.  One-time cost ~1 ms per stream; never called while a capture is open (a graph's branches are parallel by construction)."""
import torch


def _runs_beside(cand, other, probe):
    other.synchronize()
    cand.synchronize()
    spun, filled = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(other):
        torch.cuda._sleep(1_000_000)
        spun.record()
    with torch.cuda.stream(cand):
        probe.fill_(1.0)
        filled.record()
    filled.synchronize()
    beside = not spun.query()
    other.synchronize()
    return beside


def beside(streams, device, tries=12):
    """A new stream whose work overlaps the work of every stream in ``streams`` (the best candidate found if none of ``tries`` does)."""
    first = torch.cuda.Stream(device=device)
    if torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, "_sleep") or not streams:
        return first
    probe = torch.empty(64, device=torch.device("cuda", device) if isinstance(device, int) else device)
    cand, best, best_n = first, first, -1
    for _ in range(tries):
        n = sum(1 for s in streams if _runs_beside(cand, s, probe))
        if n == len(streams):
            return cand
        if n > best_n:
            best, best_n = cand, n
        cand = torch.cuda.Stream(device=device)
    return best


def lanes(k, device):
    """``k`` streams that run beside one another (not necessarily beside the current stream, which idles while they work: with four
    hardware queues five mutually independent streams do not exist -- asked for, the fourth lane ended up behind another one)."""
    out = [torch.cuda.Stream(device=device)]
    while len(out) < k:
        out.append(beside(out, device))
    return out
