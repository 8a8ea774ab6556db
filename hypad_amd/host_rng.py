"""NumPy's global generator, continued natively (include/hypad.h: hypad_host_mt19937_normal).

The reference draws its latent vectors with ``np.random.normal(size=(1, B, L))`` once per iteration (train.py:24,118,205):
319 calls and 408 320 values per epoch of the reference configuration, ~25 ns each with the interpreter lock held.
``global_normal_into`` produces the same values from the same process-wide state -- bit for bit, leaving the state where
those calls would have left it -- as float32 rows of caller-owned (pinned) arrays, without the lock, so the draws of an
epoch run on a helper thread next to the loader iteration and the ``torch.rand`` draws."""
import ctypes

import numpy as np

from . import _C


def _native_ok(state):
    return state[0] == "MT19937" and len(state[1]) == 624


PIPELINE_FROM = 1 << 20      # values per call from which the transforms run on helper threads (hypad_host_mt19937_normal_mt)


def global_normal_into(outs, chunk, rounds, threads=0):
    """for r in range(rounds): for o in outs: o.flat[r*chunk:(r+1)*chunk] = float32(np.random.normal(size=chunk)) -- the order
    successive per-iteration calls draw in.  ``outs``: C-contiguous float32 arrays of at least rounds*chunk elements.
    ``threads``: helper threads for draws of ``PIPELINE_FROM`` values and more (the generator stays on the calling thread; same values)."""
    for o in outs:
        if o.dtype != np.float32 or not o.flags.c_contiguous or o.size < chunk * rounds:
            raise _C.HypadError("global_normal_into: float32 C-contiguous arrays of rounds * chunk elements")
    if chunk * rounds * len(outs) < PIPELINE_FROM:
        threads = 0
    st = np.random.get_state()
    if not _native_ok(st):
        for r in range(rounds):
            for o in outs:
                o.reshape(-1)[r * chunk:(r + 1) * chunk] = np.random.normal(size=chunk)
        return
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos, has, cached = ctypes.c_int(int(st[2])), ctypes.c_int(int(st[3])), ctypes.c_double(float(st[4]))
    ptrs = (ctypes.c_void_p * len(outs))(*(o.ctypes.data for o in outs))
    _C.check(_C.lib.hypad_host_mt19937_normal_mt(key.ctypes.data, ctypes.byref(pos), ctypes.byref(has), ctypes.byref(cached), ptrs, len(outs),
                                                 int(chunk), int(rounds), int(threads)), "host_mt19937_normal")
    np.random.set_state(("MT19937", key, pos.value, has.value, cached.value))


TORCH_STATE_BYTES = 5056      # at::CPUGeneratorImplState (an at::mt19937 + the normal sampler's caches)


def torch_rand_into(out):
    """``out[...] = torch.rand(out.shape)`` on torch's DEFAULT CPU generator -- the same float32 values, the generator left where that
    call would have left it (include/hypad.h: hypad_host_torch_mt19937_uniform) -- at ~3x the rate of torch's serial kernel and
    without the interpreter lock.  ``out``: a contiguous float32 CPU tensor.  Falls back to ``torch.rand(out=)`` if the generator's
    state is not the layout this was written for."""
    import torch
    if out.dtype != torch.float32 or not out.is_contiguous() or out.is_cuda:
        raise _C.HypadError("torch_rand_into: a contiguous float32 CPU tensor")
    state = torch.get_rng_state()
    if state.numel() != TORCH_STATE_BYTES or state.dtype != torch.uint8 or not state.is_contiguous():
        torch.rand(out.shape, out=out)
        return out
    rc = _C.lib.hypad_host_torch_mt19937_uniform(state.data_ptr(), TORCH_STATE_BYTES, out.data_ptr(), out.numel())
    if rc != 0:                                   # (an unseeded / foreign engine state: let torch do it)
        torch.rand(out.shape, out=out)
        return out
    torch.set_rng_state(state)
    return out
