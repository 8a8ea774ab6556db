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


def global_normal_into(outs, chunk, rounds):
    """for r in range(rounds): for o in outs: o.flat[r*chunk:(r+1)*chunk] = float32(np.random.normal(size=chunk)) -- the order
    successive per-iteration calls draw in.  ``outs``: C-contiguous float32 arrays of at least rounds*chunk elements."""
    for o in outs:
        if o.dtype != np.float32 or not o.flags.c_contiguous or o.size < chunk * rounds:
            raise _C.HypadError("global_normal_into: float32 C-contiguous arrays of rounds * chunk elements")
    st = np.random.get_state()
    if not _native_ok(st):
        for r in range(rounds):
            for o in outs:
                o.reshape(-1)[r * chunk:(r + 1) * chunk] = np.random.normal(size=chunk)
        return
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos, has, cached = ctypes.c_int(int(st[2])), ctypes.c_int(int(st[3])), ctypes.c_double(float(st[4]))
    ptrs = (ctypes.c_void_p * len(outs))(*(o.ctypes.data for o in outs))
    _C.check(_C.lib.hypad_host_mt19937_normal(key.ctypes.data, ctypes.byref(pos), ctypes.byref(has), ctypes.byref(cached), ptrs, len(outs),
                                              int(chunk), int(rounds)), "host_mt19937_normal")
    np.random.set_state(("MT19937", key, pos.value, has.value, cached.value))
