"""hypad_host_mt19937_normal (include/hypad.h; host code, no GPU): NumPy's global generator continued outside NumPy.
The drop-in train_tadgan draws an epoch's latent planes with it where the reference calls np.random.normal(size=(1, B, L))
once per iteration (train.py:24,118,205): same numbers bit for bit, same final generator state."""
import numpy as np
import pytest

from hypad_amd import host_rng


def _reference_epoch(nit, nb, B, L):
    zx = np.empty((nit, B * L), np.float32); zz = np.empty((nit, B * L), np.float32); zg = np.empty((nb, B * L), np.float32)
    for it in range(nit):
        zx[it] = np.random.normal(size=(1, B, L)).reshape(-1)            # critic_x_iteration, train.py:24
        zz[it] = np.random.normal(size=(1, B, L)).reshape(-1)            # critic_z_iteration, train.py:118
    for b in range(nb):
        zg[b] = np.random.normal(size=(1, B, L)).reshape(-1)             # decoder_iteration, train.py:205
    return zx, zz, zg


@pytest.mark.parametrize("seed, B, L, nb, nc, pre", [(0, 64, 20, 29, 5, 0), (7, 16, 20, 3, 2, 1), (123, 48, 7, 2, 5, 3), (5, 256, 20, 4, 1, 0)])
def test_epoch_planes_equal_per_iteration_numpy_draws(seed, B, L, nb, nc, pre):
    nit = nb * nc
    np.random.seed(seed)
    for _ in range(pre):
        np.random.normal()                                               # an odd number of earlier draws leaves a cached gaussian behind
    st0 = np.random.get_state()
    want = _reference_epoch(nit, nb, B, L)
    end_ref = np.random.get_state()
    np.random.set_state(st0)
    zx = np.zeros((nit, B * L), np.float32); zz = np.zeros_like(zx); zg = np.zeros((nb, B * L), np.float32)
    host_rng.global_normal_into([zx, zz], B * L, nit)
    host_rng.global_normal_into([zg], B * L, nb)
    for a, b in zip(want, (zx, zz, zg)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))      # bit for bit (signed zeros included)
    end = np.random.get_state()
    assert end[0] == end_ref[0] and np.array_equal(end[1], end_ref[1]) and end[2:] == end_ref[2:]
    # ... and the stream goes on as NumPy's own would
    nxt = np.random.normal(size=5)
    np.random.set_state(end_ref)
    assert np.array_equal(nxt, np.random.normal(size=5))


def test_many_refills_and_odd_chunks():
    np.random.seed(99)
    st0 = np.random.get_state()
    want = np.random.normal(size=200_001).astype(np.float32)
    end_ref = np.random.get_state()
    np.random.set_state(st0)
    out = np.zeros(200_001, np.float32)
    host_rng.global_normal_into([out], 200_001, 1)
    assert np.array_equal(want.view(np.uint32), out.view(np.uint32))
    end = np.random.get_state()
    assert np.array_equal(end[1], end_ref[1]) and end[2:] == end_ref[2:]


def test_other_bit_generators_fall_back_to_numpy(monkeypatch):
    """Only MT19937 state is continued natively; anything else (np.random.set_state cannot install another generator, but the
    guard is there) goes through np.random.normal itself: same contract."""
    monkeypatch.setattr(host_rng, "_native_ok", lambda st: False)
    np.random.seed(3)
    want = np.random.normal(size=(4, 6)).astype(np.float32)
    np.random.seed(3)
    a = np.zeros((2, 6), np.float32); b = np.zeros((2, 6), np.float32)
    host_rng.global_normal_into([a, b], 6, 2)
    assert np.array_equal(np.stack([a[0], b[0], a[1], b[1]]), want)
