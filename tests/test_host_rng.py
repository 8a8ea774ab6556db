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


def test_torch_rand_continued_natively():
    """hypad_host_torch_mt19937_uniform == torch.rand on the default CPU generator, bit for bit, for every way the engine's block
    boundary can fall: a freshly seeded generator (no block generated yet), draws that end exactly on a boundary, across several
    blocks, interleaved with torch's own draws on both sides (uniform, normal -- whose cached sample must survive -- and integer),
    and an empty draw; the generator is left where torch.rand would have left it."""
    import torch
    from hypad_amd import host_rng
    for seed, sizes in ((0, [5, 619, 624, 1, 3000, 0, 7]), (1234, [624]), (7, [1248, 1]), (99, [100_003, 17])):
        torch.manual_seed(seed)
        want, mid = [], []
        for n in sizes:
            want.append(torch.rand(n))
            mid.append((torch.randn(3), torch.randint(0, 1000, (2,)), torch.rand(2, dtype=torch.float64)))
        end = torch.get_rng_state().clone()
        torch.manual_seed(seed)
        for n, w, m in zip(sizes, want, mid):
            got = host_rng.torch_rand_into(torch.empty(n))
            assert torch.equal(got, w), (seed, n)
            assert torch.equal(torch.randn(3), m[0]) and torch.equal(torch.randint(0, 1000, (2,)), m[1]) and torch.equal(torch.rand(2, dtype=torch.float64), m[2])
        assert torch.equal(torch.get_rng_state(), end), seed
    # a shaped, strided-free destination inside a larger buffer, as epoch_feed uses it
    torch.manual_seed(3)
    w = torch.rand(29 * (64 * 100 + 64 * 20))
    torch.manual_seed(3)
    buf = torch.empty(29 * (64 * 100 + 64 * 20))
    assert torch.equal(host_rng.torch_rand_into(buf), w)
    with pytest.raises(Exception):
        host_rng.torch_rand_into(torch.empty(4, dtype=torch.float64))


def test_threaded_normal_draw_equals_numpy():
    """hypad_host_mt19937_normal_mt (the generator on the calling thread, the transforms on helper threads, blocks of 32 768 pairs) == the
    per-iteration np.random.normal calls, bit for bit: several blocks, even and odd totals, a cached value going in and coming out, the
    generator's state afterwards."""
    old = host_rng.PIPELINE_FROM
    host_rng.PIPELINE_FROM = 1
    try:
        for seed, chunk, rounds, n_outs, warm, threads in ((1, 1280, 145, 2, 0, 3), (2, 77, 13, 3, 1, 2), (3, 5120, 29, 1, 3, 1), (4, 1, 200_001, 1, 0, 4), (5, 64, 3, 2, 1, 0)):
            np.random.seed(seed)
            np.random.normal(size=warm)                              # (an odd `warm` leaves a cached value)
            want = [np.empty(rounds * chunk, np.float32) for _ in range(n_outs)]
            for r in range(rounds):
                for o in want:
                    o[r * chunk:(r + 1) * chunk] = np.random.normal(size=(1, chunk))
            tail = np.random.normal(size=5)
            np.random.seed(seed)
            np.random.normal(size=warm)
            got = [np.full(rounds * chunk, np.nan, np.float32) for _ in range(n_outs)]
            host_rng.global_normal_into(got, chunk, rounds, threads=threads)
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (seed, chunk, rounds)
            assert np.array_equal(np.random.normal(size=5), tail), seed
    finally:
        host_rng.PIPELINE_FROM = old
