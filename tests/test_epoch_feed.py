"""hypad_amd/epoch_feed.py on the CPU: an epoch's host random numbers and minibatches staged ahead of the launch are exactly what
the reference's loop (train.py:299-356) would have drawn / fetched call by call -- values, order, and the state both global
generators and the loader are left in."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from hypad_amd.epoch_feed import DEPTH, EpochFeed, _index_matrix


class Windows:
    """Map-style dataset in hypad_amd's module namespace shape: __getitem__(i) is X[i] (utils/dataloader.py:227-232)."""
    __module__ = "hypad_amd.tests_fixture"

    def __init__(self, n, S, seed=0):
        self.X = np.random.default_rng(seed).uniform(-1, 1, (n, S, 1))
        self.test = False

    def __len__(self):
        return len(self.X)

    def __getitem__(self, i):
        return torch.from_numpy(self.X[i])


def reference_epoch(loader, B, S, L, nc):
    """The draws and batches of one epoch of train.py:315-352, call by call."""
    zx, ax, zz, az, zg, xs = [], [], [], [], [], []
    for _ in range(nc):
        for sample in loader:
            xs.append(sample.reshape(B, S).float())
            zx.append(torch.Tensor(np.random.normal(size=(1, B, L))).reshape(-1))      # critic_x_iteration: train.py:24, :64
            ax.append(torch.rand((1, B, S)).reshape(-1))
            zz.append(torch.Tensor(np.random.normal(size=(1, B, L))).reshape(-1))      # critic_z_iteration: train.py:118, :149
            az.append(torch.rand((1, B, L)).reshape(-1))
    for sample in loader:
        xs.append(sample.reshape(B, S).float())
        zg.append(torch.Tensor(np.random.normal(size=(1, B, L))).reshape(-1))          # decoder_iteration: train.py:205
    cat = lambda v: torch.cat(v)
    return dict(z_cx=cat(zx), alpha_cx=cat(ax), z_cz=cat(zz), alpha_cz=cat(az), z_gen=cat(zg)), torch.cat(xs)


def gen_states():
    s = np.random.get_state()
    return (s[1].copy(), s[2], s[3], s[4]), torch.get_rng_state().clone()


def same_states(a, b):
    return np.array_equal(a[0][0], b[0][0]) and a[0][1:] == b[0][1:] and torch.equal(a[1], b[1])


@pytest.mark.parametrize("B, S, L, n, nc, workers", [(64, 100, 20, 1916, 5, 0), (16, 33, 7, 100, 2, 0), (48, 100, 20, 200, 3, 0)])
@pytest.mark.parametrize("index_path", [True, False])
def test_staged_epochs_equal_the_call_by_call_loop(B, S, L, n, nc, workers, index_path):
    ds = Windows(n, S)
    loader = DataLoader(ds, batch_size=B, drop_last=True, shuffle=True, num_workers=workers)
    assert (_index_matrix(loader) is not None)
    epochs = 3
    np.random.seed(11); torch.manual_seed(11)
    want = [reference_epoch(loader, B, S, L, nc) for _ in range(epochs)]
    end_ref = gen_states()
    np.random.seed(11); torch.manual_seed(11)
    feed = EpochFeed(loader, B, S, L, nc, "cpu", index_path=index_path)
    assert feed.index_path == index_path
    feed.last_epoch = epochs - 1
    for e in range(epochs):
        slot = feed.prepare(e)
        assert slot == e % DEPTH
        feed.upload(slot)
        planes, xs = want[e]
        for k, v in planes.items():
            assert torch.equal(feed.noise[k], v), (e, k)
        rows = feed.x[feed.row_index.reshape(-1).long()]
        assert torch.equal(rows, xs), e
    feed.close()
    assert same_states(gen_states(), end_ref)          # nothing drawn ahead of the last epoch; both streams where the loop leaves them


def test_lists_of_host_batches_and_plain_tensors():
    B, S, L, nc = 16, 20, 5, 2
    data = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (4 * B, S, 1)))
    batches = [data[i * B:(i + 1) * B] for i in range(4)]                  # what bench.py's drop_in hands over
    np.random.seed(3); torch.manual_seed(3)
    want, xs = reference_epoch(batches, B, S, L, nc)
    end_ref = gen_states()
    np.random.seed(3); torch.manual_seed(3)
    feed = EpochFeed(batches, B, S, L, nc, "cpu")
    assert not feed.index_path
    feed.last_epoch = 0
    feed.upload(feed.prepare(0))
    feed.close()
    for k, v in want.items():
        assert torch.equal(feed.noise[k], v), k
    assert torch.equal(feed.x[feed.row_index.reshape(-1).long()], xs)
    assert same_states(gen_states(), end_ref)
    # a DataLoader straight over a tensor takes the index path
    loader = DataLoader(data, batch_size=B, drop_last=True, shuffle=True)
    np.random.seed(4); torch.manual_seed(4)
    want, xs = reference_epoch(loader, B, S, L, nc)
    np.random.seed(4); torch.manual_seed(4)
    feed = EpochFeed(loader, B, S, L, nc, "cpu")
    assert feed.index_path
    feed.last_epoch = 0
    feed.upload(feed.prepare(0))
    feed.close()
    assert torch.equal(feed.x[feed.row_index.reshape(-1).long()], xs) and all(torch.equal(feed.noise[k], v) for k, v in want.items())


def test_loaders_the_index_path_must_not_take():
    ds = Windows(64, 10)
    assert _index_matrix(DataLoader(ds, batch_size=16, collate_fn=lambda b: torch.stack(b))) is None      # custom collate
    assert _index_matrix([torch.zeros(16, 10)]) is None
    ds.test = True                                                                                          # test datasets yield tuples
    assert _index_matrix(DataLoader(ds, batch_size=16)) is None
    class Augmenting(Windows):                    # somebody else's class whose items are NOT the rows of X: its batches are fetched and staged
        __module__ = "somewhere.else"
        def __getitem__(self, i):
            return torch.from_numpy(self.X[i] + 0.01 * np.random.standard_normal(self.X[i].shape))
    class Scaling(Augmenting):
        def __getitem__(self, i):
            return torch.from_numpy(self.X[i] * 2.0)
    st = gen_states()
    assert _index_matrix(DataLoader(Augmenting(64, 10), batch_size=16)) is None
    assert _index_matrix(DataLoader(Scaling(64, 10), batch_size=16)) is None
    assert same_states(st, gen_states())                                                                   # (the probe put the generators back)


def test_a_foreign_dataset_built_like_the_references_takes_the_index_path():
    """What a user who swaps only train.py passes in: the REFERENCE's own SignalDataset (utils/dataloader.py:61-232) -- windows in ``X``,
    ``__getitem__(i)`` = ``torch.from_numpy(X[i])`` (test mode: a tuple that starts with it).  A class hypad_amd has never seen whose
    sampled items equal its rows bit for bit is read by index like hypad_amd's own; the staged planes equal the fetched-and-collated
    ones, and probing the dataset leaves the global generators where they were."""
    class TheirDataset:
        __module__ = "utils.dataloader"
        def __init__(self, n, S, test=False):
            t = np.arange(n + S - 1)
            series = np.sin(t / 7.0)
            self.X = series[np.arange(n)[:, None] + np.arange(S)[None, :]][:, :, None].copy()
            self.test, self.index = test, np.arange(n)
        def __len__(self):
            return len(self.X)
        def __getitem__(self, i):
            x = torch.from_numpy(self.X[i])
            return (x, self.index, 0, 0, 0) if self.test else x
    B, S, L, nc = 16, 10, 4, 2
    st = gen_states()
    m = _index_matrix(DataLoader(TheirDataset(70, S), batch_size=B, shuffle=True, drop_last=True))
    assert m is not None and m.shape == (70, S) and same_states(st, gen_states())
    assert _index_matrix(DataLoader(TheirDataset(70, S, test=True), batch_size=B)) is None                 # (a test dataset under the training loop: no)
    assert _index_matrix(DataLoader(TheirDataset(70, S, test=True), batch_size=B), test=True) is not None
    planes = []
    for ip in (True, False):
        np.random.seed(9); torch.manual_seed(9)
        feed = EpochFeed(DataLoader(TheirDataset(70, S), batch_size=B, shuffle=True, drop_last=True), B, S, L, nc, "cpu", index_path=ip)
        assert feed.index_path == ip
        feed.last_epoch = 0
        feed.upload(feed.prepare(0))
        feed.close()
        x = feed.x[feed.row_index.reshape(-1).long()] if ip else feed.x
        planes.append((x.reshape(-1, S).clone(), {k: v.clone() for k, v in feed.noise.items()}))
    assert torch.equal(planes[0][0], planes[1][0]) and all(torch.equal(planes[0][1][k], planes[1][1][k]) for k in planes[0][1])


def test_short_last_batch_is_refused():
    from hypad_amd._C import HypadError
    ds = Windows(40, 10)
    for ip in (True, False):
        feed = EpochFeed(DataLoader(ds, batch_size=16, shuffle=True), 16, 10, 4, 1, "cpu", index_path=ip)
        feed.last_epoch = 0
        with pytest.raises(HypadError, match="drop_last"):
            feed.prepare(0)
        feed.close()


@pytest.mark.parametrize("shuffle", [False, True])
def test_loader_batches_draws_what_the_iterator_draws(shuffle):
    """The test loop's index path (anomaly_detection.score_batches): same batches, same generator state as a real pass -- the
    iterator draws a base seed even when nothing is shuffled."""
    from hypad_amd.epoch_feed import loader_batches
    ds = Windows(150, 12)
    loader = DataLoader(ds, batch_size=64, drop_last=False, shuffle=shuffle)
    torch.manual_seed(5)
    real = [b.clone() for b in loader]
    end = torch.get_rng_state().clone()
    torch.manual_seed(5)
    idx = list(loader_batches(loader))
    assert torch.equal(torch.get_rng_state(), end)
    assert [len(b) for b in idx] == [64, 64, 22]
    for b, r in zip(idx, real):
        assert torch.equal(torch.from_numpy(ds.X[b]), r)


@pytest.mark.parametrize("index_path", [True, False])
def test_epochs_staged_ahead_by_the_producer_thread_equal_the_loop(index_path):
    """EpochFeed.get: epoch 0 on the caller's thread, then a producer thread one to two epochs ahead -- the same planes, batches and
    final generator states as the call-by-call loop; abandoning the feed mid-way joins cleanly."""
    B, S, L, n, nc, epochs = 16, 33, 7, 100, 2, 6
    ds = Windows(n, S)
    loader = DataLoader(ds, batch_size=B, drop_last=True, shuffle=True)
    np.random.seed(21); torch.manual_seed(21)
    want = [reference_epoch(loader, B, S, L, nc) for _ in range(epochs)]
    end_ref = gen_states()
    np.random.seed(21); torch.manual_seed(21)
    feed = EpochFeed(loader, B, S, L, nc, "cpu", index_path=index_path)
    feed.last_epoch = epochs - 1
    for e in range(epochs):
        slot = feed.get(e)
        assert slot == e % DEPTH
        feed.upload(slot)
        planes, xs = want[e]
        assert all(torch.equal(feed.noise[k], v) for k, v in planes.items()), e
        assert torch.equal(feed.x[feed.row_index.reshape(-1).long()], xs), e
    feed.close()
    assert same_states(gen_states(), end_ref)
    feed = EpochFeed(loader, B, S, L, nc, "cpu", index_path=index_path)           # abandoned after two epochs
    feed.last_epoch = 50
    feed.get(0); feed.get(1)
    feed.close()
    assert feed._producer is None


def test_other_samplers_take_the_generic_index_path():
    """Only DataLoader(shuffle=True, drop_last=True)'s own RandomSampler is reduced to one permutation slice; any other sampler is
    iterated through the loader's batch_sampler -- same contract."""
    from torch.utils.data import SubsetRandomSampler
    from hypad_amd.epoch_feed import _plain_random_batches
    B, S, L, nc = 16, 12, 5, 2
    ds = Windows(90, S)
    fast = DataLoader(ds, batch_size=B, drop_last=True, shuffle=True)
    assert _plain_random_batches(fast) is not None
    for loader in (DataLoader(ds, batch_size=B, drop_last=True, sampler=SubsetRandomSampler(range(10, 90))),
                   DataLoader(ds, batch_size=B, drop_last=True, shuffle=False),
                   DataLoader(ds, batch_size=B, drop_last=True, shuffle=True, generator=torch.Generator().manual_seed(5))):
        shared = loader.generator is not None
        assert (_plain_random_batches(loader) is not None) == shared
        if shared:
            loader.generator.manual_seed(5)
        np.random.seed(2); torch.manual_seed(2)
        want = [reference_epoch(loader, B, S, L, nc) for _ in range(2)]
        end_ref = gen_states()
        gstate = loader.generator.get_state().clone() if shared else None
        if shared:
            loader.generator.manual_seed(5)
        np.random.seed(2); torch.manual_seed(2)
        feed = EpochFeed(loader, B, S, L, nc, "cpu")
        assert feed.index_path
        feed.last_epoch = 1
        for e in range(2):
            feed.upload(feed.get(e))
            assert all(torch.equal(feed.noise[k], v) for k, v in want[e][0].items())
            assert torch.equal(feed.x[feed.row_index.reshape(-1).long()], want[e][1])
        feed.close()
        assert same_states(gen_states(), end_ref)
        if shared:
            assert torch.equal(loader.generator.get_state(), gstate)           # the loader's own generator too
