"""Host side of ``train.train_tadgan`` (train.py:252-385) when an epoch runs as ONE captured ``hypad_train_epoch``.

The reference's epoch is 5 passes of (critic_x_iteration, critic_z_iteration) and one pass of decoder_iteration over a
shuffling DataLoader; every iteration draws its latent vectors from NumPy's global generator and its interpolation
weights from torch's CPU generator, and every pass draws the loader's seeds from torch's generator when its
iterator is created.  None of these draws depends on a result of the training: an epoch's worth of them can be made
before the epoch is launched, *from the same generators in the same order*, written into pinned planes and uploaded
with one copy (``hypad_epoch_noise``).  ``EpochFeed.prepare`` does exactly that for epoch e while the GPU runs e - 1:

* NumPy stream (train.py:24,118,205): z_cx(it), z_cz(it) per critic iteration, then z_gen per generator batch --
  ``host_rng.global_normal_into`` on a helper thread (no interpreter lock), up to one epoch ahead.
* torch stream: per pass the loader's iterator is created / advanced to its first batch (``_BaseDataLoaderIter`` draws its
  base seed, ``RandomSampler`` its own), THEN the pass's alphas are drawn in one call -- alpha_cx(it), alpha_cz(it)
  interleaved as train.py:64,149 draws them (a single ``torch.rand`` of the concatenated length yields the same values as
  the successive calls: tests/test_epoch_feed.py).
* samples: either the loader's batches staged through pinned memory (any iterable with ``len``), or -- a plain
  ``torch.utils.data.DataLoader`` over one of ``hypad_amd.utils``' datasets or a tensor, default collate -- only the
  batch *indices*: the loader's own ``batch_sampler`` is iterated after drawing the base seed the way the iterator would,
  the window matrix stays resident in HBM and no sample is fetched, collated or copied (and no worker process started).

Assumption, stated: fetching a batch does not itself draw from the process-wide NumPy / torch generators (true for
torch's samplers, which seed a private generator, and for the reference's datasets).  A loader whose dataset does -- random
augmentation inside ``__getitem__`` -- needs ``params.per_iteration = True`` (the call-by-call loop).
"""
import queue
import threading

import numpy as np
import torch

from . import _C, host_rng

def _staging(*shape, dtype=torch.float32):
    """Host staging buffer: pinned where a GPU exists (non-blocking uploads); plain memory in the CPU-only tests of the draw order."""
    t = torch.empty(*shape, dtype=dtype)
    return t.pin_memory() if torch.cuda.is_available() else t


DEPTH = 4          # staging slots: the producer thread fills slot (e + 2) % 4 while the caller handles epoch e; that slot's last user, e - 2,
                   # finished uploading before the caller took epoch e (see EpochFeed.get)


def _index_matrix(loader, test=False):
    """The (N, S) window matrix behind `loader` if its batches can be described by indices alone, else None.  ``test``: accept the
    test-mode datasets too (their items are (window, index, y, y_index, X_index) tuples, utils/dataloader.py:229-231; the caller
    only uses the window)."""
    from torch.utils.data import DataLoader, IterableDataset
    from torch.utils.data.dataloader import default_collate
    if type(loader) is not DataLoader or loader.batch_sampler is None or loader.collate_fn is not default_collate:
        return None
    ds = loader.dataset
    if isinstance(ds, IterableDataset):
        return None
    if isinstance(ds, torch.Tensor):
        m = ds
    elif isinstance(ds, np.ndarray):
        m = torch.from_numpy(ds)
    elif type(ds).__module__.startswith("hypad_amd.") and hasattr(ds, "X") and (test or not getattr(ds, "test", False)):
        m = torch.as_tensor(np.asarray(ds.X))       # SignalDataset / MultivariateDataset: __getitem__(i) is X[i]
    elif hasattr(ds, "X") and (test or not getattr(ds, "test", False)) and _items_are_rows_of_X(ds):
        # somebody else's dataset class built like the reference's (utils/dataloader.py:61-232: the windows in ``X``, ``__getitem__(i)``
        # = ``torch.from_numpy(X[i])`` [, the index arrays]) -- what a user who swaps only train.py / anomaly_detection.py passes in
        m = torch.as_tensor(np.asarray(ds.X))
    else:
        return None
    if m.dim() < 2 or len(m) != len(ds):
        return None
    return m.reshape(len(m), -1)


def _items_are_rows_of_X(ds, probes=10):
    """Does ``ds[i]`` (its first element, for the test-mode tuples) equal ``ds.X[i]`` bit for bit?  Asked of ``probes`` items spread over
    the dataset -- first, last and fixed pseudo-random ones -- with the global generators' states put back afterwards (a dataset that
    augments its items draws from them and fails the comparison: it keeps the fetch-and-collate path)."""
    import random
    try:
        X = np.asarray(ds.X)
        n = len(ds)
        if X.ndim < 2 or len(X) != n or n < 1 or X.dtype.kind != "f":
            return False
        states = (np.random.get_state(), torch.get_rng_state(), random.getstate())
        try:
            for k in range(probes):
                i = (0, n - 1)[k] if k < 2 else (k * 2654435761) % n
                item = ds[i]
                if isinstance(item, (tuple, list)):
                    item = item[0]
                a = item.numpy() if isinstance(item, torch.Tensor) and not item.is_cuda else np.asarray(item)
                if a.shape != X[i].shape or a.dtype != X.dtype or not np.array_equal(a, X[i]):
                    return False
        finally:
            np.random.set_state(states[0]); torch.set_rng_state(states[1]); random.setstate(states[2])
        return True
    except Exception:
        return False


def loader_batches(loader):
    """The index batches one pass over `loader` would fetch, with the draws its iterator would make from torch's generators and
    nothing else: _BaseDataLoaderIter.__init__ (torch/utils/data/dataloader.py) takes one int64 from loader.generator (None = the
    default generator) as the workers' base seed -- with shuffle=False too --, then the batch sampler runs (RandomSampler draws its
    own seed at the first next()).  A generator: the caller may interleave its own draws after the first batch, as the reference's
    loop body does.  Pinned against the real iterator by tests/test_epoch_feed.py."""
    torch.empty((), dtype=torch.int64).random_(generator=loader.generator)
    yield from loader.batch_sampler


def _plain_random_batches(loader):
    """(sampler, n) when `loader`'s batches are the consecutive batch_size-runs of ONE permutation drawn by torch's own RandomSampler
    (DataLoader(shuffle=True, drop_last=True) builds exactly that), else None: the pass's indices are then one tensor slice instead of
    1 916 Python-level yields."""
    from torch.utils.data import BatchSampler, RandomSampler
    bs = loader.batch_sampler
    if type(bs) is not BatchSampler or type(bs.sampler) is not RandomSampler or not bs.drop_last:
        return None
    sm = bs.sampler
    if sm.replacement or sm._num_samples is not None:
        return None
    return sm, len(sm.data_source)


class EpochFeed:
    def __init__(self, train_loader, batch, signal_shape, latent_dim, n_critics, device, index_path=True):
        self.loader, self.B, self.S, self.L, self.nc, self.device = train_loader, int(batch), int(signal_shape), int(latent_dim), int(n_critics), device
        try:
            self.nb = len(train_loader)
        except TypeError as e:
            raise _C.HypadError("train_tadgan's epoch form needs len(train_loader); pass params.per_iteration = True for other iterables") from e
        if self.nb < 1:
            raise _C.HypadError("the loader yields no minibatch")
        B, S, L, nb, nc = self.B, self.S, self.L, self.nb, self.nc
        nit = nb * nc
        self.nit = nit
        # one contiguous block per slot: [z_cx | z_cz | z_gen | alpha_cx | alpha_cz]
        sizes = [("z_cx", nit * B * L), ("z_cz", nit * B * L), ("z_gen", nb * B * L), ("alpha_cx", nit * B * S), ("alpha_cz", nit * B * L)]
        self.offsets, off = {}, 0
        for k, n in sizes:
            self.offsets[k] = (off, n)
            off += n
        self.plane_floats = off
        self.host = [_staging(off) for _ in range(DEPTH)]
        self.host_np = [h.numpy() for h in self.host]
        # TWO device sets (epoch parity): epoch e + 1 is uploaded and launched before the host has looked at epoch e's status word, and a
        # failed epoch is repeated from ITS planes / batches (Engine.check_status)
        self.dev_sets = [torch.empty(off, dtype=torch.float32, device=device) for _ in range(2)]
        self.noise_sets = [{k: d[o:o + n] for k, (o, n) in self.offsets.items()} for d in self.dev_sets]      # Engine.train_epoch(noise=...)
        self.dev, self.noise = self.dev_sets[0], self.noise_sets[0]                                            # (the set last uploaded)
        self.alpha_tmp = torch.empty(nb * (B * S + B * L), dtype=torch.float32)
        self.alpha_np = self.alpha_tmp.numpy()
        rows = (nc + 1) * nb * B
        matrix = _index_matrix(train_loader) if index_path else None
        self.index_path = matrix is not None
        if self.index_path:
            if matrix.shape[1] != S:
                raise _C.HypadError(f"the dataset's windows hold {matrix.shape[1]} values, params.signal_shape is {S}")
            xm = matrix.to(torch.float32).contiguous().to(device)
            self.x_sets = [xm, xm]
            self.idx_host = [_staging(nc + 1, nb * B, dtype=torch.int32) for _ in range(DEPTH)]
            self.row_index_sets = [torch.empty(nc + 1, nb * B, dtype=torch.int32, device=device) for _ in range(2)]
        else:
            self.x_sets = [torch.empty(rows, S, dtype=torch.float32, device=device) for _ in range(2)]
            self.x_host = [_staging(rows, S) for _ in range(DEPTH)]
            self.x_host_np = [t.numpy().reshape(-1, B, S) for t in self.x_host]      # (batch-sized views: NumPy converts a host batch in ~3 us, a torch copy_ takes ~15)
            ri = torch.arange(rows, dtype=torch.int32, device=device).view(nc + 1, nb * B)
            self.row_index_sets = [ri, ri]
            self._on_device = None
        self.x, self.row_index = self.x_sets[0], self.row_index_sets[0]
        self._fast_sampler = _plain_random_batches(train_loader) if self.index_path else None
        self._z_thread = {}            # epoch -> helper thread drawing its latent planes
        self._producer = None          # background thread that stages epochs ahead of the caller (get)
        self._queue = None
        self._stop = False
        self._z_error = None
        self.last_epoch = None         # epochs beyond this one are never drawn (the generators end where the reference's would)

    # ---- NumPy stream ------------------------------------------------------------------------------------
    def _draw_z(self, slot):
        try:
            h = self.host_np[slot]
            view = lambda k: h[self.offsets[k][0]: self.offsets[k][0] + self.offsets[k][1]]
            # (configs[3]: 4.1 M values per epoch -- the transforms on three helper threads; the reference configuration's 371 200 stay on this one)
            host_rng.global_normal_into([view("z_cx"), view("z_cz")], self.B * self.L, self.nit, threads=3)      # train.py:24 then :118, per iteration
            host_rng.global_normal_into([view("z_gen")], self.B * self.L, self.nb)                    # train.py:205
        except BaseException as e:             # surfaced by the main thread's join
            self._z_error = e

    def _start_z(self, epoch):
        if epoch in self._z_thread or (self.last_epoch is not None and epoch > self.last_epoch):
            return
        # ONE long-lived helper thread takes the epochs' draws in order (one stream: epoch e's follow epoch e - 1's).  A thread per
        # epoch was measurably slower on the GPU box's two-socket host: each new thread started on whatever core was free, half the
        # time across the socket link from the generator's buffers and the pinned planes (configs[3]: 13 ms per epoch against 7).
        ex = self.__dict__.get("_z_exec")
        if ex is None:
            from concurrent.futures import ThreadPoolExecutor
            ex = self._z_exec = ThreadPoolExecutor(1, thread_name_prefix="hypad-z-draws")
        fut = ex.submit(self._draw_z, epoch % DEPTH)
        fut.join = fut.result                  # (the callers' vocabulary)
        self._z_thread[epoch] = fut

    # ---- torch stream + samples --------------------------------------------------------------------------
    def _alphas(self, slot, p):
        """The pass's interpolation weights, drawn where the reference draws them relative to the loader's own draws: after
        the pass's iterator exists and has produced its first batch."""
        B, S, L, nb = self.B, self.S, self.L, self.nb
        host_rng.torch_rand_into(self.alpha_tmp)          # == torch.rand(shape, out=alpha_tmp), natively (no interpreter lock, ~3x torch's rate)
        # (NumPy copies: a strided torch copy of this size wakes the whole intra-op thread pool -- milliseconds on a 256-core host)
        t = self.alpha_np.reshape(nb, B * S + B * L)
        h = self.host_np[slot]
        ox, oz = self.offsets["alpha_cx"][0], self.offsets["alpha_cz"][0]
        h[ox + p * nb * B * S: ox + (p + 1) * nb * B * S].reshape(nb, B * S)[:] = t[:, :B * S]
        h[oz + p * nb * B * L: oz + (p + 1) * nb * B * L].reshape(nb, B * L)[:] = t[:, B * S:]

    def _pass_indices(self, slot, p):
        out = self.idx_host[slot][p]
        fast = self._fast_sampler
        if fast is not None and self.loader.batch_sampler.batch_size == self.B:
            # RandomSampler.__iter__ (torch/utils/data/sampler.py), its draws in its order: the iterator's base seed, the sampler's
            # seed (default generator, unless the sampler owns one), ONE permutation from the private generator -- and a second one
            # whose head of zero indices it discards (num_samples % n == 0), which matters when the generator is a shared one
            sm, n = fast
            torch.empty((), dtype=torch.int64).random_(generator=self.loader.generator)
            if sm.generator is None:
                g = torch.Generator()
                g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            else:
                g = sm.generator
            perm = torch.randperm(n, generator=g)
            torch.randperm(n, generator=g)
            if n < self.nb * self.B or n // self.B != self.nb:
                raise _C.HypadError(f"the loader yields {n // self.B} batches, len(train_loader) is {self.nb}")
            out.copy_(perm[: self.nb * self.B])
            if p < self.nc:
                self._alphas(slot, p)
            return
        it = loader_batches(self.loader)
        n = 0
        first = True
        for idx in it:
            if len(idx) != self.B:
                raise _C.HypadError(f"a minibatch of {len(idx)} windows: build the DataLoader with drop_last=True (main.py:38) -- the fused "
                                    "iterations are bound to one batch size")
            if n >= self.nb:
                raise _C.HypadError("the loader yielded more batches than len(train_loader)")
            out[n * self.B:(n + 1) * self.B] = torch.as_tensor(idx, dtype=torch.int32)
            n += 1
            if first and p < self.nc:
                self._alphas(slot, p)
                first = False
        if n != self.nb:
            raise _C.HypadError(f"the loader yielded {n} batches, len(train_loader) is {self.nb}")

    def _pass_samples(self, slot, p):
        B, S, nb = self.B, self.S, self.nb
        base = p * nb * B
        n = 0
        for sample in self.loader:
            if n >= nb:
                raise _C.HypadError("the loader yielded more batches than len(train_loader)")
            if n == 0 and p < self.nc:
                self._alphas(slot, p)
            try:
                rows = sample.reshape(B, S)
            except (RuntimeError, AttributeError) as e:
                raise _C.HypadError(f"a minibatch must hold batch_size x signal_shape = {B} x {S} values, got {tuple(getattr(sample, 'shape', ()))}: "
                                    "build the DataLoader with drop_last=True (main.py:38)") from e
            on_dev = rows.is_cuda
            if self._on_device is None:
                self._on_device = on_dev
            elif self._on_device != on_dev:
                raise _C.HypadError("the loader mixes host and device minibatches")
            # (float64 -> float32 here: .float() of the reference.  Device batches go straight into the epoch's device set -- slot and
            # epoch have the same parity -- on the caller's stream, in the caller's order.)
            if on_dev:
                self.x_sets[slot & 1][base + n * B: base + (n + 1) * B].copy_(rows)
            elif rows.dtype in (torch.float64, torch.float32) and not rows.requires_grad:
                np.copyto(self.x_host_np[slot][p * nb + n], rows.numpy(), casting="same_kind")
            else:
                self.x_host[slot][base + n * B: base + (n + 1) * B].copy_(rows)
            n += 1
        if n != nb:
            raise _C.HypadError(f"the loader yielded {n} batches, len(train_loader) is {nb}")

    def prepare(self, epoch, z_ahead=True):
        """Everything epoch `epoch` needs from the host, staged in slot epoch % DEPTH: the NumPy stream on a helper thread next to
        the torch stream and the loader passes.  ``z_ahead``: start the next epoch's latent draws at the end (callers that prepare
        epoch by epoch on their own thread; the producer thread is ahead of the caller anyway)."""
        slot = epoch % DEPTH
        self._start_z(epoch)
        for p in range(self.nc + 1):
            (self._pass_indices if self.index_path else self._pass_samples)(slot, p)
        self._z_thread.pop(epoch).join()
        if self._z_error is not None:
            raise self._z_error
        if z_ahead:
            self._start_z(epoch + 1)
        return slot

    # ---- epochs staged ahead of the caller ----------------------------------------------------------------------------------------
    def _produce(self, first):
        try:
            for e in range(first, self.last_epoch + 1):
                if self._stop:
                    return
                slot = self.prepare(e, z_ahead=False)
                self._queue.put((e, slot, None))                 # (blocks while the caller has not taken the previous epoch)
        except BaseException as exc:
            self._queue.put((-1, -1, exc))

    def get(self, epoch):
        """Slot holding epoch `epoch`, staged.  Epoch 0 is staged by the caller's thread; from then on a producer thread runs ahead:
        while the caller handles epoch e (waits for e - 1's losses, uploads and launches e) it stages e + 1 and then e + 2, never
        more (a one-element queue).  Staging e + 2 overwrites slot (e - 2) mod 4, whose upload is known complete: the caller waited for
        epoch e - 2's losses before it took epoch e.  Minibatches that arrive as DEVICE tensors are copied on the caller's stream in
        the caller's order, so such loaders are staged epoch by epoch on the caller's thread."""
        if self.last_epoch is None:
            raise _C.HypadError("EpochFeed.get needs last_epoch (the generators must end where the reference's loop leaves them)")
        if self._producer is None:
            slot = self.prepare(epoch, z_ahead=bool(not self.index_path and self._on_device))
            if (self.index_path or not self._on_device) and epoch < self.last_epoch:
                self._queue = queue.Queue(maxsize=1)
                self._producer = threading.Thread(target=self._produce, args=(epoch + 1,), name="hypad-epoch-feed", daemon=True)
                self._producer.start()
            return slot
        e, slot, exc = self._queue.get()
        if exc is not None:
            raise exc
        if e != epoch:
            raise _C.HypadError(f"epochs must be taken in order (asked for {epoch}, staged {e})")
        return slot

    def upload(self, slot):
        """Enqueue the slot's planes (one copy) and samples / indices on the current stream into the device set of the slot's parity
        (= the epoch's: DEPTH is even); stream order puts the copies behind the launches of the epoch before last, which read that set.
        Returns (x, row_index, noise) of the set -- also left in ``self.x / .row_index / .noise``."""
        k = slot & 1
        self.dev, self.noise, self.x, self.row_index = self.dev_sets[k], self.noise_sets[k], self.x_sets[k], self.row_index_sets[k]
        self.dev.copy_(self.host[slot], non_blocking=True)
        if self.index_path:
            self.row_index.copy_(self.idx_host[slot], non_blocking=True)
        elif not self._on_device:
            self.x.copy_(self.x_host[slot], non_blocking=True)
        return self.x, self.row_index, self.noise

    def close(self):
        self._stop = True
        if self._producer is not None:
            while self._producer.is_alive():                   # (unblock a producer waiting to hand over an epoch nobody will take)
                try:
                    self._queue.get(timeout=0.05)
                except queue.Empty:
                    pass
            self._producer.join()
            self._producer = None
        for t in list(self._z_thread.values()):
            t.join()
        self._z_thread.clear()
        ex = self.__dict__.pop("_z_exec", None)
        if ex is not None:
            ex.shutdown(wait=True)
