"""Training surface of the reference (train.py) over the fused HIP iterations.

``critic_x_iteration`` / ``critic_z_iteration`` / ``decoder_iteration`` keep the reference's positional
signatures and return types (train.py:18, :107, :189); each is forward + backward + optimizer step in three
(critics) or two (generator) kernel launches.  Host-side randomness follows the reference: the
latent draw comes from NumPy's global generator and the interpolation weights from torch's CPU generator, then
travel to the device; pass ``z=`` / ``alpha=`` to inject them, or use ``hypad_amd.engine.Engine`` for the
device-RNG, HBM-resident fast path.

The optimizer argument may be a ``torch.optim.Adam`` (as the reference builds, train.py:274-281), or
``hypad_amd.optim.Adam`` / ``RiemannianAdam``: only its hyper-parameters are read; the moments live in flat
arenas attached to it on first use and are mirrored into ``optimizer.state`` as views.
"""
import logging
import os
import time
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

from . import _C
from .engine import Engine
from .models import tadgan
from .optim import Adam, RiemannianAdam

_NET_OF = {tadgan.Encoder: "enc", tadgan.Decoder: "dec", tadgan.CriticX: "cx", tadgan.CriticZ: "cz"}


class _NoiseStage:
    """Host-drawn noise of one iteration -> the kernels, without a copy launch: a ring of PINNED host rows [z | alpha].  NumPy /
    torch CPU write their draws straight into the pinned row and the iteration's first kernel reads it over PCIe through the
    row's own address (pinned host memory is mapped into the device's address space: 27 KB, read once, ~3 us) -- the
    non-blocking H2D copy this replaces cost ~10 us of host time per iteration (a torch copy_ launch plus views), in a loop that
    is bound by the host.  A pinned row must not be refilled while a kernel may still read it: the ring is walked in groups of
    GROUP slots; on entering a group, ONE event is recorded for the group just left (it covers every kernel enqueued so far) and
    the event recorded for this group a lap ago is awaited.  HYPAD_DROPIN_ZEROCOPY=0: one H2D copy per iteration into a device
    row instead (the device row is protected by stream order)."""
    SLOTS, GROUP = 32, 8
    ZEROCOPY = os.environ.get("HYPAD_DROPIN_ZEROCOPY", "0") == "1"

    def __init__(self, device, floats):
        self.floats = floats
        self.host = torch.empty(self.SLOTS, floats, dtype=torch.float32).pin_memory()
        self.host_np = self.host.numpy()
        self.dev = None if self.ZEROCOPY else torch.empty(self.SLOTS, floats, dtype=torch.float32, device=device)
        self.events = [None] * (self.SLOTS // self.GROUP)
        self.k = self.SLOTS - 1

    def slot(self):
        k = self.k = (self.k + 1) % self.SLOTS
        if k % self.GROUP == 0:
            ng = self.SLOTS // self.GROUP
            g, left = k // self.GROUP, (k // self.GROUP - 1) % ng
            ev = self.events[left]
            if ev is None:
                ev = self.events[left] = torch.cuda.Event()
            ev.record()                                   # everything enqueued so far: the readers of the group just left among it
            if self.events[g] is not None:
                self.events[g].synchronize()              # the readers of this group's rows, a lap ago
        return k

    def upload(self, k):
        if self.ZEROCOPY:
            return self.host[k]
        self.dev[k].copy_(self.host[k], non_blocking=True)
        return self.dev[k]


class _Fused:
    """Flat Adam moments + device counters attached to one optimizer."""

    def __init__(self, optim, modules, dims, resume_epoch=None):
        _check_optimizer(optim, modules)
        g = optim.param_groups[0]
        self.modules = modules
        self.key = _fused_key(modules, dims)
        self.exp_avg = {k: torch.zeros_like(m.arena()) for k, m in modules.items()}
        self.exp_avg_sq = {k: torch.zeros_like(m.arena()) for k, m in modules.items()}
        dev = next(iter(modules.values())).arena().device
        S, L, B, hyp = dims
        salt = sum(ord(c) for k in modules for c in k) * 0x9E3779B97F4A7C15        # decorrelate the three optimizers' streams
        salt += _resume_salt(resume_epoch)
        self.engine = Engine(S, L, B, hyp, 1, dev, lr=g["lr"], betas=g.get("betas", (0.9, 0.999)), eps=g.get("eps", 1e-8),
                             gen_weight_decay=g.get("weight_decay", 0.0), gen_stabilize=g.get("stabilize") or 0,
                             seed=torch.initial_seed() + salt)
        self.steps = 0
        self.step_tensor = torch.tensor(0.0)   # ONE tensor shared by every parameter's state["step"]: one fill per step, not one per tensor
        for k, m in modules.items():          # expose the moments the way torch optimizers do
            for name, (p, off, n, shape) in m._slots.items():
                optim.state[p] = {"step": self.step_tensor, "exp_avg": self.exp_avg[k][off:off + n].view(shape),
                                  "exp_avg_sq": self.exp_avg_sq[k][off:off + n].view(shape)}
        self.optim = optim
        self.noise = None
        self._bound = None

    def bind(self, others):
        """The engine pointed at the current arenas of the stepped and the frozen networks.  Re-adopting (new views, dropped
        graphs) only when a module or one of its arenas changed since the last call."""
        mods = {**others, **self.modules}
        sig = tuple((k, id(m), m.arena(fast=True).data_ptr()) for k, m in mods.items())
        if sig != self._bound:
            arenas = {k: m.arena() for k, m in mods.items()}
            for k in ("enc", "dec", "cx", "cz"):   # the ABI wants four valid pointers; untouched nets borrow one
                arenas.setdefault(k, next(iter(arenas.values())))
            self.engine.adopt(arenas, self.exp_avg, self.exp_avg_sq)
            self._bound = sig
        lr = float(self.optim.param_groups[0]["lr"])
        if lr != self.engine.lr:
            self.engine.lr = lr
        return self.engine

    def stage(self, B, L, alpha_width, z, alpha, device):
        """(z, alpha) on the device for one iteration, drawn where the reference draws them (z: NumPy's global generator,
        train.py:24,118,205; alpha: torch's CPU generator, train.py:64,149 -- in that order) unless given."""
        nz, na = B * L, B * alpha_width
        if self.noise is None or self.noise.floats != nz + na:
            self.noise = _NoiseStage(device, nz + na)
        st = self.noise
        k = st.slot()
        if z is None:
            z = np.random.normal(size=(1, B, L))
        st.host_np[k, :nz] = np.asarray(z, dtype=np.float64).reshape(-1)          # (cast to float32 on assignment)
        if na:
            row = st.host[k, nz:]
            if alpha is None:
                torch.rand((1, B, alpha_width), out=row.view(1, B, alpha_width))
            else:
                row.copy_(torch.as_tensor(alpha, dtype=torch.float32).reshape(-1))
        d = st.upload(k)
        return d[:nz].view(1, B, L), (d[nz:].view(1, B, alpha_width) if na else None)

    def stepped(self):
        self.steps += 1
        self.step_tensor.fill_(self.steps)


def _resume_salt(resume_epoch):
    """A resumed run (params.resume) builds fresh engines whose device counters -- the Philox tick among them -- start at
    zero: without this the resumed epochs would replay the latent draws, interpolation weights, dropout masks and shuffles
    of the original run's first epochs.  The resume epoch is folded into the seed instead."""
    return 0 if resume_epoch is None else (int(resume_epoch) + 1) * 0xD1B54A32D192ED03


def _check_optimizer(optim, modules):
    """Only lr, betas, eps (and, for the generator's RiemannianAdam, weight_decay / stabilize) are honoured by the fused step:
    refuse anything else loudly instead of ignoring it."""
    if len(optim.param_groups) != 1:
        raise _C.HypadError("the fused iterations support exactly one param group per optimizer (train.py:274-288 builds one)")
    g = optim.param_groups[0]
    for flag in ("amsgrad", "maximize", "capturable", "differentiable", "fused"):
        if g.get(flag):
            raise _C.HypadError(f"optimizer option {flag}=True is not supported by the fused iterations")
    if g.get("weight_decay", 0.0) and not ("dec" in modules and (getattr(optim, "riemannian", False) or "stabilize" in g)):
        raise _C.HypadError("weight_decay is only supported on the hyperbolic generator's RiemannianAdam (train.py:282-288); "
                            "the critics' torch.optim.Adam steps have none (train.py:274-281)")
    owned = {id(p) for m in modules.values() for p in m.parameters()}
    if {id(p) for p in g["params"]} != owned:
        raise _C.HypadError("the optimizer must hold exactly the parameters of the module(s) this iteration updates")


def _fused_key(modules, dims):
    return (tuple(sorted((k, id(m)) for k, m in modules.items())), tuple(dims))


def _fused(optim, modules, params, hyperbolic):
    """The flat moments + engine attached to `optim`.  Bound to the modules and dimensions of its first use: a later call with
    other modules, window length, latent width, batch size or geometry would silently restart Adam from zero moments, so it
    raises instead (build a new optimizer, as the reference would)."""
    f = getattr(optim, "_hypad", None)
    dims = (params.signal_shape, params.latent_space_dim, params.batch_size, bool(hyperbolic))
    if f is None:
        f = _Fused(optim, modules, dims, params.resume_epoch if getattr(params, "resume", False) else None)
        optim._hypad = f
    elif f.key != _fused_key(modules, dims):
        raise _C.HypadError("this optimizer is already bound to other modules / dimensions (signal_shape, latent_space_dim, "
                            "batch_size, hyperbolic): its Adam moments cannot be carried over -- create a new optimizer.  (A last, "
                            "smaller minibatch does this too: build the DataLoader with drop_last=True, as main.py:38 does.)")
    return f


def _sample(sample, params):
    try:
        x = sample.reshape(1, params.batch_size, params.signal_shape)
    except RuntimeError as e:
        raise _C.HypadError(f"a minibatch must hold batch_size x signal_shape = {params.batch_size} x {params.signal_shape} values, got "
                            f"{tuple(sample.shape)}: build the DataLoader with drop_last=True (main.py:38) -- the fused iterations are "
                            "bound to one batch size") from e
    if not x.is_cuda:
        x = _SampleStage.get(x).upload(x)
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    return x if x.is_contiguous() else x.contiguous()


class _SampleStage:
    """A minibatch that arrives in host memory (the reference's DataLoader yields CPU tensors): `sample.cuda()` from pageable memory
    is a blocking copy -- the host stops until every kernel enqueued before it has run, so host and GPU time of an iteration add
    up.  Instead the sample is copied into a ring of pinned rows (a 51 KB memcpy) and goes to the device with a non-blocking
    copy; one event per GROUP rows guards the ring (recorded on entering the next group, awaited a lap later)."""
    SLOTS, GROUP = 16, 4
    _rings = {}

    @classmethod
    def get(cls, x):
        key = (tuple(x.shape), x.dtype)
        ring = cls._rings.get(key)
        if ring is None:
            ring = cls._rings[key] = cls(x)
        return ring

    def __init__(self, x):
        self.host = torch.empty((self.SLOTS,) + tuple(x.shape), dtype=x.dtype).pin_memory()
        self.dev = torch.empty((self.SLOTS,) + tuple(x.shape), dtype=x.dtype, device="cuda")
        self.events = [None] * (self.SLOTS // self.GROUP)
        self.k = self.SLOTS - 1

    def upload(self, x):
        k = self.k = (self.k + 1) % self.SLOTS
        if k % self.GROUP == 0:
            ng = self.SLOTS // self.GROUP
            g, left = k // self.GROUP, (k // self.GROUP - 1) % ng
            ev = self.events[left]
            if ev is None:
                ev = self.events[left] = torch.cuda.Event()
            ev.record()
            if self.events[g] is not None:
                self.events[g].synchronize()
        self.host[k].copy_(x)
        self.dev[k].copy_(self.host[k], non_blocking=True)
        return self.dev[k]


def _train_flag(*mods):
    flags = {bool(m.training) for m in mods}
    if len(flags) != 1:
        raise _C.HypadError("mixed train()/eval() modes across the networks of one iteration are not supported")
    return flags.pop()


def critic_x_iteration(sample, decoder, critic_x, optim_cx, params, z=None, alpha=None, dropout_masks=None):
    """train.py:18-104.  Returns the 0-d loss (float64, as the reference's loss is: SURVEY.md §7 hard part 2)."""
    x = _sample(sample, params)
    f = _fused(optim_cx, {"cx": critic_x}, params, decoder.hyperbolic)
    eng = f.bind({"dec": decoder})
    zd, ad = f.stage(params.batch_size, params.latent_space_dim, params.signal_shape, z, alpha, x.device)
    losses = eng.critic_x_iteration(x, None, zd, ad, _train_flag(decoder, critic_x), dropout_masks)
    f.stepped()
    return losses.view(-1)[0].to(torch.float64)


def critic_z_iteration(sample, encoder, critic_z, optim_cz, params, z=None, alpha=None, dropout_masks=None):
    """train.py:107-186."""
    x = _sample(sample, params)
    f = _fused(optim_cz, {"cz": critic_z}, params, False)
    eng = f.bind({"enc": encoder})
    zd, ad = f.stage(params.batch_size, params.latent_space_dim, params.latent_space_dim, z, alpha, x.device)
    losses = eng.critic_z_iteration(x, None, zd, ad, _train_flag(encoder, critic_z), dropout_masks)
    f.stepped()
    return losses.view(-1)[0]


def decoder_iteration(sample, encoder, decoder, critic_x, critic_z, optim_dec, params, err_loss=None, z=None,
                      dropout_masks=None):
    """train.py:189-249.  Returns (loss_dec, hyper_loss, mse) with the reference's conventions:
    hyperbolic -> (loss, hyper_loss, torch.Tensor([0])); Euclidean -> (loss, 0, mse_loss)."""
    if err_loss is not None and not isinstance(err_loss, nn.MSELoss):
        raise NotImplementedError("decoder_iteration: only nn.MSELoss() (the reference's default) is fused")
    x = _sample(sample, params)
    f = _fused(optim_dec, {"dec": decoder, "enc": encoder}, params, decoder.hyperbolic)
    if decoder.hyperbolic and not getattr(optim_dec, "riemannian", False) and "stabilize" not in optim_dec.param_groups[0]:
        raise _C.HypadError("hyperbolic decoder_iteration needs a RiemannianAdam optimizer (train.py:282-288)")
    eng = f.bind({"cx": critic_x, "cz": critic_z})
    zd, _ = f.stage(params.batch_size, params.latent_space_dim, 0, z, None, x.device)
    losses = eng.decoder_iteration(x, None, zd, _train_flag(encoder, decoder, critic_x, critic_z), dropout_masks)
    f.stepped()
    flat = losses.view(-1)                       # (a fresh (1, 4) tensor per call: views of it are the caller's to keep)
    if decoder.hyperbolic:
        return flat[0], flat[1], torch.Tensor([0])
    return flat[0], 0, flat[1]


def encoder_iteration(sample, encoder, decoder, critic_x, critic_z, optim_enc, params, err_loss=None, z=None, dropout_masks=None):
    """The name BASELINE.json's north_star uses for the generator step.  The reference has no live function of this name
    (its call is commented out at train.py:348): ``decoder_iteration`` updates the encoder *and* the decoder
    with one optimizer, and this is that function under the other name."""
    return decoder_iteration(sample, encoder, decoder, critic_x, critic_z, optim_enc, params, err_loss, z, dropout_masks)


def _set_requires_grad(modules, flag):
    for m in modules:
        for p in m.parameters():
            p.requires_grad = flag


def make_optimizers(encoder, decoder, critic_x, critic_z, params):
    """train.py:274-288."""
    optim_cx = Adam(critic_x.parameters(), lr=params.lr, betas=(0.9, 0.999))
    optim_cz = Adam(critic_z.parameters(), lr=params.lr, betas=(0.9, 0.999))
    gen = list(decoder.parameters()) + list(encoder.parameters())
    if params.hyperbolic:
        optim_dec = RiemannianAdam(gen, lr=params.lr, weight_decay=1e-5, stabilize=10)
    else:
        optim_dec = Adam(gen, lr=params.lr, betas=(0.9, 0.999))
    return optim_cx, optim_cz, optim_dec


class _SavedLayout:
    """``torch.save(module, f)`` for a module whose parameters are views of ONE flat storage (hypad_amd's arena modules), repeated: the
    archive ``torch.save`` writes is a plain stored zip -- the pickled object graph (``data.pkl``: identical from save to save as long as
    the module's structure is), a few constant members, and the storage's raw bytes as ``data/0`` -- so every save after the first
    re-writes that zip with only the storage record replaced.  Pickling the object graph is ~0.5 ms of interpreter time per module
    (a Python call-back per pickled object); 128 files per checkpoint of 32 models made the checkpoint worker, not the GPU, the bound of
    ``train_signals_resident`` (9 ms per epoch against 5.8).  ``parse`` returns None for anything it does not recognise exactly (no or
    several storage records with the arena's bytes, a compressed member): the caller then keeps calling ``torch.save``."""

    def __init__(self, members, record, nbytes):
        self.members, self.record, self.nbytes = members, record, nbytes

    @classmethod
    def parse(cls, raw, storage_bytes):
        import io, re, zipfile
        try:
            with zipfile.ZipFile(io.BytesIO(raw)) as z:
                infos = z.infolist()
                if any(i.compress_type != zipfile.ZIP_STORED for i in infos):
                    return None
                records = [i for i in infos if re.fullmatch(r".*/data/\d+", i.filename)]
                # (other storage records -- the ball's curvature and the like, a few bytes each, not trained: nothing outside the arena is -- are kept as they are)
                mine = [i for i in records if i.file_size == len(storage_bytes) and z.read(i) == storage_bytes]
                if len(mine) != 1:
                    return None
                members = [(i.filename, None if i is mine[0] else z.read(i)) for i in infos]
            return cls(members, mine[0].filename, len(storage_bytes))
        except Exception:
            return None

    def write(self, f, storage_bytes):
        import zipfile
        if len(storage_bytes) != self.nbytes:
            raise _C.HypadError("checkpoint layout: the storage changed size")
        import struct
        with zipfile.ZipFile(f, "w", compression=zipfile.ZIP_STORED) as z:
            for name, data in self.members:
                info = zipfile.ZipInfo(name)
                # every record's bytes start on a 64-byte boundary, as in torch.save's own archives (PyTorchStreamWriter pads the local
                # header with an "FB" extra field): what lets torch.load(mmap=True) and other readers map the storage in place
                start = z.fp.tell() + 30 + len(name.encode()) + 4
                pad = -start % 64
                info.extra = b"FB" + struct.pack("<H", pad) + b"Z" * pad
                z.writestr(info, storage_bytes if data is None else data)


class _CheckpointWriter:
    """The checkpoint files of train.py:381-385 without stopping the epoch pipeline for them: the training thread copies the
    parameter arenas on the device (``snapshot``: stream-ordered behind the epoch just queued, in front of the next one -- 1 MB per
    model, microseconds), a worker thread puts a copy into ITS OWN module objects (``templates``: deep copies of the live modules,
    taken at the first checkpoint -- or the modules the caller hands over) and pickles those with ``torch.save``: the
    device-to-host copies and the file writes, 2-4 ms per model, which every tenth epoch used to wait for with an empty queue
    behind it.  Same files; all of them complete when ``close`` returns."""

    def __init__(self, device, modules=None, max_jobs=0):
        self.modules, self.device = modules, device            # modules: {key: live module} to deep-copy once, or None
        self.templates = None
        self.max_jobs = max_jobs
        self.layouts = {}                                      # id(template module) -> _SavedLayout, or False (keep calling torch.save)
        self.jobs = None
        self.thread = None
        self.error = None

    def snapshot(self, tensors=None):
        """{key: device copy} of the live modules' arenas (or of ``tensors``) + the event that follows the copies."""
        src = tensors if tensors is not None else {k: m.arena().detach() for k, m in self.modules.items()}
        snap = {k: t.clone() for k, t in src.items()}
        ev = torch.cuda.Event()
        ev.record()
        return snap, ev

    def submit(self, snapshot, files, pick=None):
        """files: {key: path} -- or, with ``pick``, a list of (template module, key, index into the snapshot's first dimension, path).
        Returns at once (unless ``max_jobs`` snapshots are already waiting); ``close`` waits for every file."""
        import queue, threading
        if self.error is not None:                   # an earlier file could not be written: fail at this checkpoint, not at the end of the run
            e, self.error = self.error, None
            raise e
        if self.thread is None:
            self.jobs = queue.Queue(maxsize=self.max_jobs)
            self.thread = threading.Thread(target=self._run, name="hypad-checkpoints", daemon=True)
            self.thread.start()
        self.jobs.put((snapshot, files, pick))

    def _run(self):
        torch.cuda.set_device(self.device)
        # a stream of its own: the device-to-host copies inside torch.save wait for THEIR stream -- on the training stream that wait
        # would stand behind the epochs queued there and keep the training thread from queueing the next one meanwhile
        side = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(side):
            self._serve()

    def _serve(self):
        import copy, io
        while True:
            job = self.jobs.get()
            if job is None:
                return
            if self.error is not None:
                continue
            try:
                (snap, ev), files, pick = job
                torch.cuda.current_stream().wait_event(ev)
                if pick is None:
                    if self.templates is None:               # the writer's OWN deep copies: the flags the reference's loop has set when it saves
                        self.templates = {k: copy.deepcopy(m) for k, m in self.modules.items()}      # (train.py:333-340, 381-385: generator trainable,
                        for k, t in self.templates.items():                                          # critics frozen) are pickled with the module
                            _set_requires_grad((t,), k in ("enc", "dec"))
                    pick = [(self.templates[k], k, None, f) for k, f in files.items()]
                # (modules handed in through `pick` belong to the caller -- train_signals_resident sets those flags when it builds them, on its
                # own thread: this thread never changes an object somebody else may hold)
                host = {}                                    # the snapshot on the host: ONE copy per arena tensor, not one per file -- a small
                for t, k, i, f in pick:                      # copy queues behind the kernels that hold the chip (128 of them: 40-130 ms)
                    src = snap[k] if i is None else snap[k][i]
                    layout = self.layouts.get(id(t))
                    if layout:                               # every save after the module's first: its zip with the storage record replaced
                        if k not in host:
                            host[k] = snap[k].cpu().numpy()
                        layout.write(f, (host[k] if i is None else host[k][i]).reshape(-1).tobytes())
                        continue
                    if not next(t.parameters()).is_cuda:
                        t.to(self.device)
                    t.arena().data.copy_(src)
                    if layout is None and isinstance(f, (str, os.PathLike)):
                        buf = io.BytesIO()
                        torch.save(t, buf)
                        raw = buf.getvalue()
                        with open(f, "wb") as fh:
                            fh.write(raw)
                        self.layouts[id(t)] = _SavedLayout.parse(raw, t.arena().detach().reshape(-1).cpu().numpy().tobytes()) or False
                    else:
                        torch.save(t, f)
            except BaseException as e:                       # (re-raised by close on the caller's thread)
                self.error = e

    def close(self):
        if self.thread is not None:
            self.jobs.put(None)
            self.thread.join()
            self.thread = None
        if self.error is not None:
            e, self.error = self.error, None
            raise e


def _per_iteration_wanted(train_loader, params):
    if getattr(params, "per_iteration", False) or os.environ.get("HYPAD_TRAIN_PER_ITERATION") == "1":
        return True
    return not hasattr(train_loader, "__len__")


def train_tadgan(train_loader, encoder, decoder, critic_x, critic_z, n_epochs=2000, params=[], path=""):
    """train.py:252-385 at epoch speed: per epoch 5 passes of (critic_x_iteration, critic_z_iteration) over the loader, then one pass
    of decoder_iteration -- the reference's schedule, its host random numbers (latent vectors from NumPy's global generator, the
    interpolation weights and the loader's seeds from torch's CPU generator, each stream in the reference's order: epoch_feed.py),
    its prints and its checkpoint cadence -- but the 319 iterations of an epoch run as ONE captured ``hypad_train_epoch``: the
    epoch's random planes and minibatches (or just their window indices) are staged on the host while the previous epoch runs,
    uploaded with one copy and consumed through ``hypad_epoch_noise``; the per-iteration losses come back in one buffer.
    Dropout masks (the reference draws them on the GPU, not reproducibly) come from the device Philox generator.
    ``params.per_iteration = True`` selects the call-by-call loop instead."""
    if _per_iteration_wanted(train_loader, params):
        return train_tadgan_per_iteration(train_loader, encoder, decoder, critic_x, critic_z, n_epochs, params, path)
    from .epoch_feed import EpochFeed
    logging.debug("Starting training")
    B, S, L = params.batch_size, params.signal_shape, params.latent_space_dim
    hyp = bool(decoder.hyperbolic)
    train_mode = _train_flag(encoder, decoder, critic_x, critic_z)
    mods = {"enc": encoder, "dec": decoder, "cx": critic_x, "cz": critic_z}
    dev = encoder.arena().device
    resume = getattr(params, "resume", False)
    eng = Engine(S, L, B, hyp, 1, dev, lr=params.lr, betas=(0.9, 0.999), gen_weight_decay=1e-5 if hyp else 0.0,          # train.py:274-288
                 gen_stabilize=10 if hyp else 0, seed=torch.initial_seed() + _resume_salt(params.resume_epoch if resume else None))
    eng.adopt({k: m.arena() for k, m in mods.items()})
    eng.epoch_flags = int(getattr(params, "epoch_flags", 0))      # hypad_epoch_io.flags (A/B forms of the critic phase; tests)
    n_critics = 5
    feed = EpochFeed(train_loader, B, S, L, n_critics, dev, index_path=not getattr(params, "stage_samples", False))
    nb = feed.nb
    iters = (2 * n_critics + 1) * nb
    losses_dev = [torch.empty(1, iters, 4, dtype=torch.float32, device=dev) for _ in range(2)]       # per epoch parity, like the feed's device sets
    back = [torch.empty(iters * 4 + 8, dtype=torch.float32).pin_memory() for _ in range(2)]        # losses | counters (as bits)
    done = [torch.cuda.Event() for _ in range(2)]
    hist = SimpleNamespace(cx=[], cz=[], dec=[], hyper=[], mse=[], wall=[], repairs=0)      # wall: time.perf_counter() when each epoch's losses were on the
    # host; repairs: resident critic launches that gave up (bounded wait) and were repeated launch by launch -- a ~10x slower epoch each
    actual_epoch = 0
    if resume:
        n_epochs = n_epochs - params.resume_epoch
        actual_epoch = params.resume_epoch + 1
    feed.last_epoch = n_epochs - 1
    state = {"repaired_until": -1}
    saves = lambda e: ((actual_epoch + e + 1) % 10 == 0) or ((actual_epoch + e + 1) == (n_epochs - 1))      # train.py:381 (cadence kept as is)

    from . import streams as _streams
    copy_stream = _streams.beside([torch.cuda.current_stream()], dev)      # (a stream that shares the training stream's hardware queue would copy BEHIND the epoch, not under it)
    uploaded = [torch.cuda.Event() for _ in range(2)]
    writer = _CheckpointWriter(dev, mods)
    snaps = {}

    def enqueue(e, slot):
        # the epoch's planes and batches go up on a copy stream, under the previous epoch's kernels: its device set was last read by
        # epoch e - 2, whose completion event the copy waits for; the replay waits for the copy
        main = torch.cuda.current_stream()
        with torch.cuda.stream(copy_stream):
            if e >= 2:
                copy_stream.wait_event(done[e % 2])
            else:
                copy_stream.wait_stream(main)
            x, row_index, noise = feed.upload(slot)
            uploaded[e % 2].record()
        main.wait_event(uploaded[e % 2])
        eng.train_epoch_graph(x, row_index, nb, n_critics, train_mode, losses=losses_dev[e % 2], noise=noise)
        b = back[e % 2]
        b[: iters * 4].copy_(losses_dev[e % 2].view(-1), non_blocking=True)
        b[iters * 4:].view(torch.int32).copy_(eng.counters, non_blocking=True)
        done[e % 2].record()
        if saves(e):
            snaps[e] = writer.snapshot()                         # epoch e's weights, before epoch e + 1 is queued

    def finish(e):
        """Epoch e's losses on the host: the reference's end-of-epoch bookkeeping (train.py:329-385).  Epoch e + 1 is already queued
        behind it (the GPU never waits for this bookkeeping); a checkpoint's files hold exactly epoch e's weights all the same: they
        are written from the device copy enqueue took between the two epochs (_CheckpointWriter).  A resident critic launch that
        gave up stops everything behind it: check_status repeats epoch e AND the epoch queued behind it, each from its own planes,
        batches and loss buffer -- a checkpoint epoch among them is copied again, behind its repeat."""
        done[e % 2].synchronize()
        b = back[e % 2]
        if e > state["repaired_until"]:
            if int(b[iters * 4:].view(torch.int32)[4]) != 0:     # status word of the resident critic launch (hypad_epoch_status)
                def recopy(i):
                    if e + i in snaps:
                        snaps[e + i] = writer.snapshot()
                eng.check_status(on_epoch=recopy)                # restores, repeats every queued epoch with per-iteration launches
                state["repaired_until"] = e + 1
                hist.repairs += 1
            else:
                eng.confirm_epochs(1)                            # epoch e completed: one epoch less for a later repair to look at
        if e <= state["repaired_until"]:
            b[: iters * 4].copy_(losses_dev[e % 2].view(-1))     # (its pinned copy was taken from the stopped run)
        rows = b[: iters * 4].view(iters, 4)
        crit = rows[: 2 * n_critics * nb, 0].view(n_critics, nb, 2).mean(1)          # per pass (train.py:329-330), then over the passes
        gen = rows[2 * n_critics * nb:].mean(0)
        hist.cx.append(float(crit[:, 0].mean())); hist.cz.append(float(crit[:, 1].mean())); hist.dec.append(float(gen[0]))
        if hyp:
            hist.hyper.append(float(gen[1])); hist.mse.append(0.0)
        else:
            hist.mse.append(float(gen[1]))
        print("Encoder decoder training done in epoch {}".format(e))
        if hyp:
            print("Hyperbolic loss {}".format(hist.hyper[-1]))
        else:
            print("Eucl mse loss {}".format(hist.mse[-1]))
        print("critic x loss {:.3f} critic z loss {:.3f} \ndecoder loss {:.3f}\n".format(hist.cx[-1], hist.cz[-1], hist.dec[-1]))
        hist.wall.append(time.perf_counter())
        if saves(e):
            ae = actual_epoch + e + 1
            writer.submit(snaps.pop(e), {"enc": path + "/encoder_{}.pt".format(ae), "dec": path + "/decoder_{}.pt".format(ae),
                                         "cx": path + "/critic_x_{}.pt".format(ae), "cz": path + "/critic_z_{}.pt".format(ae)})

    try:
        for epoch in range(n_epochs):
            logging.debug("Epoch {}".format(epoch))
            slot = feed.get(epoch)                  # epoch e staged on the host (a producer thread runs up to two epochs ahead)
            enqueue(epoch, slot)
            if epoch > 0:
                finish(epoch - 1)
        if n_epochs > 0:
            finish(n_epochs - 1)
    finally:
        feed.close()
        writer.close()                              # every checkpoint file is complete when the call returns
    _set_requires_grad((decoder, encoder), True)     # the flags the reference's loop leaves behind (train.py:331-332)
    _set_requires_grad((critic_x, critic_z), False)
    return hist


def train_tadgan_per_iteration(train_loader, encoder, decoder, critic_x, critic_z, n_epochs=2000, params=[], path=""):
    """train.py:252-385, call by call: per epoch 5 passes of (critic_x_iteration, critic_z_iteration) over the loader, then one pass of
    decoder_iteration -- three (critics) / two (generator) launches and one H2D copy per call, the host in between.  Kept for loaders
    the epoch form cannot stage (no ``len``; datasets that draw from the global generators while a batch is fetched) and as the
    reference point of its parity test; ``train_tadgan`` lands here with ``params.per_iteration = True``."""
    logging.debug("Starting training")
    optim_cx, optim_cz, optim_dec = make_optimizers(encoder, decoder, critic_x, critic_z, params)
    cx_epoch_loss, cz_epoch_loss, decoder_epoch_loss, hyp_dec_loss, eucl_dec_loss, wall = [], [], [], [], [], []
    actual_epoch = 0
    if params.resume:
        n_epochs = n_epochs - params.resume_epoch
        actual_epoch = params.resume_epoch + 1
    for epoch in range(n_epochs):
        logging.debug("Epoch {}".format(epoch))
        n_critics = 5
        cx_nc_loss, cz_nc_loss = [], []
        _set_requires_grad((decoder, encoder), False)
        _set_requires_grad((critic_x, critic_z), True)
        for _ in range(n_critics):
            cx_loss, cz_loss = [], []
            for sample in train_loader:
                sample = sample.cuda()
                cx_loss.append(critic_x_iteration(sample, decoder, critic_x, optim_cx, params))
                cz_loss.append(critic_z_iteration(sample, encoder, critic_z, optim_cz, params))
            cx_nc_loss.append(torch.mean(torch.stack(cx_loss).float()))      # one device->host sync per pass
            cz_nc_loss.append(torch.mean(torch.stack(cz_loss).float()))
        _set_requires_grad((decoder, encoder), True)
        _set_requires_grad((critic_x, critic_z), False)
        logging.debug("Critic training done in epoch {}".format(epoch))
        decoder_loss, hyp_loss, mse_losss = [], [], []
        for sample in train_loader:
            dec_loss, hyper_loss, mse_loss = decoder_iteration(sample.cuda(), encoder, decoder, critic_x, critic_z, optim_dec, params)
            decoder_loss.append(dec_loss)
            if params.hyperbolic:
                hyp_loss.append(hyper_loss.float())
            mse_losss.append(mse_loss.float().reshape(()).to(dec_loss.device))
        cx_epoch_loss.append(torch.mean(torch.stack(cx_nc_loss)).item())
        cz_epoch_loss.append(torch.mean(torch.stack(cz_nc_loss)).item())
        decoder_epoch_loss.append(torch.mean(torch.stack(decoder_loss)).item())
        if params.hyperbolic:
            hyp_dec_loss.append(torch.mean(torch.stack(hyp_loss)).item())
        eucl_dec_loss.append(torch.mean(torch.stack(mse_losss)).item())
        print("Encoder decoder training done in epoch {}".format(epoch))
        if params.hyperbolic:
            print("Hyperbolic loss {}".format(hyp_dec_loss[-1]))
        else:
            print("Eucl mse loss {}".format(eucl_dec_loss[-1]))
        print("critic x loss {:.3f} critic z loss {:.3f} \ndecoder loss {:.3f}\n".format(
            cx_epoch_loss[-1], cz_epoch_loss[-1], decoder_epoch_loss[-1]))
        wall.append(time.perf_counter())
        actual_epoch += 1
        if (actual_epoch % 10 == 0) or (actual_epoch == (n_epochs - 1)):       # train.py:381 (cadence kept as is)
            torch.save(encoder, path + "/encoder_{}.pt".format(actual_epoch))
            torch.save(decoder, path + "/decoder_{}.pt".format(actual_epoch))
            torch.save(critic_x, path + "/critic_x_{}.pt".format(actual_epoch))
            torch.save(critic_z, path + "/critic_z_{}.pt".format(actual_epoch))
    return SimpleNamespace(cx=cx_epoch_loss, cz=cz_epoch_loss, dec=decoder_epoch_loss, hyper=hyp_dec_loss, mse=eucl_dec_loss, wall=wall, repairs=0)


def model_path(params):
    """Directory naming of train.py:428-437."""
    kind = "hyper" if params.hyperbolic else "eucl"
    base = f"./trained_models/models_{kind}_{params.dataset}_{str(params.epochs)}_{str(params.lr)}/{params.dataset}"
    return base if params.signal == "multivariate" else f"{base}/{params.signal}"


def resume_ckpt(params):
    """train.py:388-406 with the undefined ``resume_path`` of the reference resolved to the model directory."""
    path = model_path(params) + "/"
    load = lambda n: torch.load(path + n, weights_only=False).cuda().train()
    print("model resumed from {}".format(path))
    return load("encoder.pt"), load("decoder.pt"), load("critic_x.pt"), load("critic_z.pt")


def train(train_loader, params, config_path):
    """train.py:409-466."""
    params.latent_space_dim = 20
    encoder = tadgan.Encoder(params.signal_shape, params.latent_space_dim).cuda().train()
    decoder = tadgan.Decoder(params.signal_shape, params.latent_space_dim, params.hyperbolic).cuda().train()
    critic_x = tadgan.CriticX(params.signal_shape, params.latent_space_dim).cuda().train()
    critic_z = tadgan.CriticZ(params.latent_space_dim).cuda().train()
    PATH = model_path(params)
    os.makedirs(PATH, exist_ok=True)
    if config_path and os.path.exists(config_path):
        import shutil
        shutil.copyfile(config_path, os.path.join(PATH, "config.yaml"))
    if params.resume:
        encoder, decoder, critic_x, critic_z = resume_ckpt(params)
    train_tadgan(train_loader, encoder, decoder, critic_x, critic_z, n_epochs=params.epochs, params=params, path=PATH)
    torch.save(encoder, PATH + "/encoder.pt")
    torch.save(decoder, PATH + "/decoder.pt")
    torch.save(critic_x, PATH + "/critic_x.pt")
    torch.save(critic_z, PATH + "/critic_z.pt")
    return encoder, decoder, critic_x, critic_z, PATH


# ------------------------------------------------------------------------------------------------ resident fast path
import contextlib


_ONE_THREAD_LOCK = __import__("threading").Lock()
_ONE_THREAD_STATE = {"depth": 0, "saved": None}
_ONE_THREAD_LOCAL = __import__("threading").local()


@contextlib.contextmanager
def _one_host_thread():
    """Module construction is thousands of tiny CPU tensor ops (fills and copies of <= 40 000 floats).  On a many-core host torch's
    intra-op pool wakes every core for the ones above its grain size -- torch.zeros of a 63 000-float arena took 1.7 ms on the GPU box's
    256-core host, half of train_signals_resident's set-up -- so they run on the calling thread.  Overlapping uses (a nested call, two
    calls on two threads) must not leave the process at one thread whatever the order they end in: the value to go back to is taken ONCE,
    by the first use in flight (a later one would read the 1 the first has set wherever torch keeps the count per process; with OpenMP it
    is kept per thread), and every thread restores it when ITS outermost use ends."""
    tl = _ONE_THREAD_LOCAL
    with _ONE_THREAD_LOCK:
        if _ONE_THREAD_STATE["depth"] == 0:
            _ONE_THREAD_STATE["saved"] = torch.get_num_threads()
        _ONE_THREAD_STATE["depth"] += 1
        tl.depth = getattr(tl, "depth", 0) + 1
        if tl.depth == 1:
            tl.saved = _ONE_THREAD_STATE["saved"]
            torch.set_num_threads(1)
    try:
        yield
    finally:
        with _ONE_THREAD_LOCK:
            _ONE_THREAD_STATE["depth"] -= 1
            tl.depth -= 1
            if tl.depth == 0:
                torch.set_num_threads(tl.saved)


def _host_shuffle_generator(dev, seed, stream):
    """The device generator that draws a signal's shuffles when they cannot be drawn inside the captured epoch (more than
    Engine.SHUFFLE_MAX_WINDOWS windows): keyed by (run seed, the signal's stream number = first_signal + slot) -- ONE rule for
    train_tadgan_resident and train_signals_resident, so that a long signal trains the same way alone and inside a group, and two
    signals of a run never share their permutations.  (Round 5 changed this derivation -- train_tadgan_resident used ``seed & 0x7FFFFFFF`` before --,
    so runs and resumes of signals longer than SHUFFLE_MAX_WINDOWS windows do not reproduce the shuffles of builds older than ABI 6.)"""
    return torch.Generator(device=dev).manual_seed((int(seed) ^ (0x9E3779B97F4A7C15 * int(stream))) & 0x7FFFFFFFFFFFFFFF)


def _resident_epoch_means(rows, n_critics, n_batches):
    """(critic_x, critic_z, generator, hyperbolic-or-mse) epoch means of ONE model from its (iterations, 4) loss rows ON THE HOST: the
    one reduction both resident loops use (train_tadgan_resident for its model, train_signals_resident slot by slot), so that a
    signal's history does not depend on which of them trained it."""
    crit = rows[: 2 * n_critics * n_batches, 0].reshape(n_critics * n_batches, 2).mean(0)
    gl = rows[2 * n_critics * n_batches:].mean(0)
    return float(crit[0]), float(crit[1]), float(gl[0]), float(gl[1])


def train_tadgan_resident(dataset, encoder, decoder, critic_x, critic_z, n_epochs, params, path="", seed=None, log=print, first_signal=0):
    """train_tadgan (train.py:252-385) with everything on the device: the same epoch schedule -- 5 passes of
    (critic_x_iteration, critic_z_iteration) over the shuffled minibatches, then one pass of decoder_iteration -- as one
    ``hypad_train_epoch`` call per epoch (205 kernel launches, no host round trip inside).  ``dataset``: a
    ``hypad_amd.utils.dataloader.SignalDataset`` (trained straight from its scaled series, no window matrix), or an
    (N, signal_shape) array / tensor of windows.  Differences from the drop-in loop: latent noise, interpolation weights
    and dropout masks come from the device Philox generator (seeded by ``seed`` / torch.initial_seed()) instead of
    NumPy / torch CPU, and the DataLoader's shuffles are ``torch.randperm`` draws on the device (drop_last semantics:
    n_batches = N // batch_size, a fresh permutation per pass).  Checkpoint cadence and file names as the reference.
    ``first_signal``: the device random stream number of this model (Engine): the run then equals that signal's training inside a
    ``train_signals_resident`` group bit for bit."""
    B, S = params.batch_size, params.signal_shape
    dev = encoder.arena().device
    eng = Engine(S, params.latent_space_dim, B, bool(params.hyperbolic), 1, dev, lr=params.lr,
                 gen_weight_decay=1e-5 if params.hyperbolic else 0.0, gen_stabilize=10 if params.hyperbolic else 0,
                 seed=(torch.initial_seed() if seed is None else seed)
                 + _resume_salt(params.resume_epoch if getattr(params, "resume", False) else None), first_signal=first_signal)
    mods = {"enc": encoder, "dec": decoder, "cx": critic_x, "cz": critic_z}
    eng.adopt({k: m.arena() for k, m in mods.items()})
    eng.epoch_flags = int(getattr(params, "epoch_flags", 0))      # hypad_epoch_io.flags (A/B forms of the critic phase; tests)
    if hasattr(dataset, "window_view"):
        x, n_windows, stride = dataset.window_view(dev)
    else:
        x = torch.as_tensor(np.asarray(dataset), dtype=torch.float32).reshape(-1, S).to(dev).contiguous()
        n_windows, stride = x.shape[0], 0
    n_batches = n_windows // B
    if n_batches < 1:
        raise _C.HypadError(f"{n_windows} windows do not fill one batch of {B}")
    n_critics = 5
    gen = _host_shuffle_generator(dev, eng.seed, first_signal)
    perm_buf = torch.empty(n_critics + 1, n_batches * B, dtype=torch.int32, device=dev)
    history = SimpleNamespace(cx=[], cz=[], dec=[], hyper=[], mse=[], repairs=0)      # repairs: resident critic launches that gave up and were repeated
    actual_epoch = 0
    if getattr(params, "resume", False):
        n_epochs = n_epochs - params.resume_epoch
        actual_epoch = params.resume_epoch + 1
    # one uniform permutation per pass: drawn by the library inside the captured epoch where it can (<= 4096 windows:
    # hypad_epoch_shuffles), else by one batched torch sort (argsort of uniform keys) into the buffer the epoch reads
    in_graph = n_windows <= eng.SHUFFLE_MAX_WINDOWS
    iters = (2 * n_critics + 1) * n_batches
    back = [torch.empty(iters * 4 + 8, dtype=torch.float32).pin_memory() for _ in range(2)]        # losses | counters (as bits), by epoch parity
    done = [torch.cuda.Event() for _ in range(2)]
    writer = _CheckpointWriter(dev, mods)
    snaps, state = {}, {"repaired_until": -1, "repaired": {}}
    first_epoch = actual_epoch
    saves = lambda e: bool(path) and (((first_epoch + e + 1) % 10 == 0) or ((first_epoch + e + 1) == (n_epochs - 1)))      # train.py:381 (cadence kept as is)

    def enqueue(e):
        if not in_graph:
            perm = torch.rand(n_critics + 1, n_windows, device=dev, generator=gen).argsort(dim=1)[:, : n_batches * B]
            perm_buf.copy_(perm)
        # the epoch is a fixed launch sequence: captured once as a hipGraph, replayed every epoch (Engine.train_epoch_graph)
        losses = eng.train_epoch_graph(x, perm_buf, n_batches, n_critics, True, x_row_stride=stride, shuffle_windows=n_windows if in_graph else 0)
        b = back[e % 2]
        b[: iters * 4].copy_(losses.view(-1), non_blocking=True)
        b[iters * 4:].view(torch.int32).copy_(eng.counters, non_blocking=True)
        done[e % 2].record()
        if saves(e):
            snaps[e] = writer.snapshot()                         # epoch e's weights, before epoch e + 1 is queued (_CheckpointWriter)

    def finish(e):
        """Epoch e's losses on the host -- with its shuffles drawn inside the captured epoch the next epoch is already queued behind it.
        Did its resident critic launch complete?  If one of its bounded waits gave up (a CU withheld by a CU mask / a shared device),
        the launches behind it were no-ops: check_status restores the critics and repeats the queued epochs with one launch per critic
        iteration, for good."""
        done[e % 2].synchronize()
        b = back[e % 2]
        if e > state["repaired_until"]:
            if int(b[iters * 4:].view(torch.int32)[4]) != 0:
                def redo(i):
                    state["repaired"][e + i] = eng._last_epoch["losses"].detach().cpu().view(-1)
                    if e + i in snaps:
                        snaps[e + i] = writer.snapshot()
                eng.check_status(on_epoch=redo)
                state["repaired_until"] = max(state["repaired"]) if state["repaired"] else e
                history.repairs += 1
            else:
                eng.confirm_epochs(1)
        rows = (state["repaired"].pop(e) if e in state["repaired"] else b[: iters * 4]).view(iters, 4)
        cx_, cz_, dec_, aux_ = _resident_epoch_means(rows, n_critics, n_batches)      # (the reduction on the host: one copy per epoch)
        history.cx.append(cx_); history.cz.append(cz_); history.dec.append(dec_)
        (history.hyper if params.hyperbolic else history.mse).append(aux_)
        if log:
            log("epoch {}: critic x loss {:.3f} critic z loss {:.3f} decoder loss {:.3f} {} {:.5f}".format(
                e, history.cx[-1], history.cz[-1], history.dec[-1], "hyperbolic loss" if params.hyperbolic else "mse", aux_))
        if saves(e):
            ae = first_epoch + e + 1
            writer.submit(snaps.pop(e), {k: path + "/{}_{}.pt".format(nm, ae) for k, nm in (("enc", "encoder"), ("dec", "decoder"), ("cx", "critic_x"), ("cz", "critic_z"))})

    try:
        for epoch in range(n_epochs):
            enqueue(epoch)
            if not in_graph:                        # host-drawn shuffles: a repair repeats epochs from the buffer as it is then -- one epoch at a time
                finish(epoch)
            elif epoch > 0:
                finish(epoch - 1)
        if in_graph and n_epochs > 0:
            finish(n_epochs - 1)
    finally:
        writer.close()                              # every checkpoint file is complete when the call returns
    return history


def train_resident(dataset, params, config_path=None, seed=None, log=print, first_signal=0):
    """train.train (train.py:409-466) on the resident fast path; returns (encoder, decoder, critic_x, critic_z, PATH)."""
    params.latent_space_dim = 20
    encoder = tadgan.Encoder(params.signal_shape, params.latent_space_dim).cuda().train()
    decoder = tadgan.Decoder(params.signal_shape, params.latent_space_dim, params.hyperbolic).cuda().train()
    critic_x = tadgan.CriticX(params.signal_shape, params.latent_space_dim).cuda().train()
    critic_z = tadgan.CriticZ(params.latent_space_dim).cuda().train()
    PATH = model_path(params)
    os.makedirs(PATH, exist_ok=True)
    if config_path and os.path.exists(config_path):
        import shutil
        shutil.copyfile(config_path, os.path.join(PATH, "config.yaml"))
    if getattr(params, "resume", False):
        encoder, decoder, critic_x, critic_z = resume_ckpt(params)
    history = train_tadgan_resident(dataset, encoder, decoder, critic_x, critic_z, params.epochs, params, PATH, seed, log, first_signal)
    for name, m in (("encoder", encoder), ("decoder", decoder), ("critic_x", critic_x), ("critic_z", critic_z)):
        torch.save(m, PATH + "/{}.pt".format(name))
    return encoder, decoder, critic_x, critic_z, PATH, history


# ------------------------------------------------------------------------------------------------ many signals (one model per signal)
GROUP_MODELS = 32          # models advanced by one launch sequence (Engine): 2 * 32 * batch/16 critic workgroups = every CU at batch 64


def plan_signal_groups(n_windows, batch, world=1, rank=0, group=GROUP_MODELS):
    """Which signals this rank trains, and in which launch groups.  The reference trains one model per signal, nothing couples two
    signals (train.py:428-437): models whose epochs have the same number of minibatches can share every launch (grid.y = model;
    their window counts may differ: each model shuffles its own windows).  All signals of the call are ordered by (minibatches per
    epoch, position in the list); a signal's rank in that order is its *stream number* (Engine.first_signal + slot): its latent
    vectors, interpolation weights, dropout masks and shuffles are keyed by it, so its training does not depend on the world size
    or on what it is grouped with.  Each run of equal batch counts is cut into contiguous, balanced pieces, one per rank (ranks
    take turns getting the longer pieces), and a rank's piece into groups of at most ``group`` models.
    Returns [(first_stream, [signal indices])] for this rank and the {signal index: stream number} map of the whole call."""
    nb = [int(n) // int(batch) for n in n_windows]
    for i, b in enumerate(nb):
        if b < 1:
            raise _C.HypadError(f"signal {i}: {n_windows[i]} windows do not fill one batch of {batch}")
    order = sorted(range(len(nb)), key=lambda i: (nb[i], i))
    stream = {sig: p for p, sig in enumerate(order)}
    groups, start, turn = [], 0, 0
    while start < len(order):
        end = start
        while end < len(order) and nb[order[end]] == nb[order[start]]:
            end += 1
        n = end - start
        base, extra = divmod(n, world)
        pos = start
        for k in range(world):                      # piece k of this run goes to rank (k + turn) % world
            size = base + (1 if k < extra else 0)
            if (k + turn) % world == rank:
                for g0 in range(pos, pos + size, group):
                    groups.append((g0, order[g0: min(g0 + group, pos + size)]))
            pos += size
        turn = (turn + extra) % world
        start = end
    return groups, stream


def train_signals_resident(datasets, params, names=None, seed=None, init_seed=None, group=None, log=print, save=True, process_group=None):
    """One TadGAN per signal (train.py:409-466 once per signal, as the reference's users run it in a loop over main.py) for a LIST of
    signals, sharded over the ranks of ``torch.distributed`` (no collective on the data path; the per-signal
    metrics are gathered at the end) and advanced in groups of up to 32 models per launch sequence on each GPU.

    datasets: per signal a ``SignalDataset`` / an (N, signal_shape) array of windows -- lengths may differ.  names: per-signal
    ``params.signal`` (checkpoint directories follow train.py:428-437: .../{dataset}/{signal}/encoder.pt, encoder_{epoch}.pt ...).
    Model i is initialised under ``torch.manual_seed(init_seed + i)`` (reference construction order, train.py:415-426) and trained
    with the device random streams of its stream number (plan_signal_groups): the same bits as
    ``train_resident(datasets[i], ..., seed=seed, first_signal=stream[i])`` after that manual_seed, whatever the world size.
    Returns {name: {"path", "history", "final", "stream", "rank"}} for ALL signals on every rank (histories of remote signals come
    through ``gather_signal_metrics``), and the local models as ``result[name]["modules"]`` for the signals this rank trained."""
    import copy
    from . import parallel as par
    import torch.distributed as dist
    B, S = params.batch_size, params.signal_shape
    params.latent_space_dim = 20
    L, hyp = params.latent_space_dim, bool(params.hyperbolic)
    names = list(names) if names is not None else [f"signal{i}" for i in range(len(datasets))]
    if len(names) != len(datasets) or len(set(names)) != len(names):
        raise _C.HypadError("one distinct name per signal")
    world = dist.get_world_size(process_group) if par._group_active(process_group) else 1
    rank = dist.get_rank(process_group) if world > 1 or par._group_active(process_group) else 0
    mats = [None] * len(datasets)

    def windows(i):
        if mats[i] is None:
            d = datasets[i]
            m = np.asarray(d.X if hasattr(d, "X") else d, dtype=np.float64).reshape(len(d), -1)
            if m.shape[1] != S:
                raise _C.HypadError(f"signal {names[i]}: windows of {m.shape[1]} values, params.signal_shape is {S}")
            mats[i] = m
        return mats[i]
    counts = [len(d) for d in datasets]
    plan, stream = plan_signal_groups(counts, B, world, rank, group or GROUP_MODELS)
    seed = torch.initial_seed() if seed is None else int(seed)
    init_seed = seed if init_seed is None else int(init_seed)
    dev = torch.device("cuda", torch.cuda.current_device())
    n_critics, n_epochs = 5, params.epochs
    engines = []
    for first, members in plan:
        k = len(members)
        nb = counts[members[0]] // B
        eng = Engine(S, L, B, hyp, k, dev, lr=params.lr, gen_weight_decay=1e-5 if hyp else 0.0, gen_stabilize=10 if hyp else 0, seed=seed,
                     first_signal=first)
        eng.epoch_flags = int(getattr(params, "epoch_flags", 0))      # hypad_epoch_io.flags (A/B forms of the critic phase; tests)
        nmax = max(counts[i] for i in members)
        templates = []
        x = torch.zeros(k, nmax, S, dtype=torch.float32, device=dev)
        with _one_host_thread():
            for slot, i in enumerate(members):
                torch.manual_seed(init_seed + i)                 # train.py:415-426 construction order
                mods = dict(enc=tadgan.Encoder(S, L), dec=tadgan.Decoder(S, L, hyp), cx=tadgan.CriticX(S, L), cz=tadgan.CriticZ(L))
                templates.append(mods)                           # (the signal's module objects: checkpoints and the result are written through them)
                _set_requires_grad((mods["enc"], mods["dec"]), True)     # what the reference's loop leaves behind (train.py:331-340): set here, once, so
                _set_requires_grad((mods["cx"], mods["cz"]), False)      # that the returned modules and every checkpoint carry them with or without `save`
                x[slot, : counts[i]] = torch.from_numpy(windows(i)).to(torch.float32)
            for net in ("enc", "dec", "cx", "cz"):               # the group's initial weights: ONE upload per network (a module's host arena IS the
                host = torch.stack([t[net]._arena for t in templates])      # engine's row layout: arena.py / hypad_param_info), not 54 tensor copies per model
                if host.shape != eng.params[net].shape:
                    raise _C.HypadError(f"arena layout mismatch for {net}: {tuple(host.shape)} vs {tuple(eng.params[net].shape)}")
                eng.params[net].copy_(host)
        in_graph = nmax <= eng.SHUFFLE_MAX_WINDOWS
        ri = torch.empty(k, n_critics + 1, nb * B, dtype=torch.int32, device=dev)
        gens = None if in_graph else [_host_shuffle_generator(dev, eng.seed, first + s) for s in range(k)]
        engines.append(dict(eng=eng, members=members, nb=nb, x=x, ri=ri, in_graph=in_graph, gens=gens, counts=[counts[i] for i in members], templates=templates,
                            back=[torch.empty(k * (2 * n_critics + 1) * nb * 4 + 8, dtype=torch.float32).pin_memory() for _ in range(2)],
                            done=[torch.cuda.Event() for _ in range(2)], snaps={}))
    torch.manual_seed(seed)
    hist = {names[i]: SimpleNamespace(cx=[], cz=[], dec=[], hyper=[], mse=[], repairs=0) for _, ms in plan for i in ms}
    paths = {}
    for _, ms in plan:
        for i in ms:
            p = copy.copy(params)
            p.signal = names[i]
            paths[names[i]] = model_path(p)
            if save:
                os.makedirs(paths[names[i]], exist_ok=True)

    NETS = ("enc", "dec", "cx", "cz")
    FILES = dict(enc="encoder", dec="decoder", cx="critic_x", cz="critic_z")
    writer = _CheckpointWriter(dev, max_jobs=2)
    saves = lambda e: save and (((e + 1) % 10 == 0) or ((e + 1) == (n_epochs - 1)))        # train.py:381 (cadence kept as is; e + 1 = actual_epoch)
    # Every group's epoch is one graph replay; its per-iteration losses and the counters come back in ONE copy per group -- looked at
    # one epoch later, when the next epoch is already queued -- and each model's epoch means are taken from them on the host
    # (_resident_epoch_means, slot by slot: the bits train_tadgan_resident gets for its one model).  The round-4 loop read 4 values
    # per model and epoch with a synchronising float() each: 9.2 ms per epoch of 32 models against 6.0 for the launches alone.
    # Groups whose shuffles are drawn on the host (signals beyond the in-graph sort's 4 096 windows) are finished epoch by epoch: a
    # repair repeats epochs from the buffers as they are then.
    pipelined = all(g["in_graph"] for g in engines)
    # Small groups -- signals of many different lengths leave plan_signal_groups with few models per batch count -- are latency-bound
    # one by one (an epoch of 4 models takes 3.2 ms, of 32 models 5.8): dealt over LANES (streams that really run beside each other,
    # hypad_amd/streams.py) their epochs overlap.  Measured, epochs of all groups: 8 groups of 4 models 25.5 -> 8.9 ms on four lanes, 8 x 1
    # model 23.2 -> 6.4, 4 x 8 models 13.3 -> 7.9 on two.  At most 16 models run at any time (their resident critic launches hold 8 CUs
    # per model and need the other half of the chip for their record producers), so a group of more than 8 models keeps the GPU alone.
    from contextlib import nullcontext
    biggest = max((len(g["members"]) for g in engines), default=1)
    active_limit = max(1, torch.cuda.get_device_properties(dev).multi_processor_count // 16)      # models whose critics fill half the CUs: 16 on an MI355X
    n_lanes = max(1, min(4, active_limit // biggest, len(engines))) if len(engines) > 1 else 1
    if getattr(params, "lanes", None) is not None:               # (A/B timing, tests)
        n_lanes = max(1, min(int(params.lanes), n_lanes))
    lane_streams = []
    if n_lanes > 1:
        from . import streams as _streams
        lane_streams = _streams.lanes(n_lanes, dev)
        here = torch.cuda.current_stream()
        for st in lane_streams:
            st.wait_stream(here)                                 # (the groups' data went up on this stream)
    for i, g in enumerate(engines):
        g["lane"] = lane_streams[i % n_lanes] if lane_streams else None
        if lane_streams:
            # several resident critic launches at once: their workgroups in id order (dealt evenly over the XCDs, 16 per XCD at the 16-model
            # limit) instead of a critic's chunks packed onto one XCD -- packed, every launch would want the SAME first XCDs (32 workgroups
            # on a 32-CU XCD at the limit: no slack for whatever else the dispatcher put there).  Same bits (hypad.h: HYPAD_EPOCH_ID_ORDER).
            g["eng"].epoch_flags |= _C.EPOCH_ID_ORDER
    on_lane = lambda g: torch.cuda.stream(g["lane"]) if g["lane"] is not None else nullcontext()

    def enqueue(e):
        for g in engines:
            with on_lane(g):
                enqueue_group(g, e)

    def enqueue_group(g, e):
        eng = g["eng"]
        if not g["in_graph"]:
            for s_, c in enumerate(g["counts"]):
                g["ri"][s_].copy_(torch.rand(n_critics + 1, c, device=dev, generator=g["gens"][s_]).argsort(dim=1)[:, : g["nb"] * B])
        g["losses"] = eng.train_epoch_graph(g["x"], g["ri"], g["nb"], n_critics, True, shuffle_windows=g["counts"] if g["in_graph"] else 0)
        b, n = g["back"][e % 2], g["losses"].numel()
        b[:n].copy_(g["losses"].view(-1), non_blocking=True)
        b[n:].view(torch.int32).copy_(eng.counters, non_blocking=True)
        g["done"][e % 2].record()
        if saves(e):
            g["snaps"][e] = writer.snapshot(eng.params)          # epoch e's weights, before epoch e + 1 is queued

    def finish(e):
        for g in engines:
            eng, k, nb = g["eng"], len(g["members"]), g["nb"]
            g["done"][e % 2].synchronize()
            b, n = g["back"][e % 2], g["losses"].numel()
            if e > g.get("repaired_until", -1):
                if int(b[n:].view(torch.int32)[4]) != 0:             # the resident critic launch gave up: this epoch and the one queued behind it were no-ops
                    kept = {}

                    def redo(i, g=g, e=e, kept=kept):
                        kept[e + i] = g["losses"].detach().cpu()     # (the repeat's losses: the buffer is the next repeat's too)
                        if e + i in g["snaps"]:
                            g["snaps"][e + i] = writer.snapshot(g["eng"].params)
                    with on_lane(g):                                 # (the repeats on the group's own stream, behind whatever it still holds)
                        eng.check_status(on_epoch=redo)
                    g["repaired"] = kept
                    g["repaired_until"] = max(kept) if kept else e
                    for i in g["members"]:                           # (every model of the group went through the repeat)
                        hist[names[i]].repairs += 1
                else:
                    eng.confirm_epochs(1)
            rows = g["repaired"][e].view(k, -1, 4) if e <= g.get("repaired_until", -1) and e in g.get("repaired", {}) else b[:n].view(k, -1, 4)
            for slot, i in enumerate(g["members"]):
                h = hist[names[i]]
                cx_, cz_, dec_, aux_ = _resident_epoch_means(rows[slot], n_critics, nb)
                h.cx.append(cx_); h.cz.append(cz_); h.dec.append(dec_)
                (h.hyper if hyp else h.mse).append(aux_)
            if saves(e):
                pick = [(g["templates"][slot][net], net, slot, paths[names[i]] + "/{}_{}.pt".format(FILES[net], e + 1))
                        for slot, i in enumerate(g["members"]) for net in NETS]
                writer.submit(g["snaps"].pop(e), None, pick)
        if log:
            mine = [names[i] for g in engines for i in g["members"]]
            log("epoch {}: {} signal(s) on rank {}: mean critic x loss {:.3f} critic z loss {:.3f} decoder loss {:.3f}".format(
                e, len(mine), rank, *(float(np.mean([getattr(hist[n], k)[-1] for n in mine])) if mine else float("nan") for k in ("cx", "cz", "dec"))))

    try:
        for epoch in range(n_epochs):
            enqueue(epoch)
            if not pipelined:
                finish(epoch)
            elif epoch > 0:
                finish(epoch - 1)
        if pipelined and n_epochs > 0:
            finish(n_epochs - 1)
        # the final weights: into the signals' own module objects (the result), and -- train.py:461-464 -- their files
        last = {}
        for g in engines:
            with on_lane(g):
                last[id(g)] = writer.snapshot(g["eng"].params)
        for st in lane_streams:
            torch.cuda.current_stream().wait_stream(st)         # (what follows on this stream reads the groups' weights)
        if save:
            for g in engines:
                writer.submit(last[id(g)], None, [(g["templates"][slot][net], net, slot, paths[names[i]] + "/{}.pt".format(FILES[net]))
                                                  for slot, i in enumerate(g["members"]) for net in NETS])
    finally:
        writer.close()
    local = {}
    for g in engines:
        snap = last[id(g)][0]
        for slot, i in enumerate(g["members"]):
            mods = []
            for net in NETS:
                m = g["templates"][slot][net]
                if not next(m.parameters()).is_cuda:
                    m.to(dev)
                m.arena().data.copy_(snap[net][slot])
                mods.append(m.train())
            h = hist[names[i]]
            local[names[i]] = {"path": paths[names[i]], "history": vars(h), "stream": stream[i], "rank": rank,
                               "final": {k: (getattr(h, k)[-1] if getattr(h, k) else None) for k in ("cx", "cz", "dec", "hyper", "mse")}}
            g.setdefault("mods", {})[names[i]] = mods
    merged = par.gather_signal_metrics(local, process_group)
    for g in engines:
        for n, mods in g.get("mods", {}).items():
            merged[n] = dict(merged[n], modules=mods)
    return merged
