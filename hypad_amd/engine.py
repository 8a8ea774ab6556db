"""Device-resident training engine: n_signals independent TadGAN models stepped side by side.

The reference trains one model per signal (train.py:428-437) and nothing couples two signals, so the
natural MI355X unit is a *group* of models advanced by the same kernel launches (grid.y = signal).  The
engine owns, per network, one (n_signals, param_count) arena plus Adam moments, the device counters and the
workspace, and exposes the three iterations and a whole epoch (train.py:299-356) over a window matrix that
stays resident in HBM.  ``hypad_amd.train`` wraps it with n_signals = 1 around user-visible nn.Modules.
"""
import ctypes
import os
import logging

import numpy as np
import torch

from . import _C

NETS = ("enc", "dec", "cx", "cz")
_NET_ID = dict(enc=_C.NET_ENCODER, dec=_C.NET_DECODER, cx=_C.NET_CRITIC_X, cz=_C.NET_CRITIC_Z)


class Engine:
    def __init__(self, signal_shape, latent_dim, batch, hyperbolic, n_signals=1, device="cuda", lr=5e-4, betas=(0.9, 0.999),
                 eps=1e-8, gen_weight_decay=1e-5, gen_stabilize=10, seed=0, first_signal=0):
        """first_signal: model s draws its device random streams (and, through draw_shuffles, its shuffles) as stream
        first_signal + s (hypad_dims.first_signal): model k of a group == a single model with first_signal + k, bit for bit."""
        if batch % 16:
            raise _C.HypadError("batch size must be a multiple of 16 (row tiles of the fused kernels)")
        self.S, self.L, self.B, self.hyperbolic, self.n = int(signal_shape), int(latent_dim), int(batch), bool(hyperbolic), int(n_signals)
        self.device = torch.device(device)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.gen_wd, self.gen_stab = float(gen_weight_decay), int(gen_stabilize or 0)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.count = {k: _C.lib.hypad_param_count(_NET_ID[k], self.S, self.L, int(self.hyperbolic)) for k in NETS}
        z = lambda k: torch.zeros(self.n, self.count[k], dtype=torch.float32, device=self.device)
        self.params = {k: z(k) for k in NETS}
        self.exp_avg = {k: z(k) for k in NETS}
        self.exp_avg_sq = {k: z(k) for k in NETS}
        self.counters = torch.zeros(8, dtype=torch.int32, device=self.device)    # [0..2] optimizer steps, [3] rng tick, [4] status word
        self.epoch_flags = 0                     # hypad_epoch_io.flags of every epoch (EPOCH_PER_ITERATION after a recovery)
        self._last_epoch = None                  # arguments of the last train_epoch / train_epoch_graph call
        self._pending = []                       # ... of every epoch queued since the last check_status that found nothing (it re-runs the lost ones)
        self._steps_checked = (0, 0)             # counters[0], [2] (critic_x / generator optimizer steps) at that check
        self.first_signal = int(first_signal)
        self.dims = _C.Dims(self.S, self.L, self.B, int(self.hyperbolic), self.n, self.first_signal)
        nbytes = _C.lib.hypad_train_workspace_bytes(ctypes.byref(self.dims))
        if nbytes == 0:
            raise _C.HypadError("unsupported dimensions for the fused training kernels")
        self.workspace = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
        self._ws_bytes = nbytes

    def _grow_workspace(self, nbytes):
        if nbytes > self._ws_bytes:
            self.workspace = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            self._ws_bytes = nbytes
            self._drop_graphs()                  # captured epochs hold the old workspace's address

    MAX_AUX_STREAMS = 7

    def _aux_streams_wanted(self):
        """``self.aux_streams`` if set, else HYPAD_AUX_STREAMS, else the measured default: one auxiliary stream (two groups) from 28
        models per GPU on, none below.  Measured on MI355X, epoch ms with 0 / 1 / 2 / 3 / 7 auxiliary streams: 8 models 3.33 / 3.49 /
        - / 3.57 / 5.49; 12: 3.66 / 3.58 / 3.86; 16: 4.22 / 4.09 / 4.43; 24: 5.22 / 5.49 / 5.13; 32: 6.47 / 5.97 / 6.69 / 6.13 / 8.69 --
        a second hardware queue costs ~7 us per dependent launch, which only the 32-model launches (58 us generator launches that
        fill the chip's CUs, 54 us optimizer launches that fill its memory pipes) earn back."""
        if hasattr(self, "aux_streams"):
            return int(self.aux_streams)
        if "HYPAD_AUX_STREAMS" in os.environ:
            return int(os.environ["HYPAD_AUX_STREAMS"])
        return 1 if self.n >= 28 else 0

    def _aux_stream_args(self):
        """(array of raw stream handles, count) for hypad_epoch_io: with several signals (models) per GPU the generator phase runs
        them in groups on side streams, a group's optimizer launch overlapping another group's generator launch (include/hypad.h,
        ABI 4).  0 (or one signal) keeps everything on the caller's stream; same bits either way."""
        want = min(self.MAX_AUX_STREAMS, self.n - 1, self._aux_streams_wanted())
        if want <= 0:
            return None, 0
        pool = self.__dict__.setdefault("_aux_pool", [])
        while len(pool) < want:
            pool.append(torch.cuda.Stream(device=self.device))
        arr = (ctypes.c_void_p * want)(*(s.cuda_stream for s in pool[:want]))
        self._aux_arr = arr                       # (kept alive for the duration of the call)
        return ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p)), want

    ENC_TABLE_MIN_SIGNALS = 14

    def _enc_table_args(self, x, x_row_stride):
        """(pointer, rows) of hypad_epoch_io.enc_table: encoder(x) once per window row instead of once per critic pass, from
        ``ENC_TABLE_MIN_SIGNALS`` models per engine on (``self.enc_table = True / False`` forces it) -- below that the critic phase is not
        bound by its record producers and the extra launch only costs (measured, ms per epoch without / with: 8 models 3.32 / 3.33,
        12: 3.66 / 3.68, 14: 4.06 / 3.91, 16: 4.25 / 3.97, 24: 5.23 / 5.09, 32: 5.99 / 5.76)."""
        want = self.__dict__.get("enc_table")
        if want is None:
            want = self.n >= self.ENC_TABLE_MIN_SIGNALS
        if not want:
            return None, 0
        rows = int(x.shape[1]) - (self.S - 1 if int(x_row_stride) == 1 else 0)      # window rows per model: matrix rows, or positions of the series view
        if rows < 1:
            return None, 0
        t = self.__dict__.get("_enc_table")
        if t is None or t.shape[1] != rows:
            t = self._enc_table = torch.empty(self.n, rows, self.L, dtype=torch.float32, device=self.device)
            self._drop_graphs()                  # captured epochs hold the old table's address
        return t.data_ptr(), rows

    def _drop_graphs(self):
        self.__dict__.pop("_graphs", None)

    def _graph_state_key(self):
        """Everything a captured epoch froze besides its call arguments: the arenas', moments', counters' and workspace's
        addresses and the optimizer scalars (passed by value)."""
        ptrs = tuple(d[k].data_ptr() for d in (self.params, self.exp_avg, self.exp_avg_sq) for k in NETS)
        return ptrs + (self.counters.data_ptr(), self.workspace.data_ptr(), self._ws_bytes, self.lr, self.betas, self.eps, self.gen_wd,
                       self.gen_stab, self.epoch_flags, self._aux_streams_wanted(), self.__dict__.get("enc_table"))

    # ---- weights in / out ------------------------------------------------------------------------------
    def catalogue(self, net):
        return _C.param_catalogue(_NET_ID[net], self.S, self.L, self.hyperbolic)[0]

    def load_state_dict(self, net, sd, signal=0):
        for name, off, shape in self.catalogue(net):
            n = int(np.prod(shape))
            self.params[net][signal, off:off + n].copy_(sd[name].detach().reshape(-1).to(torch.float32))

    def state_dict(self, net, signal=0):
        out = {}
        for name, off, shape in self.catalogue(net):
            n = int(np.prod(shape))
            out[name] = self.params[net][signal, off:off + n].view(shape).clone()
        return out

    def adopt(self, arenas, exp_avg=None, exp_avg_sq=None):
        """Use caller-owned flat arenas (n_signals == 1): the nn.Module views of hypad_amd.train."""
        for k, t in arenas.items():
            self.params[k] = t.view(1, -1)
        for k, t in (exp_avg or {}).items():
            self.exp_avg[k] = t.view(1, -1)
        for k, t in (exp_avg_sq or {}).items():
            self.exp_avg_sq[k] = t.view(1, -1)
        self._st = None
        self._drop_graphs()

    # ---- C structs -------------------------------------------------------------------------------------
    def _nets(self, d):
        return _C.Nets(*(d[k].data_ptr() for k in NETS))

    def _state(self):
        """hypad_train_state for the next call.  The pointer part is rebuilt only after adopt() (the arenas, moments and counters
        are allocated once and updated in place); the scalars are refreshed on every call."""
        st = self.__dict__.get("_st")
        if st is None:
            st = self._st = _C.TrainState(self._nets(self.params), self._nets(self.exp_avg), self._nets(self.exp_avg_sq),
                                          self.counters.data_ptr(), 0.0, 0.0, 0.0, 0.0, 0.0, 0)
        st.lr, st.beta1, st.beta2, st.eps, st.gen_weight_decay, st.gen_stabilize = self.lr, self.betas[0], self.betas[1], self.eps, self.gen_wd, self.gen_stab
        return st

    def _check_x(self, x, x_row_stride=0):
        """x: (n_signals, n_windows, S) window matrices, or -- with x_row_stride=1 -- (n_signals, T) scaled series whose
        windows are the overlapping rows x[n : n + S] (SignalDataset.window_view: nothing is materialised)."""
        _C.require_cuda(x, "x")
        if x_row_stride:
            if x.dim() == 1:
                x = x.unsqueeze(0)
            if x.dim() != 2 or x.shape[0] not in (1, self.n) or x.shape[1] < self.S or not x.is_contiguous():
                raise _C.HypadError(f"a series view must be a contiguous (n_signals, T >= {self.S}) tensor")
            return x, (0 if (x.shape[0] == 1 and self.n > 1) else x.shape[1])
        if x.dim() == 2:
            x = x.unsqueeze(0)
        if x.shape[0] not in (1, self.n) or x.shape[2] != self.S:
            raise _C.HypadError(f"x must be (n_signals, n_windows, {self.S})")
        stride = 0 if (x.shape[0] == 1 and self.n > 1) else x.shape[1] * x.shape[2]
        return x, stride

    def _iter(self, fn, x, row_index, z, alpha, train_mode, masks, x_row_stride=0):
        x, stride = self._check_x(x, x_row_stride)
        losses = torch.empty(self.n, 4, dtype=torch.float32, device=self.device)
        io = self.__dict__.get("_iter_io")              # one struct per engine, refilled per call (the drop-in loop calls this 319 times per epoch)
        if io is None:
            io = self._iter_io = _C.IterIO()
        p = lambda t: None if t is None else t.data_ptr()
        io.x, io.x_signal_stride, io.x_row_stride, io.row_index, io.z, io.alpha = x.data_ptr(), stride, int(x_row_stride), p(row_index), p(z), p(alpha)
        io.drop.train_mode, io.drop.masks, io.drop.seed, io.drop.offset = int(train_mode), p(masks), self.seed, 0
        io.losses, io.workspace, io.workspace_bytes = losses.data_ptr(), self.workspace.data_ptr(), self._ws_bytes
        st = self._state()
        _C.check(fn(ctypes.byref(self.dims), ctypes.byref(st), ctypes.byref(io), _C.stream()), fn.__name__)
        return losses

    # Critic iterations as one-iteration phases of the hoisted form (include/hypad.h, hypad_critic_x_iteration: taken when the workspace
    # has room).  GPU time per call 45 / 29 us instead of 66 / 39 (critic_x / critic_z), but five launches instead of three: the
    # reference-style Python loop (bench.py `drop_in`) is bound by the host and comes out SLOWER with it (70.5 against 62.5 us per
    # iteration), so it is off by default; worth turning on where the calls are captured into a graph or enqueued from a faster host.
    iteration_phase = os.environ.get("HYPAD_ENGINE_ITER_PHASE", "0") == "1"

    def _room_for_iteration_phase(self):
        if self.iteration_phase and not self.__dict__.get("_iter_phase_room"):
            self._grow_workspace(_C.lib.hypad_epoch_workspace_bytes(ctypes.byref(self.dims), 1, 1))
            self._iter_phase_room = True

    def critic_x_iteration(self, x, row_index=None, z=None, alpha=None, train_mode=True, masks=None, x_row_stride=0):
        self._room_for_iteration_phase()
        return self._iter(_C.lib.hypad_critic_x_iteration, x, row_index, z, alpha, train_mode, masks, x_row_stride)

    def critic_z_iteration(self, x, row_index=None, z=None, alpha=None, train_mode=True, masks=None, x_row_stride=0):
        self._room_for_iteration_phase()
        return self._iter(_C.lib.hypad_critic_z_iteration, x, row_index, z, alpha, train_mode, masks, x_row_stride)

    def decoder_iteration(self, x, row_index=None, z=None, train_mode=True, masks=None, x_row_stride=0):
        return self._iter(_C.lib.hypad_decoder_iteration, x, row_index, z, None, train_mode, masks, x_row_stride)

    def critic_phase_persistent(self):
        """True when train_epoch runs the critic phase as one resident launch (hypad_critic_phase_persistent; the engine's
        ``epoch_flags`` may ask for the per-iteration / per-minibatch forms instead)."""
        if self.epoch_flags & (_C.EPOCH_PER_ITERATION | _C.EPOCH_PER_MINIBATCH):
            return False
        return bool(_C.lib.hypad_critic_phase_persistent(ctypes.byref(self.dims)))

    def critic_phase_producers(self, n_iters):
        """True when that resident launch also produces the phase's records itself (hypad_critic_phase_producers)."""
        if not self.critic_phase_persistent() or (self.epoch_flags & _C.EPOCH_NO_PRODUCERS):
            return False
        return bool(_C.lib.hypad_critic_phase_producers(ctypes.byref(self.dims), int(n_iters)))

    def rng_fill(self, kind, n, tick, stream, signal=0, p_drop=0.0, seed=None):
        """n draws of one device random stream (hypad_rng_fill): kind 0 N(0,1), 1 U[0,1), 2 dropout keep-scale."""
        out = torch.empty(int(n), dtype=torch.float32, device=self.device)
        _C.check(_C.lib.hypad_rng_fill(int(kind), self.seed if seed is None else int(seed), int(tick), int(stream), int(signal),
                                       float(p_drop), out.data_ptr(), int(n), _C.stream()), "rng_fill")
        return out

    def epoch_records(self, n_batches, n_critics, critic):
        """The records the last hoisted train_epoch left in the workspace for critic 0 (x) / 1 (z), as a
        (n_signals, n_iters, batch/16, record_floats) view, plus their geometry (hypad_epoch_record_info)."""
        info = _C.RecordInfo()
        _C.check(_C.lib.hypad_epoch_record_info(ctypes.byref(self.dims), n_batches, n_critics, critic, ctypes.byref(info)), "record_info")
        n = self.n * n_batches * n_critics * (self.B // 16) * info.record_floats
        view = self.workspace[info.offset_floats: info.offset_floats + n].view(self.n, n_batches * n_critics, self.B // 16, info.record_floats)
        return view, info

    def profile_iteration(self, kind, x, row_index=None, train_mode=True):
        """Per-kernel milliseconds of one iteration (0 critic_x, 1 critic_z, 2 decoder, 3 critic_x || critic_z pair,
        4 = 145 iterations of train_epoch's hoisted critic phase: precompute, first launch / re-initialisation, mean time per
        iteration; 5 = the generator step's two kernels as an epoch launches them, 64 back-to-back launches each: mean per launch),
        HIP events on the current stream."""
        x, stride = self._check_x(x)
        if kind == 4:
            self._grow_workspace(_C.lib.hypad_epoch_workspace_bytes(ctypes.byref(self.dims), 145, 1))
        losses = torch.empty(290 * self.n, 4, dtype=torch.float32, device=self.device)
        drop = _C.Dropout(int(train_mode), None, self.seed, 0)
        io = _C.IterIO(x.data_ptr(), stride, 0, None if row_index is None else row_index.data_ptr(), None, None, drop,
                       losses.data_ptr(), self.workspace.data_ptr(), self._ws_bytes)
        st = self._state()
        out = (ctypes.c_float * 3)()
        _C.check(_C.lib.hypad_profile_iteration(int(kind), ctypes.byref(self.dims), ctypes.byref(st), ctypes.byref(io), out, 3,
                                                _C.stream()), "profile_iteration")
        return list(out)[: 2 if kind in (2, 5) else 3]

    NOISE_PLANES = ("z_cx", "alpha_cx", "z_cz", "alpha_cz", "z_gen", "masks_cx", "masks_cz", "masks_gen")

    SHUFFLE_MAX_WINDOWS = 4096

    def draw_shuffles(self, row_index, n_windows):
        """The DataLoader's shuffles of one epoch, drawn by the library (capturable, keyed by the engine's seed, the model's stream
        number and the device rng tick) into ``row_index``: (n_passes, n_batches * batch) int32 on the device -- ONE plane, every
        model of the engine sees the same shuffles (hypad_epoch_shuffles) -- or (n_signals, n_passes, n_batches * batch): a plane
        per model, each of permutations of its OWN window count (``n_windows``: an int, or one count per model) --
        hypad_epoch_shuffles_signals; plane s of a group == the plane of a single model with first_signal + s."""
        planes = row_index.shape[0] if row_index.dim() == 3 else 1
        n_passes, take = row_index.shape[-2:]
        if not row_index.is_contiguous():
            raise _C.HypadError("row_index must be contiguous")
        nw = self._window_counts(n_windows, planes)
        _C.check(_C.lib.hypad_epoch_shuffles_signals(row_index.data_ptr(), int(n_passes * take), planes, self.first_signal, nw.data_ptr(),
                                                     int(n_passes), int(take), self.seed ^ 0x5DEECE66D, self.counters.data_ptr(), _C.stream()),
                 "epoch_shuffles_signals")
        return row_index

    def _window_counts(self, n_windows, planes):
        """Device int32[planes] of window counts (cached: a captured epoch holds its address)."""
        counts = tuple(int(v) for v in (n_windows.tolist() if hasattr(n_windows, "tolist") else
                                        (n_windows if isinstance(n_windows, (list, tuple)) else [n_windows] * planes)))
        if len(counts) != planes or min(counts) < 1 or max(counts) > self.SHUFFLE_MAX_WINDOWS:
            raise _C.HypadError(f"draw_shuffles: {planes} window count(s) in [1, {self.SHUFFLE_MAX_WINDOWS}] expected")
        cache = self.__dict__.setdefault("_nw_cache", {})
        if counts not in cache:
            cache[counts] = torch.tensor(counts, dtype=torch.int32, device=self.device)
        return cache[counts]

    def train_epoch_graph(self, x, row_index, n_batches, n_critics=5, train_mode=True, losses=None, x_row_stride=0, shuffle_windows=0,
                          noise=None):
        """train_epoch captured once into a hipGraph and replayed: the epoch is a fixed launch sequence (device counters, device
        Philox, no host round trip; include/hypad.h: "every call may be captured"), ~62 launches at the reference
        configuration.  `x`, `row_index` and `losses` must be the SAME buffers on every call (their addresses are frozen
        in the graph; refill `row_index` in place with the epoch's shuffles -- or pass ``shuffle_windows`` = number of windows
        (<= 4096): the shuffles are then drawn INSIDE the captured sequence by hypad_epoch_shuffles, fresh at every replay, and
        `row_index` is only the buffer they land in).  ``noise``: injected planes as in train_epoch -- static device buffers too,
        refilled in place before every replay (the drop-in train_tadgan uploads an epoch's host-drawn z / alpha planes there).
        Returns `losses`."""
        x, _ = self._check_x(x, x_row_stride)
        iters = (2 * n_critics + 1) * n_batches
        if losses is None:
            losses = getattr(self, "_graph_losses", None)
            if losses is None or losses.shape != (self.n, iters, 4):
                losses = self._graph_losses = torch.empty(self.n, iters, 4, dtype=torch.float32, device=self.device)
        self._grow_workspace(_C.lib.hypad_epoch_workspace_bytes(ctypes.byref(self.dims), n_batches, n_critics))
        shuffle_key = tuple(shuffle_windows) if isinstance(shuffle_windows, (list, tuple)) else int(shuffle_windows)
        key = (x.data_ptr(), row_index.data_ptr(), tuple(row_index.shape), losses.data_ptr(), n_batches, n_critics, bool(train_mode), int(x_row_stride),
               self.seed, shuffle_key, self._graph_state_key(),
               tuple(sorted((k, t.data_ptr()) for k, t in (noise or {}).items() if t is not None)))
        graphs = self.__dict__.setdefault("_graphs", {})
        if key not in graphs:
            if shuffle_windows:                          # (the device copy of the window counts: made before, not inside, the capture)
                self._window_counts(shuffle_windows, row_index.shape[0] if row_index.dim() == 3 else 1)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):                    # (side stream: _C.stream() is torch's current stream inside the block)
                if shuffle_windows:
                    self.draw_shuffles(row_index, shuffle_windows)
                self.train_epoch(x, row_index, n_batches, n_critics, train_mode, losses=losses, x_row_stride=x_row_stride, noise=noise)
            graphs[key] = (g, x, row_index, losses, noise)      # (the buffers whose addresses the graph holds stay alive with it)
            # the capture itself did not execute anything
        graphs[key][0].replay()
        self._last_epoch = dict(x=x, row_index=row_index, n_batches=n_batches, n_critics=n_critics, train_mode=train_mode, losses=losses,
                                x_row_stride=x_row_stride, noise=noise)
        self._queued(dict(self._last_epoch), shuffle_windows)
        return losses

    MAX_PENDING = 4096

    def confirm_epochs(self, n=1):
        """The caller has read status 0 (counters[4], e.g. from its own read-back of the counters) BEHIND the n oldest queued epochs:
        they completed; check_status need not look at them again.  (What check_status does itself when it finds status 0 -- for callers
        that keep more than one epoch in flight and read the counters without synchronising the stream.)"""
        for _ in range(min(int(n), len(self._pending))):
            call, _ = self._pending.pop(0)
            self._steps_checked = (self._steps_checked[0] + call["n_batches"] * call["n_critics"], self._steps_checked[1] + call["n_batches"])

    def _queued(self, call, shuffle_windows=0):
        """Book-keeping for check_status: the epochs queued since the last check that found status 0."""
        if len(self._pending) >= self.MAX_PENDING:       # never checked: nothing more is recorded, and a failure cannot be repaired any more
            self._pending_overflow = True
            return
        self._pending.append((call, shuffle_windows))

    # ---- status channel of the resident critic launch (include/hypad.h: hypad_epoch_status / hypad_epoch_restore) -------------
    def status(self):
        """counters[4] after everything enqueued so far (synchronises the stream): 0, or the code of the bounded wait that gave up."""
        st = self._state()
        out = ctypes.c_int(0)
        _C.check(_C.lib.hypad_epoch_status(ctypes.byref(st), ctypes.byref(out), _C.stream()), "epoch_status")
        return out.value

    def check_status(self, recover=True, on_epoch=None):
        """Call where the host reads an epoch's losses.  If a resident critic launch gave up (a withheld CU: CU mask, partitioned or
        shared device), every launch behind it -- the rest of that epoch and ALL epochs queued after it -- was a no-op
        (fail-stop).  With ``recover`` the critics and counters are put back to the state the failed epoch began from and every
        epoch queued since (the failed one and those behind it, found from the optimizer-step counter) is repeated in order with
        one launch per critic iteration (same random streams: bit for bit healthy epochs in that form); every later epoch of this
        engine uses that form.  The repeats read the callers' buffers as they are NOW: epochs whose inputs the caller refills
        between launches (host-drawn shuffles or noise planes) must be checked one by one, as train.train_tadgan does; epochs that
        draw their shuffles inside the captured sequence, or read static buffers, may be queued in any number.
        ``on_epoch(i)``: called behind the i-th repeated epoch's launches (i = 0 for the failed one), stream-ordered in front of the
        next one's -- a caller that wants the weights of one of them (a checkpoint) takes its copy there.
        Returns the status code that was found (0 = nothing happened); raises without ``recover``."""
        c = self.counters.cpu()                  # (synchronises the stream: everything queued so far has run or was skipped)
        code = int(c[4])
        if code == 0:
            self._pending.clear()
            self._pending_overflow = False
            self._steps_checked = (int(c[0]), int(c[2]))
            return 0
        if not recover or not self._pending or self.__dict__.get("_pending_overflow"):
            raise _C.HypadError(f"the resident critic launch gave up (status 0x{code:x}: a bounded wait for a sibling workgroup timed out)")
        st = self._state()
        _C.check(_C.lib.hypad_epoch_restore(ctypes.byref(self.dims), ctypes.byref(st), self.workspace.data_ptr(), self._ws_bytes, _C.stream()),
                 "epoch_restore")
        # the epochs that completed before the failed one, from the optimizer-step counters (restored to the failed epoch's start):
        # critic_x steps and generator steps done since the last clean check
        c = self.counters.cpu()
        done_c, done_g = int(c[0]) - self._steps_checked[0], int(c[2]) - self._steps_checked[1]
        lost = list(self._pending)
        while lost:
            nc_, ng_ = lost[0][0]["n_batches"] * lost[0][0]["n_critics"], lost[0][0]["n_batches"]
            if done_c < nc_ or done_g < ng_:
                break
            done_c, done_g = done_c - nc_, done_g - ng_
            lost.pop(0)
        logging.getLogger("hypad_amd").warning(
            "resident critic launch gave up (status 0x%x): restoring the critics and repeating %d epoch(s) with per-iteration launches", code, len(lost))
        self.epoch_flags = (self.epoch_flags & 0xff) | _C.EPOCH_PER_ITERATION      # (any test-hook bits above bit 7 go)
        self._drop_graphs()
        self._pending.clear()
        self._rerunning = True
        try:
            for i, (call, shuffle_windows) in enumerate(lost):
                if shuffle_windows:
                    self.draw_shuffles(call["row_index"], shuffle_windows)
                self.train_epoch(**call)
                if on_epoch is not None:
                    on_epoch(i)
        finally:
            self._rerunning = False
        again = self.status()
        if again:
            raise _C.HypadError(f"status 0x{again:x} after the per-iteration re-run")
        c = self.counters.cpu()
        self._steps_checked = (int(c[0]), int(c[2]))
        return code

    def _row_index_stride(self, row_index, n_batches, n_critics):
        """0 for one shared (n_passes, n_batches * batch) plane, the plane stride for a (n_signals, n_passes, n_batches * batch) tensor."""
        want = (n_critics + 1, n_batches * self.B)
        if row_index.dim() == 3:
            if tuple(row_index.shape) != (self.n,) + want or not row_index.is_contiguous():
                raise _C.HypadError(f"per-signal row_index must be a contiguous {(self.n,) + want} int32 tensor")
            return want[0] * want[1]
        return 0

    def train_epoch(self, x, row_index, n_batches, n_critics=5, train_mode=True, losses=None, hoist=True, x_row_stride=0, noise=None,
                    workspace_iters=None, flags=None):
        """One epoch of train.py:299-356.  row_index: int32 (n_critics+1, n_batches*batch) on device.
        hoist=False keeps the per-minibatch launch groups for the critic phase (A/B checks).
        x_row_stride=1: x is the scaled series (SignalDataset.window_view), not a window matrix.
        workspace_iters: hand the library only the workspace of that many critic iterations (it then processes the phase in
        slices of that length: what it does on its own for phases longer than 512 iterations) -- tests.
        noise: optional dict of injected planes (hypad_epoch_noise: z_cx, alpha_cx, z_cz, alpha_cz, z_gen, masks_*; float32
        device tensors, iteration-major) replacing the device Philox draws -- parity runs.
        flags: hypad_epoch_io.flags for this call (default: the engine's ``epoch_flags``)."""
        x, stride = self._check_x(x, x_row_stride)
        if hoist:
            self._grow_workspace(_C.lib.hypad_epoch_workspace_bytes(ctypes.byref(self.dims), n_batches, n_critics))
        iters = (2 * n_critics + 1) * n_batches
        if losses is None:
            losses = torch.empty(self.n, iters, 4, dtype=torch.float32, device=self.device)
        nz = None
        if noise:
            unknown = set(noise) - set(self.NOISE_PLANES)
            if unknown:
                raise _C.HypadError(f"unknown noise planes {sorted(unknown)}")
            nz = _C.EpochNoise(*(None if noise.get(k) is None else _C.require_cuda(noise[k], k).data_ptr() for k in self.NOISE_PLANES))
        io = _C.EpochIO(x.data_ptr(), stride, int(x_row_stride), row_index.data_ptr(), n_batches, n_critics, int(train_mode), self.seed,
                        losses.data_ptr(), self.workspace.data_ptr(),
                        (self._ws_bytes if workspace_iters is None else _C.lib.hypad_epoch_workspace_bytes(ctypes.byref(self.dims), int(workspace_iters), 1))
                        if hoist else _C.lib.hypad_train_workspace_bytes(ctypes.byref(self.dims)),
                        ctypes.pointer(nz) if nz is not None else None, int(self.epoch_flags if flags is None else flags),
                        *self._aux_stream_args(), self._row_index_stride(row_index, n_batches, n_critics), *self._enc_table_args(x, x_row_stride))
        st = self._state()
        _C.check(_C.lib.hypad_train_epoch(ctypes.byref(self.dims), ctypes.byref(st), ctypes.byref(io), _C.stream()), "train_epoch")
        if not torch.cuda.is_current_stream_capturing():
            self._last_epoch = dict(x=x, row_index=row_index, n_batches=n_batches, n_critics=n_critics,
                                    train_mode=train_mode, losses=losses, hoist=hoist, x_row_stride=x_row_stride, noise=noise,
                                    workspace_iters=workspace_iters)
            if not self.__dict__.get("_rerunning"):
                self._queued(dict(self._last_epoch))
        return losses
