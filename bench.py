#!/usr/bin/env python3
"""Benchmark of the HypAD / TadGAN training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): one univariate signal per GPU, hyperbolic=True, batch 64, window 100,
1 916 synthetic windows (sine + noise + one jump) resident in HBM as fp32.  One *step* = one
training epoch of train.py:299-356 over those windows: 5 passes of (critic_x_iteration, critic_z_iteration) and one
pass of decoder_iteration over the 29 minibatches = 319 optimizer steps, train-mode dropout and all noise drawn on
the device.  Metric: epoch-windows/s = n_gpus * signals_per_gpu * 29 * 64 * steps / wall time.
With N GPUs every rank trains its own signal(s) (one model per signal, no collective on the data path): weak scaling.

`python bench.py --gpus N` starts its own one-process-per-GPU launcher when none is present.

The JSON line also carries
  roofline         -- the kernel holding the largest share of the epoch, its algorithmic FLOPs per launch over its mean
                      duration measured with HIP events on the launch stream (hypad_profile_iteration);
  roofline_hbm     -- kernel-only GB/s of the row-wise Poincare-ball kernels (pre-allocated outputs, HIP events);
  roofline_scoring -- the same for every scoring kernel (fused test-loop forward, un-roll median, DTW, ...);
  secondary        -- BASELINE.json configs[2]'s per-GPU share: 8 signals per GPU;
  cpu_baseline     -- the CPU oracle (oracle/train_iters.py: the reference's nn.LSTM / autograd / Adam structure)
                      timed on this node's host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S, L, B, N_WINDOWS = 100, 20, 64, 1916
N_BATCHES = N_WINDOWS // B          # 29 (drop_last, main.py:38)
N_CRITICS = 5                       # train.py:301


class Cfg:
    """One training workload: window length, batch, windows per signal, geometry (BASELINE.json configs[0..3])."""

    def __init__(self, name, S=S, B=B, n_windows=N_WINDOWS, hyperbolic=True, data="sine"):
        self.name, self.S, self.L, self.B, self.n_windows, self.hyperbolic, self.data = name, S, L, B, n_windows, hyperbolic, data
        self.nb = n_windows // B
        self.n_it = self.nb * N_CRITICS

    # Algorithmic MACs per window, SURVEY.md §8(d) accounting (1 MAC = 2 FLOP), split by the kernel that does the work.
    def mac(self):
        S_, L_ = self.S, self.L
        f_enc = 2 * (4 * 50) * S_ + 100 * L_                                  # 42 000 at S = 100
        f_dec = L_ * 50 + 2 * 256 * 50 + 2 * 256 * 128 + 128 * S_             # 104 936
        f_cx = L_ * S_ + 3 * L_ * L_ + L_                                     # 3 220
        f_cz = 2 * L_ * L_ + L_                                               # 820
        head = S_ * S_ if self.hyperbolic else 0                              # the Moebius head's F.linear
        return {
            # critic phase as hypad_train_epoch runs it (critic_fused.hip): the frozen generator's forwards of ALL iterations by the
            # record producers (or one precompute launch), then the (critic_x || critic_z) iterations
            "critic_precompute": (f_dec + head) + f_enc,                           # decoder(z) + head, encoder(x)
            "critic_iteration": 10 * (f_cx + f_cz),                                # 3 fwd + 3 backward-data + second-order chain + 3 weight-gradient passes
            "gen": 2 * f_enc + 4 * (f_dec + head) + head + 2 * f_cx + 2 * f_cz,    # fwd + backward-data of decoder_iteration
            "dw_gen": f_enc + 2 * (f_dec + head) + head,                           # its weight-gradient third
        }

    def epoch_flop_per_signal(self):
        m = self.mac()
        return 2.0 * self.B * self.nb * (N_CRITICS * (m["critic_precompute"] + m["critic_iteration"]) + m["gen"] + m["dw_gen"])


CFG1 = Cfg("configs[1]")
MAC_PER_WINDOW = CFG1.mac()
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense


def synth_windows(n, s, seed):
    """SURVEY.md §8d synthetic stand-in for art_daily_jumpsup @interval 600."""
    rng = np.random.default_rng(seed)
    t = np.arange(n + s - 1)
    period = 288.0 if seed == 0 else rng.uniform(200, 400)
    series = np.sin(2 * np.pi * (t + rng.uniform(0, period) * (seed != 0)) / period) + 0.05 * rng.standard_normal(len(t))
    series[n // 2: n // 2 + 40] += 0.8
    series = np.clip(series, -1, 1)
    return series[np.arange(n)[:, None] + np.arange(s)[None, :]]


def build_engine(spg, rank, hyperbolic, device, cfg=None):
    from hypad_amd.engine import Engine
    from hypad_amd.models import tadgan
    cfg = cfg or Cfg("configs[1]" if hyperbolic else "configs[0]", hyperbolic=hyperbolic)
    eng = Engine(cfg.S, L, cfg.B, cfg.hyperbolic, n_signals=spg, device=device, lr=5e-4, seed=1234 + rank)
    xs = []
    for s in range(spg):
        sid = rank * spg + s
        torch.manual_seed(sid)      # random-init weights of the reference architecture (train.py:415-426 order)
        mods = dict(enc=tadgan.Encoder(cfg.S, L), dec=tadgan.Decoder(cfg.S, L, cfg.hyperbolic), cx=tadgan.CriticX(cfg.S, L), cz=tadgan.CriticZ(L))
        for k, m in mods.items():
            eng.load_state_dict(k, m.state_dict(), s)
        if cfg.data == "uniform":   # rows U(-1, 1)
            xs.append(np.random.default_rng(sid).uniform(-1, 1, (cfg.n_windows, cfg.S)))
        else:
            xs.append(synth_windows(cfg.n_windows, cfg.S, sid))
    x = torch.from_numpy(np.stack(xs)).to(device, torch.float32).contiguous()
    return eng, x


def make_step(eng, x, spg, gen, device, graph=True, host_shuffle=False, cfg=CFG1):
    """One timed step = one epoch.  Returns (step, losses).  The DataLoader's shuffles -- a fresh uniform permutation for each of
    the 5 critic passes and the generator pass -- are drawn on the device by the library (hypad_epoch_shuffles: argsort of Philox
    keys in LDS, keyed by the rng tick) as the first node of the captured epoch, so a step is ONE graph replay with no host work
    and no torch kernel; `host_shuffle` (and signals longer than the in-kernel sort's 4 096 windows) draws them with torch instead
    (rand + argsort into the static buffer the epoch reads).  Eager launches: same bits as the replay."""
    nb, Bc, nw = cfg.nb, cfg.B, cfg.n_windows
    host_shuffle = host_shuffle or nw > eng.SHUFFLE_MAX_WINDOWS
    losses = torch.empty(spg, (2 * N_CRITICS + 1) * nb, 4, device=device)
    perm_buf = torch.empty(N_CRITICS + 1, nb * Bc, dtype=torch.int32, device=device)

    def step():
        if host_shuffle:
            perm = torch.rand(N_CRITICS + 1, nw, device=device, generator=gen).argsort(dim=1)[:, : nb * Bc]
            perm_buf.copy_(perm)
        if graph:
            eng.train_epoch_graph(x, perm_buf, nb, N_CRITICS, train_mode=True, losses=losses, shuffle_windows=0 if host_shuffle else nw)
        else:
            if not host_shuffle:
                eng.draw_shuffles(perm_buf, nw)
            eng.train_epoch(x, perm_buf, nb, N_CRITICS, train_mode=True, losses=losses)
    return step, losses


def profile_kernels(eng, x, spg, device, reps=24, cfg=CFG1):
    """Per-kernel launch durations of one epoch at `spg` signals, HIP events on the launch stream (hypad_profile_iteration):
    kind 4 = the critic phase of one configs[1] epoch (145 iterations) exactly as train_epoch launches it: ONE resident launch
    (critic_persistent_kernel; reported per iteration and per launch; with its own record producers, or behind a precompute
    launch) or, where that form cannot run, 145 per-iteration launches (the mean of the steady-state ones); kind 5 = the generator
    step's two kernels as the epoch launches them, the mean over 64 back-to-back launches each (an event pair around ONE 9 us launch
    also times the event path).  Returns per-launch ms, launches per epoch, epoch share and the algorithmic FLOPs of one launch of
    each kernel (SURVEY.md §8d accounting)."""
    persistent = eng.critic_phase_persistent()
    names = {4: ["critic_precompute", "critic_first_or_reinit", "critic_iteration"], 5: ["gen", "dw_gen"]}
    acc = {n: [] for v in names.values() for n in v}
    idx = torch.arange(cfg.B, device=device, dtype=torch.int32)
    for rep in range(reps):
        for kind in (4, 5):
            ms = eng.profile_iteration(kind, x, idx, train_mode=True)
            if rep >= 4:
                for n, v in zip(names[kind], ms):
                    acc[n].append(v)
    kern_ms = {n: float(np.mean(v)) for n, v in acc.items() if n != "critic_first_or_reinit"}
    n_it, nb, m = cfg.n_it, cfg.nb, cfg.mac()
    producers = persistent and eng.critic_phase_producers(n_it)      # the resident launch writes its own records: no precompute launch
    pre_scale = n_it / 145.0                                         # (kind 4 precomputes 145 iterations' records)
    if persistent:
        per_launch = {"critic_persistent_kernel": kern_ms["critic_iteration"] * n_it, "gen_kernel": kern_ms["gen"], "dw_adam_kernel": kern_ms["dw_gen"]}
        launches = {"critic_persistent_kernel": 1, "gen_kernel": nb, "dw_adam_kernel": nb}
        if not producers:
            per_launch["critic_phase_precompute_kernel"] = kern_ms["critic_precompute"] * pre_scale
            launches["critic_phase_precompute_kernel"] = 1
    else:
        per_launch = {"critic_iteration_kernel": kern_ms["critic_iteration"], "critic_phase_precompute_kernel": kern_ms["critic_precompute"] * pre_scale,
                      "gen_kernel": kern_ms["gen"], "dw_adam_kernel": kern_ms["dw_gen"]}
        launches = {"critic_iteration_kernel": n_it + 1, "critic_phase_precompute_kernel": 1, "gen_kernel": nb, "dw_adam_kernel": nb}
    epoch_share = {k: per_launch[k] * launches[k] for k in per_launch}
    # (with producers the resident launch also does the precompute's work: both parts are its algorithmic FLOPs)
    mac = {"critic_persistent_kernel": (m["critic_iteration"] + (m["critic_precompute"] if producers else 0)) * n_it,
           "critic_iteration_kernel": m["critic_iteration"],
           "critic_phase_precompute_kernel": m["critic_precompute"] * n_it, "gen_kernel": m["gen"],
           "dw_adam_kernel": m["dw_gen"]}
    return {"persistent": persistent, "producers": producers, "kern_ms": kern_ms, "per_launch": per_launch, "launches": launches,
            "epoch_share_ms": epoch_share, "dominant": max(epoch_share, key=epoch_share.get),
            "flop_per_launch": {k: 2.0 * mac[k] * cfg.B * spg for k in per_launch}}


EPOCH_FLOP_PER_SIGNAL = CFG1.epoch_flop_per_signal()


def bench_signals_product(n_signals, device, epochs=24):
    """The product loop over many signals -- ``hypad_amd.train.train_signals_resident``: per-signal models, histories, checkpoint
    cadence and files (train.py:428-437, 381-385) -- on the workload of the `signals32` section: wall time between the epochs'
    log lines inside ONE call, without and with checkpoint files."""
    import tempfile
    from types import SimpleNamespace
    from hypad_amd import train as ht
    datasets = [synth_windows(N_WINDOWS, S, s) for s in range(n_signals)]
    out = {"what": "train_signals_resident(%d signals of %d windows): ms between the log lines of consecutive epochs inside one %d-epoch call (median; mean from "
                   "epoch 3 on), save=False and with the reference's checkpoint cadence (every 10th epoch: 4 files per signal, written by a worker thread)"
                   % (n_signals, N_WINDOWS, epochs), "signals": n_signals, "epochs": epochs, "unit": "windows/s (this GPU)"}
    cwd = os.getcwd()
    for save in (False, True):
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)                               # (model_path is relative to the working directory, as in the reference)
            try:
                P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, epochs=epochs, dataset="bench", signal="s",
                                    resume=False, resume_epoch=0)
                stamps = []
                t0 = time.perf_counter()
                ht.train_signals_resident(datasets, P, seed=1, log=lambda s_: stamps.append(time.perf_counter()), save=save)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                os.chdir(cwd)
        w = np.diff(np.asarray(stamps)) * 1e3
        key = "with_checkpoints" if save else "no_files"
        out[key] = {"ms_per_epoch_median": float(np.median(w)), "ms_per_epoch_mean": float(w[2:].mean()), "value": n_signals * (N_WINDOWS // B) * B / float(w[2:].mean()) * 1e3,
                    "call_ms": 1e3 * dt, "setup_and_first_epoch_ms": 1e3 * (stamps[0] - t0), "after_last_epoch_ms": 1e3 * (t0 + dt - stamps[-1])}
    # signals of many different lengths: the planner is left with small groups (one launch sequence per batch count); dealt over lanes
    # (streams that run beside each other) against one after the other
    counts = [B * nb + 7 * (i % 5) for i, nb in enumerate([6, 6, 6, 9, 9, 9, 9, 12, 12, 12, 15, 15, 15, 15, 18, 18, 18, 21, 21, 21, 21, 24, 24, 24, 27, 27, 27, 27, 29, 29, 29, 29])][:n_signals]
    ragged = [synth_windows(n, S, s) for s, n in enumerate(counts)]
    out["ragged"] = {"what": "%d signals of %d different lengths (%d..%d windows): ms per epoch of all, the groups one after the other on one stream / dealt over lanes"
                             % (len(counts), len(set(n // B for n in counts)), min(counts), max(counts)), "groups": [len(m) for _, m in ht.plan_signal_groups(counts, B)[0]]}
    for lanes in (1, None):
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)
            try:
                P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, epochs=epochs, dataset="bench", signal="s",
                                    resume=False, resume_epoch=0, lanes=lanes)
                stamps = []
                ht.train_signals_resident(ragged, P, seed=1, log=lambda s_: stamps.append(time.perf_counter()), save=False)
            finally:
                os.chdir(cwd)
        w = np.diff(np.asarray(stamps)) * 1e3
        out["ragged"]["one_stream" if lanes == 1 else "lanes"] = {"ms_per_epoch_median": float(np.median(w)),
                                                                  "value": sum(n // B * B for n in counts) / float(np.median(w)) * 1e3}
    return out


class SectionAborted(Exception):
    """Raised inside a detail section on the ranks that did NOT fail, at their next guarded collective."""


class RankGuard:
    """The collectives a detail section may use under a process group, made unable to dead-lock the job when ONE rank fails in
    the middle of a section (main.py:32-70's per-signal loop shards without a collective; these sections only agree on times).
    Every guarded collective -- barrier(), max() and the agreement that closes a section -- is the SAME operation: one
    all_reduce(MAX) of [value, failed].  A rank whose section raises posts exactly one more of them with failed = 1 (`run` does
    it) and leaves the section; every other rank meets that message at ITS next guarded collective -- the sequence numbers
    agree, because the failed rank's abort is the collective it would have performed next --, gets SectionAborted and leaves the
    section too, without posting anything.  So all ranks leave at the same collective, the section is reported as skipped, and
    the next section starts with the ranks in step.  Without a process group everything is the identity."""

    def __init__(self, dist=None, device=None, world=1):
        self.dist, self.device, self.world = dist, device, world

    def _reduce(self, value, failed):
        if self.dist is None:
            return value
        t = torch.tensor([value, 1.0 if failed else 0.0], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        v, f = t.tolist()
        if f > 0 and not failed:
            raise SectionAborted("another rank failed in this section")
        return v

    def max(self, ms):
        """A per-rank duration -> the slowest rank's (the job's)."""
        return self._reduce(float(ms), False)

    def barrier(self):
        self._reduce(0.0, False)

    def run(self, fn, *a, **kw):
        """fn(*a, **kw), or {"error": ...} in its place ON EVERY RANK when it raised on any rank: a detail section never costs
        the run its headline, and never leaves a rank behind in a collective."""
        import traceback
        try:
            res = fn(*a, **kw)
            self._reduce(0.0, False)                  # the closing agreement (a rank that failed after the section's last collective is met here)
            return res
        except SectionAborted as e:
            return {"error": "skipped on all ranks: %s" % e}
        except Exception as e:
            traceback.print_exc(file=sys.stderr)
            try:
                self._reduce(0.0, True)               # the abort message: this rank's next -- and last -- collective of the section
            except Exception:                         # (the process group itself is gone: nothing more to agree on)
                traceback.print_exc(file=sys.stderr)
            return {"error": f"{type(e).__name__}: {e}"[:300]}


def bench_signals(spg, rank, device, gen, warmup=5, steps=20, cfg=CFG1, what=None, eager=True, guard=None, world=1):
    """`spg` signals (models) per GPU of workload `cfg`: the epoch replayed as a captured hipGraph (static shuffle buffer),
    `warmup` untimed + `steps` timed epochs; the same epochs launched eagerly (host-bound wherever ~61 launches of CPU enqueue
    exceed the GPU time); per-kernel launch times at this signal count and the chip-level rate: all algorithmic FLOPs of an
    epoch over the epoch's time, against the fp32-MFMA peak."""
    guard = guard or RankGuard()
    eng, x = build_engine(spg, spg * rank, cfg.hyperbolic, device, cfg)
    out = {"workload": what or ("configs[2] per-GPU share: %d signals (%d models) per GPU, otherwise as configs[1]" % (spg, spg)),
           "signals_per_gpu": spg, "window": cfg.S, "batch": cfg.B, "windows_per_signal": cfg.n_windows, "hyperbolic": cfg.hyperbolic,
           "iterations_per_step": (2 * N_CRITICS + 1) * cfg.nb, "steps": steps, "warmup": warmup, "unit": "windows/s (this GPU)",
           "critic_phase_persistent": eng.critic_phase_persistent()}
    modes = ("graph", "eager") if eager else ("graph",)
    runs = {m: [] for m in modes}
    fns = {mode: make_step(eng, x, spg, gen, device, graph=mode == "graph", cfg=cfg) for mode in runs}
    for _ in range(2):                                # the launch modes alternately, twice: every run is reported, the faster one counts
        for mode, (step, losses) in fns.items():      # (a replay measured right after other GPU processes once came out 15 % slower than the
            for _ in range(warmup):                   # eager launches of the same epoch next to it; it does not reproduce in isolation)
                step()
            torch.cuda.synchronize()
            guard.barrier()                           # every rank's timed region starts together; the job's time is the slowest rank's
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            runs[mode].append(guard.max(1e3 * (time.perf_counter() - t0) / steps))
            eng.check_status()
            assert bool(torch.isfinite(losses).all())
    for mode, v in runs.items():
        out[mode + "_ms_per_step"] = min(v)
        out[mode + "_ms_per_step_runs"] = v
    out["launch"] = "hipGraph replay of the captured epoch" + (" (shuffles: torch rand + argsort per epoch -- more than 4 096 windows)"
                                                               if cfg.n_windows > eng.SHUFFLE_MAX_WINDOWS else "")
    out["ms_per_step"] = out["graph_ms_per_step"]
    out["value"] = world * spg * cfg.nb * cfg.B / (1e-3 * out["ms_per_step"])      # all ranks' windows / the slowest rank's time
    out["n_gpus"] = world
    out["unit"] = "windows/s (all %d GPU(s); per-rank times max-reduced)" % world
    prof = profile_kernels(eng, x, spg, device, reps=12, cfg=cfg)
    flop_step = spg * cfg.epoch_flop_per_signal()
    tflops = flop_step / (1e-3 * out["ms_per_step"]) / 1e12
    dom = prof["dominant"]
    out["roofline"] = {"bound": "mfma", "what": "chip level: every kernel's algorithmic FLOPs of one epoch / the epoch's time",
                       "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_F32_MFMA_TFLOPS,
                       "flop_per_step": flop_step, "kernel_ms": prof["per_launch"], "launches_per_step": prof["launches"],
                       "epoch_share_ms": prof["epoch_share_ms"], "dominant": dom,
                       "dominant_achieved": prof["flop_per_launch"][dom] / (prof["per_launch"][dom] * 1e-3) / 1e12,
                       "dominant_frac": prof["flop_per_launch"][dom] / (prof["per_launch"][dom] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                       "critic_phase": ("resident launch" + (" with its own record producers" if prof["producers"] else " behind a precompute launch"))
                                       if prof["persistent"] else "one launch per iteration"}
    del eng, x
    return out


def bench_signals_sharded(device, world, rank, per_gpu=8, epochs=24, warm=3):
    """BASELINE.json configs[2] through the PRODUCT path under a process group: ``per_gpu * world`` signals of 1 916 windows through
    ``hypad_amd.train.train_signals_resident`` -- plan_signal_groups shards the list over the ranks, every rank trains its own signals (one
    model per signal, no collective on the data path), the per-signal metrics are gathered at the end (all_gather_object through RCCL).
    All ranks call this.  dist.barrier() in front of and behind the call; `call_ms` = the slowest rank's; the steady-state epoch time =
    the spacing of a rank's epoch log lines from epoch `warm` on, max over ranks; value = all ranks' windows / that.  At one GPU a
    world-size-1 nccl group is created for the section, as bench_scoring_sharded does."""
    import tempfile
    from types import SimpleNamespace
    import torch.distributed as dist
    from hypad_amd import train as ht
    own_group = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
        own_group = True
    try:
        n = per_gpu * world
        datasets = [synth_windows(N_WINDOWS, S, s) for s in range(n)]                  # (every rank holds the list; it uploads only its own signals)
        names = ["sig%03d" % i for i in range(n)]
        plan, _ = ht.plan_signal_groups([N_WINDOWS] * n, B, world, rank)
        mine = sum(len(m) for _, m in plan)
        P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=True, epochs=epochs, dataset="bench", signal="s",
                            resume=False, resume_epoch=0)
        stamps = []
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)
            try:
                dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = ht.train_signals_resident(datasets, P, names=names, seed=1, log=lambda s_: stamps.append(time.perf_counter()), save=False)
                torch.cuda.synchronize(); dist.barrier()
                call_ms = 1e3 * (time.perf_counter() - t0)
            finally:
                os.chdir(cwd)
        assert sorted(res) == names and all(len(res[k]["history"]["dec"]) == epochs for k in names)      # every rank holds every signal's history
        assert all(np.isfinite(res[k]["history"]["dec"]).all() for k in names)
        w = np.diff(np.asarray(stamps))[warm - 1:] * 1e3
        guard = RankGuard(dist, device, dist.get_world_size())
        steady = guard.max(float(w.mean()))
        setup = guard.max(1e3 * (stamps[0] - t0))
        call_ms = guard.max(call_ms)
        return {"what": "configs[2] through train_signals_resident under a process group: %d signals of %d windows (%d per GPU), hyperbolic, %d epochs; "
                        "value = all ranks' windows per epoch / the slowest rank's steady-state epoch time (epochs %d.. of one call)" % (n, N_WINDOWS, per_gpu, epochs, warm),
                "rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(), "signals": n, "signals_this_rank": mine,
                "ranks_of_signals": sorted({res[k]["rank"] for k in names}), "epochs": epochs,
                "ms_per_epoch": steady, "value": n * N_BATCHES * B / steady * 1e3, "unit": "windows/s (all ranks)",
                "call_ms": call_ms, "setup_and_first_epoch_ms": setup, "call_value": n * N_BATCHES * B * epochs / call_ms * 1e3,
                "repairs": int(sum(res[k]["history"].get("repairs", 0) for k in names))}
    finally:
        if own_group:
            dist.destroy_process_group()


def bench_call_level(device, epochs=40):
    """What a user's call costs, set-up included (the headline is steady state): ``hypad_amd.train.train(train_loader, params, config_path)``
    -- main.py:53 -- at configs[1] with the reference's 40 epochs (configs/univariate.yaml:3): models built, engine and graph set up, 40
    epochs, final checkpoint files; wall time of the whole call and of its parts."""
    import contextlib
    import io
    import tempfile
    from types import SimpleNamespace
    from torch.utils.data import DataLoader
    from hypad_amd import train as ht
    ds = _synthetic_signal_dataset()
    out = {"what": "train.train(DataLoader(SignalDataset, batch 64, shuffle, drop_last), params, None): configs[1], %d epochs (configs/univariate.yaml), "
                   "checkpoints at the reference's cadence; call_ms = the whole call (second of two calls in this process; first_call_ms = the first: "
                   "library load, stream probe, allocator warm-up)" % epochs, "epochs": epochs}
    cwd = os.getcwd()
    calls = []
    for rep in range(2):
        loader = DataLoader(ds, batch_size=B, drop_last=True, shuffle=True, num_workers=0)
        P = SimpleNamespace(batch_size=B, signal_shape=S, lr=5e-4, hyperbolic=True, epochs=epochs, dataset="bench", signal="call%d" % rep, resume=False, resume_epoch=0)
        torch.manual_seed(rep); np.random.seed(rep)
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)
            try:
                stamps = []

                class Stamper(io.StringIO):              # the epochs' print lines (train.py:367), stamped as they are written
                    def write(self, text):
                        if "training done in epoch" in text:
                            stamps.append(time.perf_counter())
                        return len(text)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(Stamper()):
                    ht.train(loader, P, None)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                calls.append(1e3 * (t1 - t0))
                parts = {"setup_and_first_epoch_ms": 1e3 * (stamps[0] - t0), "ms_per_epoch": 1e3 * float(np.diff(stamps)[1:].mean()),
                         "after_last_epoch_ms": 1e3 * (t1 - stamps[-1])}
                parts["setup_ms"] = parts["setup_and_first_epoch_ms"] - parts["ms_per_epoch"]
            finally:
                os.chdir(cwd)
    out["first_call_ms"], out["call_ms"] = calls
    out.update(parts)                                    # (of the second call)
    out["value"] = epochs * N_BATCHES * B / (calls[1] * 1e-3)
    out["unit"] = "windows/s over the whole call"
    return out


def _synthetic_signal_dataset(test=False):
    """configs[1]'s data shape through the product's own dataset class: 2 016 samples at 600 s (art_daily_jumpsup aggregated at
    interval 600) -> 1 916 windows of 100, float64, scaled to [-1, 1] (utils/dataloader.py:61-232)."""
    import pandas as pd
    from hypad_amd.utils.dataloader import SignalDataset
    n = N_WINDOWS + S
    rng = np.random.default_rng(0)
    t = np.arange(n)
    v = np.sin(2 * np.pi * t / 288.0) + 0.05 * rng.standard_normal(n)
    v[n // 2: n // 2 + 40] += 0.8
    df = pd.DataFrame({"timestamp": 1_400_000_000 + 600 * t, "value": v})
    return SignalDataset(df, interval=600, windows_size=S, test=test)


def bench_drop_in(hyperbolic, device, epochs=(3, 30)):
    """The reference's call surface at speed: ``hypad_amd.train.train_tadgan(train_loader, encoder, decoder, critic_x, critic_z,
    n_epochs, params, path)`` -- the reference's signature (train.py:252), its epoch schedule, its host random numbers (z from
    NumPy's global generator, alpha and the loader's seeds from torch's CPU generator, in the reference's order) -- where every
    epoch runs as one captured hypad_train_epoch fed from host-staged planes (hypad_amd/epoch_feed.py).  Timed: steady-state
    windows/s inside ONE call of `epochs[1]` + 2 epochs (engine construction, graph capture and the first two epochs excluded; prints
    and loss read-back included).
      host_samples   -- train_loader = a list of 29 host (64, 100, 1) float64 minibatches (what round 3 timed call by call)
      dataloader     -- torch DataLoader(SignalDataset, batch 64, shuffle, drop_last), as main.py:33-39 builds it: the index path
      dataloader_staged -- the same loader with every batch really fetched, collated and staged (params.stage_samples)
      call_by_call   -- the loop over critic_x_iteration / critic_z_iteration / decoder_iteration (params.per_iteration)"""
    import contextlib
    import io
    from types import SimpleNamespace
    from torch.utils.data import DataLoader
    from hypad_amd import anomaly_detection as had
    from hypad_amd import train as ht
    from hypad_amd.models import tadgan

    def run(loader, n_epochs, **kw):
        P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=hyperbolic, resume=False, resume_epoch=0, **kw)
        torch.manual_seed(0); np.random.seed(0)
        mods = [m.to(device).train() for m in (tadgan.Encoder(S, L), tadgan.Decoder(S, L, hyperbolic), tadgan.CriticX(S, L), tadgan.CriticZ(L))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            hist = ht.train_tadgan(loader, *mods, n_epochs=n_epochs, params=P, path="/tmp")
        torch.cuda.synchronize()
        assert np.isfinite(hist.dec).all() and len(hist.dec) == n_epochs
        return time.perf_counter() - t0, hist, mods

    def rate(loader, **kw):
        """Steady state of ONE call: train_tadgan stamps the moment each epoch's losses reach the host (hist.wall) -- the mean spacing
        of those stamps from the third epoch on.  Best of two calls."""
        best = float("inf")
        n_ep = epochs[0] if kw.get("per_iteration") else epochs[1]
        for _ in range(2):
            _, hist, mods = run(loader, n_ep + 2, **kw)
            best = min(best, (hist.wall[-1] - hist.wall[1]) / n_ep)
        return {"value": N_BATCHES * B / best, "ms_per_epoch": 1e3 * best, "us_per_iteration": 1e6 * best / ((2 * N_CRITICS + 1) * N_BATCHES)}, mods

    data = torch.from_numpy(synth_windows(N_WINDOWS, S, 0)[: N_BATCHES * B, :, None])      # float64, like SignalDataset (dataloader.py:227-232)
    host_list = [data[b * B:(b + 1) * B] for b in range(N_BATCHES)]
    ds = _synthetic_signal_dataset()
    loader = DataLoader(ds, batch_size=B, drop_last=True, shuffle=True, num_workers=0)
    out = {"what": "hypad_amd.train.train_tadgan(train_loader, ...) -- reference signature, schedule and host RNG order; one captured "
                   "hypad_train_epoch per epoch (epoch_feed.py); steady-state windows/s inside one call of %d epochs (the first two excluded)" % (epochs[1] + 2),
           "unit": "windows/s", "host_cores": os.cpu_count()}
    out["host_samples"], mods = rate(host_list)
    out["dataloader"], _ = rate(loader)
    out["dataloader_staged"], _ = rate(loader, stage_samples=True)
    out["call_by_call"], _ = rate(host_list, per_iteration=True)
    out["value"] = out["host_samples"]["value"]
    out["ms_per_epoch"] = out["host_samples"]["ms_per_epoch"]
    # the same call at configs[3]'s shape (window 150, batch 256, 20 480 windows U(-1, 1)): an epoch draws 4.5 M latent values from
    # NumPy's generator and 17.4 M interpolation weights from torch's -- the host random numbers are what bounds it
    try:
        Sm, Bm, Nm = 150, 256, 20480
        mdata = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (Nm, Sm, 1)))
        mloader = DataLoader(mdata, batch_size=Bm, drop_last=True, shuffle=True, num_workers=0)
        Pm = SimpleNamespace(batch_size=Bm, signal_shape=Sm, latent_space_dim=L, lr=5e-4, hyperbolic=True, resume=False, resume_epoch=0)
        torch.manual_seed(0); np.random.seed(0)
        mm = [m.to(device).train() for m in (tadgan.Encoder(Sm, L), tadgan.Decoder(Sm, L, True), tadgan.CriticX(Sm, L), tadgan.CriticZ(L))]
        with contextlib.redirect_stdout(io.StringIO()):
            mh = ht.train_tadgan(mloader, *mm, n_epochs=14, params=Pm, path="/tmp")
        w = np.diff(np.asarray(mh.wall))[1:]
        out["multivariate"] = {"what": "train_tadgan over a DataLoader at configs[3]'s shape (window 150, batch 256, 20 480 windows, hyperbolic): ms between the "
                                       "epochs' losses reaching the host inside one 14-epoch call (mean from epoch 3 on; median)",
                               "ms_per_epoch": 1e3 * float(w.mean()), "ms_per_epoch_median": 1e3 * float(np.median(w)),
                               "value": (Nm // Bm) * Bm / float(w.mean()), "unit": "windows/s"}
        del mm, mdata, mloader
    except Exception as e:          # (reported, not hidden)
        out["multivariate"] = f"{type(e).__name__}: {e}"[:300]
    # ---- the test loop (anomaly_detection.py:67-113) through a batch-64 DataLoader, as main.py:40-46 builds it
    enc, dec, cx = mods[0], mods[1], mods[2]
    tds = _synthetic_signal_dataset(test=True)
    tloader = DataLoader(tds, batch_size=B, drop_last=False, shuffle=False, num_workers=0)
    P = SimpleNamespace(batch_size=B, signal_shape=S, hyperbolic=hyperbolic)

    def timed(fn, reps=5):                        # median of single calls (each returns host arrays or is drained: synchronous)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]
    n = len(tds)
    t_call = timed(lambda: had.test_tadgan(tloader, enc, dec, cx, path="", signal_shape=S, params=P))
    t_per = timed(lambda: had.score_batches_per_batch(tloader, enc, dec, cx, S), reps=2)
    big = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (125_000, S, 1)))
    bloader = DataLoader(big, batch_size=B, drop_last=False, shuffle=False, num_workers=0)
    t_big = timed(lambda: had.test_tadgan(bloader, enc, dec, cx, path="", signal_shape=S, params=P))
    import pandas as pd
    from hypad_amd.utils.dataloader import SignalDataset
    tt = np.arange(125_000 + S)
    sds = SignalDataset(pd.DataFrame({"timestamp": 1_400_000_000 + 600 * tt, "value": np.sin(2 * np.pi * tt / 288.0) + 0.05 * np.random.default_rng(2).standard_normal(len(tt))}),
                        interval=600, windows_size=S, test=True)
    sloader = DataLoader(sds, batch_size=B, drop_last=False, shuffle=False, num_workers=0)
    t_sig = timed(lambda: had.test_tadgan(sloader, enc, dec, cx, path="", signal_shape=S, params=P))
    out["scoring"] = {"what": "anomaly_detection.test_tadgan(test_loader, ...) through a batch-64 DataLoader (anomaly_detection.py:67-113): "
                              "loader -> ONE fused forward -> results back as NumPy (D2H included); per_batch = one pack + forward launch per "
                              "loader batch with every batch fetched and collated (the round-3 form)",
                      "windows": n, "value": n / t_call, "unit": "windows/s", "ms_per_call": 1e3 * t_call,
                      "per_batch_value": n / t_per, "per_batch_ms": 1e3 * t_per,
                      "windows_125000": {"value": 125_000 / t_big, "ms_per_call": 1e3 * t_big,
                                         "what": "the same call over a DataLoader of 125 000 float64 windows in host memory (one GPU's share of configs[4]): "
                                                 "float64 -> float32 + H2D of the window matrix and the NumPy results' D2H included"},
                      "signal_125000": {"value": len(sds) / t_sig, "ms_per_call": 1e3 * t_sig, "windows": len(sds),
                                        "what": "the same call over DataLoader(SignalDataset(test=True)) of one 125 100-sample signal: the ordered windows are "
                                                "read from the scaled series on the device (0.5 MB up instead of the 100 MB window matrix)"}}
    return out


def _cpu_rate(threads, hyperbolic, budget, max_batches):
    """(windows/s, minibatches, seconds) of the oracle's epoch body at `threads` intra-op threads."""
    from types import SimpleNamespace
    from oracle import tadgan as ot
    from oracle import train_iters as oi
    P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=hyperbolic)
    data = torch.from_numpy(synth_windows(4 * B, S, 0)[:, :, None])
    batches = [data[i * B:(i + 1) * B] for i in range(4)]
    torch.set_num_threads(threads)
    enc, dec, cx, cz = ot.build_models(S, L, hyperbolic, seed=0)
    opt = oi.make_optimizers(enc, dec, cx, cz, P)
    np.random.seed(0)
    oi.train_epoch(batches[:1], enc, dec, cx, cz, opt, P)                  # warm-up: one minibatch's 11 iterations
    t0 = time.perf_counter()
    nb = 0
    while nb < 1 or (time.perf_counter() - t0 < budget and nb < max_batches):
        oi.train_epoch(batches[nb % 4: nb % 4 + 1], enc, dec, cx, cz, opt, P)
        nb += 1
    dt = time.perf_counter() - t0
    return nb * B / dt, nb, dt


def cpu_baseline(hyperbolic, budget_s=24.0):
    """The oracle's epoch (same iteration mix) on a bounded number of minibatches: at 1 thread and at a modest intra-op pool
    (the baseline is the better of the two), and ONCE at every core of the host (SURVEY.md §8d asks for it; recorded even when
    slower -- these layer sizes, <= 256 x 128, do not scale past a few cores and a 256-thread pool mostly waits on itself)."""
    ncores = os.cpu_count() or 1
    run = lambda threads, budget, max_batches: _cpu_rate(threads, hyperbolic, budget, max_batches)
    best = None
    for threads in sorted({1, min(ncores, 8)}):
        rate, nb, dt = run(threads, budget_s / 2, 4096)
        if best is None or rate > best["value"]:
            best = dict(value=rate, cores=threads, sample=f"{nb} minibatches x (5 critic_x + 5 critic_z + 1 decoder) iterations, "
                                                               f"B={B}, window={S}, train-mode dropout, {dt:.1f} s")
    # every core, once, in a child process with a deadline: on a 256-core host one minibatch took 123 s (0.5 windows/s) -- the
    # intra-op pool synchronises 256 threads around each of ~30 000 tiny ops
    all_cores = None
    if ncores > 8:
        import subprocess
        code = ("import sys, json; sys.path.insert(0, %r); import bench; "
                "r = bench._cpu_rate(%d, %r, 2.0, 4); print(json.dumps(r))" % (ROOT, ncores, bool(hyperbolic)))
        try:
            res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
            rate, nb, dt = json.loads(res.stdout.strip().splitlines()[-1])
            all_cores = dict(value=rate, cores=ncores, sample=f"{nb} minibatch(es), {dt:.1f} s")
            if rate > best["value"]:
                best = dict(value=rate, cores=ncores, sample=all_cores["sample"])
        except subprocess.TimeoutExpired:
            all_cores = dict(value=None, cores=ncores, sample="warm-up + one minibatch (22 iterations) did not finish within 25 s "
                                                              "(< %.1f windows/s): slower than 1 thread by orders of magnitude" % (2 * B / 25.0))
        except (ValueError, IndexError, OSError):
            all_cores = dict(value=None, cores=ncores, sample="child process failed")
    torch.set_num_threads(ncores)
    best.update(unit="windows/s", kind="port", host_cores=ncores, all_cores=all_cores)
    return best


def cpu_scoring_baseline(n):
    """The same scoring pass on the host: the oracle's networks (torch CPU, the reference's module structure) for the test-loop
    forward and oracle/scoring.py (NumPy / SciPy / pandas restatement of utils/anomaly_detection_utils.py) for the numerics, on a
    bounded sample of `n` windows.  The KDE critic smoothing -- scipy.stats.gaussian_kde once per timestep -- dominates."""
    from oracle import scoring as osc
    from oracle import tadgan as ot
    ncores = os.cpu_count() or 1
    threads = min(ncores, 8)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    enc, dec, cx = ot.Encoder(S, L).eval(), ot.Decoder(S, L, True).eval(), ot.CriticX(S, L).eval()
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (n, S))
    t0 = time.perf_counter()
    with torch.no_grad():
        xt = torch.from_numpy(x)
        hyper, eucl = dec(enc(xt.view(1, -1, S)))
        critic = cx(xt.view(1, -1, S)).reshape(-1).numpy()
        real = dec.hyperbolic_linear(xt.float()).numpy()
        eucl = eucl.reshape(-1, S).numpy()
        hyper = hyper.reshape(-1, S).numpy()
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    true = osc.unroll_true(x)
    pred, _ = osc.unroll_predictions(eucl, False)
    w = min(max(n // 100, 2), 200)
    e1 = osc.zscore_clip(osc.rolling_mean_centered(osc.point_error(true, pred), w))
    e2 = osc.zscore_clip(osc.rolling_mean_centered(osc.dtw_error(true, pred.astype(np.float64), 10), w))
    t_num = time.perf_counter() - t0
    t0 = time.perf_counter()
    cs = osc.final_critic_scores(critic, n, S)
    t_kde = time.perf_counter() - t0
    assert np.isfinite(e1).all() and np.isfinite(e2).all() and np.isfinite(cs).all() and np.isfinite(hyper).all() and np.isfinite(real).all()
    torch.set_num_threads(ncores)
    tot = t_fwd + t_num + t_kde
    return {"value": n / tot, "unit": "windows/s", "cores": threads, "host_cores": ncores, "kind": "port",
            "sample": f"{n} windows of 100: forward {t_fwd:.2f} s (torch CPU, {threads} threads), numerics {t_num:.2f} s, KDE critic smoothing "
                      f"{t_kde:.2f} s (NumPy / SciPy, 1 thread)",
            "without_kde_value": n / (t_fwd + t_num)}


def _event_ms_median(fn, rounds=21, per_round=3):
    """Median over `rounds` of the mean duration of `per_round` back-to-back calls (HIP events on the launch stream): one slow round
    (another process on the box, a clock ramp) does not move it."""
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(per_round):
            fn()
        b.record()
        b.synchronize()
        out.append(a.elapsed_time(b) / per_round)
    return float(np.median(out))


HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec; 6 290 measured for a float4 copy


def bench_scoring(device, n=125_000, reps=5, cpu_sample=0, smooth=200, kernels=True):
    """BASELINE.json's second metric, anomaly-score windows/s, on this GPU's share of configs[4] (10^6 windows over 8 GPUs): the
    test-loop forward with the hyperbolic row distance (anomaly_detection.py:67-113), then un-roll median + point and DTW errors +
    rolling mean + z-score (utils/anomaly_detection_utils.py:866-962, 516-524).  Returns (scoring, roofline_hbm, roofline_scoring):
    kernel-only rates -- every output pre-allocated, HIP events around back-to-back launches through the C ABI -- of the row-wise
    Poincare-ball kernels (algorithmic bytes per row) and of each scoring kernel.  ``smooth``: window of the
    reconstruction errors' rolling mean (the reference smooths with 1 % of the windows: anomaly_detection_utils.py:459-460);
    ``kernels=False``: only the `scoring` section (the 10^6-window run)."""
    from hypad_amd import _C
    from hypad_amd.hyperspace import gmath
    from hypad_amd.models import tadgan
    from hypad_amd.utils import anomaly_detection_utils as adu
    torch.manual_seed(0)
    enc, dec, cx = tadgan.Encoder(S, L).to(device).eval(), tadgan.Decoder(S, L, True).to(device).eval(), tadgan.CriticX(S, L).to(device).eval()
    g = torch.Generator(device=device).manual_seed(3)
    x = (torch.rand(n, S, device=device, generator=g) * 2 - 1).contiguous()
    new = lambda *shape, dtype=torch.float32: torch.empty(*shape, device=device, dtype=dtype)
    hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)
    st = _C.stream

    def timed(fn):
        fn()
        best = float("inf")
        for _ in range(3):                  # (the fastest of three rounds of `reps` calls: an allocator hiccup must not count)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
        return best

    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=device)

    def forward():
        _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, _C.ptr(hyper),
                                                   _C.ptr(eucl), _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist), n, S, L, 1, ws.data_ptr(),
                                                   ws_bytes, st()), "score_forward_packed")

    def numerics():
        true = adu.unroll_true(x)
        pred, _ = adu.unroll_predictions(eucl, False)
        e1 = adu.rolling_mean(adu._point_wise_error(true, pred), smooth)
        e2 = adu.rolling_mean(adu._dtw_error(true, pred, 10), smooth)
        return adu.zscore_clip(e1), adu.zscore_clip(e2)

    def critic_smoothing():         # final_critic_scores (utils/anomaly_detection_utils.py:365-404): KDE mode per timestep, trimmed z-score, rolling mean
        return adu._compute_critic_score(adu.kde_modes(critic, S), n // 100)

    def whole_pass():               # as score_anomalies queues it: the critic smoothing on a side stream beside the reconstruction numerics
        forward()
        return adu.concurrently(numerics, critic_smoothing)

    t_fwd, t_num, t_kde, t_pass = timed(forward), timed(numerics), timed(critic_smoothing), timed(whole_pass)
    scoring = {"windows": n, "value": n / t_pass, "unit": "windows/s", "pass_ms": 1e3 * t_pass, "stage_sum_value": n / (t_fwd + t_num + t_kde),
               "forward_windows_per_s": n / t_fwd, "numerics_windows_per_s": n / t_num, "critic_smoothing_windows_per_s": n / t_kde,
               "without_kde_value": n / (t_fwd + t_num),
               "forward_ms": 1e3 * t_fwd, "numerics_ms": 1e3 * t_num, "critic_smoothing_ms": 1e3 * t_kde, "smoothing_window": smooth,
               "what": "value = one whole pass: test-loop forward, then the reconstruction numerics (un-roll median, point + DTW(11) errors, rolling "
                       "mean(smoothing_window), z-score) with the KDE critic smoothing score_anomalies runs (utils/anomaly_detection_utils.py:470-506) on a "
                       "side stream beside them (anomaly_detection_utils.concurrently, as hypad_amd's score_anomalies queues them); host wall clock around "
                       "the wrappers.  stage_sum_value = the three stages timed one by one and added (the `value` of rounds 3-4a); "
                       "without_kde_value = forward + numerics only (the figure of rounds 1-2)"}
    # the whole pass as ONE replayed hipGraph (parallel.replay_scorer: the pass is a fixed launch sequence -- nothing goes through the host):
    # reported next to `value`
    try:
        from hypad_amd import parallel as par

        def whole():
            forward()
            num, crit = adu.concurrently(numerics, critic_smoothing)
            return num + (crit,)
        rep = lambda: par.replay_scorer(whole, x, enc.arena(), dec.arena(), cx.arena(), key=("bench_scoring", n, smooth))
        rep()
        scoring["graph_replay_value"] = n / timed(rep)
    except Exception as e:
        scoring["graph_replay_value"] = f"{type(e).__name__}: {e}"[:200]

    # the same pass when the window matrix arrives in (pinned) host memory: one 50 MB H2D copy per pass in front of the forward --
    # reported next to `value`, never as it (inputs resident in HBM is the contract's figure); the series view of the scorers
    # (x_row_stride = 1: windows n = series[n : n + S]) moves 0.5 MB instead
    x_host = x.cpu().pin_memory()
    series_host = torch.empty(n + S - 1, dtype=torch.float32).pin_memory()
    x_dev, series_dev = torch.empty_like(x), torch.empty(n + S - 1, dtype=torch.float32, device=device)
    t_h2d = timed(lambda: x_dev.copy_(x_host, non_blocking=True))
    t_h2d_series = timed(lambda: series_dev.copy_(series_host, non_blocking=True))
    scoring["pcie_inclusive"] = {"window_matrix_value": n / (t_pass + t_h2d), "h2d_ms": 1e3 * t_h2d,
                                 "h2d_GB_per_s": x.numel() * 4 / t_h2d / 1e9,
                                 "series_view_value": n / (t_pass + t_h2d_series), "series_h2d_ms": 1e3 * t_h2d_series,
                                 "what": "value with the input crossing PCIe from pinned host memory in front of every pass: the (N, S) fp32 window "
                                         "matrix, or the scaled series the windows are views of"}
    if cpu_sample:
        scoring["cpu_baseline"] = cpu_scoring_baseline(cpu_sample)
    if not kernels:
        return scoring, None, None

    # ---- each scoring kernel on its own: pre-allocated outputs, HIP events
    T = n + S - 1
    true64, pred32 = adu.unroll_true(x), new(T)
    err64, sm64, z64, modes64 = new(T, dtype=torch.float64), new(T, dtype=torch.float64), new(T, dtype=torch.float64), new(T, dtype=torch.float64)
    stats = torch.empty(_C.STATS_WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    roll_bytes = _C.lib.hypad_rolling_workspace_bytes(T)
    roll_ws = torch.empty(roll_bytes, dtype=torch.uint8, device=device)
    x64 = x.to(torch.float64)
    C = _C.lib
    kern = {
        "score_forward_packed_kernel": (forward, "mfma", 340312.0 * n, None),
        "unroll_median_kernel": (lambda: C.hypad_unroll_median(_C.ptr(eucl), _C.ptr(pred32), None, n, S, st()), "hbm", None, (4 * S + 4) * n),
        "unroll_true": (lambda: C.hypad_unroll_true(_C.ptr(x64), _C.ptr(true64), n, S, st()), "hbm", None, 16 * T),
        "point_error": (lambda: C.hypad_point_error(_C.ptr(true64), _C.ptr(pred32), _C.ptr(err64), T, st()), "hbm", None, 20 * T),
        "dtw_error_kernel<11>": (lambda: C.hypad_dtw_error(_C.ptr(true64), _C.ptr(pred32), _C.ptr(err64), T, 10, st()), "valu", 121.0 * T, 20 * T),
        "rolling_mean(200)": (lambda: C.hypad_rolling_mean(_C.ptr(err64), None, _C.ptr(sm64), T, 200, 0, roll_ws.data_ptr(), roll_bytes, st()), "hbm", None, 16 * T),
        "rolling_mean(1250) of |true - pred|": (lambda: C.hypad_rolling_mean(_C.ptr(true64), _C.ptr(pred32), _C.ptr(sm64), T, n // 100, 0, roll_ws.data_ptr(), roll_bytes, st()), "hbm", None, 20 * T),
        "zscore_clip": (lambda: C.hypad_zscore_clip(_C.ptr(sm64), _C.ptr(z64), T, stats.data_ptr(), _C.STATS_WORKSPACE_BYTES, st()), "hbm", None, 24 * T),
        "kde_mode_kernel": (lambda: C.hypad_kde_mode(_C.ptr(critic), _C.ptr(modes64), n, S, st()), "valu", float(S) * S * T, 4 * n + 8 * T),
        # np.quantile 25 / 75 % by radix selection (3 histogram passes + 1 compaction pass over the keys, then one workgroup) + quantile-trimmed
        # z-score, nothing through the host
        "critic_score (device quantiles + trimmed z-score)": (lambda: C.hypad_critic_score(_C.ptr(modes64), _C.ptr(z64), T, cs_ws.data_ptr(), cs_bytes, st()),
                                                               "hbm", None, (4 * 8 + 24) * T),
    }
    cs_bytes = C.hypad_critic_score_workspace_bytes()
    cs_ws = torch.empty(cs_bytes, dtype=torch.uint8, device=device)
    roofline_scoring = {}
    for name, (fn, bound, work, nbytes) in kern.items():
        ms = _event_ms_median(fn)           # (median of 21 rounds of 3 launches, as roofline_hbm / roofline_lstm: a mean of five calls moved by 10 % run to run)
        ent = {"bound": bound, "ms": ms, "windows_per_s": n / (ms * 1e-3)}
        if bound == "mfma":
            ent.update(achieved=work / (ms * 1e-3) / 1e12, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s")
            ent["frac"] = ent["achieved"] / ent["peak"]
        else:
            ent.update(achieved=nbytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s", algorithmic_bytes=nbytes)
            ent["frac"] = ent["achieved"] / ent["peak"]
            if work:
                ent["work_per_s"] = work / (ms * 1e-3)          # DTW cell updates/s; KDE kernel evaluations/s
        roofline_scoring[name] = ent

    # ---- row-wise ball kernels on 2 * 10^6 rows (0.8 GB per operand: well past the 256 MB Infinity Cache, so the rate is HBM's)
    m = 2_000_000
    xb = (torch.rand(m, S, device=device, generator=g) * 2 - 1).contiguous()
    ball = gmath.expmap0(0.03 * torch.randn(m, S, device=device, generator=g))
    other = gmath.expmap0(0.03 * torch.randn(m, S, device=device, generator=g))
    out, dv = new(m, S), new(m)
    ops = {"expmap0 (unary_rows)": (lambda: C.hypad_expmap0_fwd(_C.ptr(xb), _C.ptr(out), m, S, st()), 8 * S),
           "logmap0 (unary_rows)": (lambda: C.hypad_logmap0_fwd(_C.ptr(ball), _C.ptr(out), m, S, st()), 8 * S),
           "project (unary_rows)": (lambda: C.hypad_project_fwd(_C.ptr(xb), _C.ptr(out), m, S, st()), 8 * S),
           "mobius_add (mobius_add_rows)": (lambda: C.hypad_mobius_add_fwd(_C.ptr(ball), _C.ptr(other), _C.ptr(out), m, S, m, st()), 12 * S),
           "poincare_rowdist (rowdist_rows)": (lambda: C.hypad_poincare_rowdist_fwd(_C.ptr(ball), _C.ptr(other), _C.ptr(dv), m, S, st()), 8 * S + 4)}
    roofline_hbm = {"rows": m, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "kernels": {}}
    roofline_hbm["timing"] = "median of 21 rounds of 3 back-to-back launches, HIP events"
    for name, (fn, bpr) in ops.items():
        ms = _event_ms_median(fn)
        gb = m * bpr / (ms * 1e-3) / 1e9
        roofline_hbm["kernels"][name] = {"bytes_per_row": bpr, "ms": ms, "achieved": gb, "frac": gb / HBM_PEAK_GBPS}
    return scoring, roofline_hbm, roofline_scoring


def bench_lstm_layers(device, rows=200_000):
    """The reference's bidirectional LSTM layers as stand-alone kernels (hypad_lstm_bidir_fwd: T = 1, h0 = c0 = 0, models/tadgan.py:15-27,
    35-38,58-62 -- the weights-stationary LDS forms of ops_dense.hip) at 200 000 rows: microseconds per launch, the rate of the gate
    products the algorithm needs (2 x 3 x H x K MAC per row and direction) against the fp32-MFMA peak, and the bytes the layer must
    move (x once, h, the saved gates) against HBM; and the general-T layer (hypad_lstm_bidir_seq_fwd: W_hh in LDS, one barrier per
    step) in microseconds per time step."""
    from hypad_amd import _C
    from hypad_amd import autograd as hag
    out = {"rows": rows, "unit": "us per launch", "timing": "median of 21 rounds of 3 back-to-back launches, HIP events", "layers": {}, "sequence": {}}
    torch.manual_seed(0)
    for in_dim, hidden in ((100, 50), (128, 64), (50, 64)):
        lstm = torch.nn.LSTM(input_size=in_dim, hidden_size=hidden, num_layers=1, bidirectional=True).to(device)
        x = torch.randn(rows, in_dim, device=device)
        o, gates = torch.empty(rows, 2 * hidden, device=device), torch.empty(rows, 8 * hidden, device=device)
        ps = [_C.ptr(getattr(lstm, n).detach().contiguous()) for n in ("weight_ih_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse",
                                                                        "bias_ih_l0_reverse", "bias_hh_l0_reverse")]
        for gs, tag in ((gates, "gates saved"), (None, "forward only")):
            fn = lambda: _C.check(_C.lib.hypad_lstm_bidir_fwd(_C.ptr(x), *ps, _C.ptr(o), _C.ptr(gs) if gs is not None else None, rows, in_dim, hidden, _C.stream()), "lstm")
            ms = _event_ms_median(fn)
            flop = 2.0 * 2 * 3 * hidden * in_dim * rows
            nbytes = rows * 4 * (in_dim + 2 * hidden + (8 * hidden if gs is not None else 0))
            out["layers"]["%d -> 2 x %d, %s" % (in_dim, hidden, tag)] = {
                "us": 1e3 * ms, "tflops": flop / (ms * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": flop / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "algorithmic_bytes": nbytes, "GB_per_s": nbytes / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    for T, nrows, in_dim, hidden in ((100, 64, 100, 50), (100, 64, 128, 64), (100, 4096, 128, 64)):
        lstm = torch.nn.LSTM(input_size=in_dim, hidden_size=hidden, num_layers=1, bidirectional=True).to(device)
        x = torch.randn(T, nrows, in_dim, device=device)
        with torch.no_grad():
            fn = lambda: hag.lstm_seq_forward(x, lstm)
            ms = _event_ms_median(fn, rounds=7, per_round=2)
        flop = 2.0 * 2 * 4 * hidden * (in_dim + hidden) * nrows * T
        out["sequence"]["T=%d rows=%d %d -> 2 x %d" % (T, nrows, in_dim, hidden)] = {"us_per_call": 1e3 * ms, "us_per_step": 1e3 * ms / T,
                                                                                   "tflops": flop / (ms * 1e-3) / 1e12}
    return out


def bench_scoring_sharded(device, world, rank, per_gpu=125_000, reps=5):
    """configs[4] across the ranks: 125 000 windows per GPU of one long series.  Rank 0 holds the trained weights; ONE RCCL
    broadcast of the parameter arenas (~1 MB) gives them to every rank (parallel.broadcast_weights); every rank then scores its
    window range (+ halos) and all-gathers the per-window / per-timestep vectors.  Hyperbolic branch (row-wise Poincare
    distance) and Euclidean branch (un-roll median + DTW): hypad_amd/parallel.py.  All ranks call this.  At one GPU a
    world-size-1 `nccl` group is created for the section, so the collectives really run through RCCL (communicator creation,
    broadcast, all_gather_into_tensor, all_reduce) instead of being skipped."""
    import torch.distributed as dist
    from hypad_amd import parallel as par
    from hypad_amd.models import tadgan
    own_group = False
    if not dist.is_initialized():
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
        own_group = True
    out = {"rccl_world_size": dist.get_world_size(), "backend": dist.get_backend()}
    n = per_gpu * world
    g = torch.Generator(device=device).manual_seed(3)
    series = (torch.rand(n + S - 1, device=device, generator=g) * 2 - 1).contiguous()
    for hyperbolic in (True, False):
        torch.manual_seed(1000 * rank + 7)                               # every rank starts from DIFFERENT weights ...
        mods = [tadgan.Encoder(S, L).to(device).eval(), tadgan.Decoder(S, L, hyperbolic).to(device).eval(), tadgan.CriticX(S, L).to(device).eval()]
        out["broadcast_bytes"] = par.broadcast_weights(mods, src=0)       # ... and takes rank 0's
        enc, dec, cx = mods
        if hyperbolic:
            fn = lambda t: par.score_windows_sharded(series, enc, dec, cx, S, "mult", x_row_stride=1, as_tensor=t)
            want = n
        else:
            y = series.unfold(0, S, 1)[:n].contiguous()                   # (N, S) window matrix for the un-roll
            fn = lambda t: par.score_anomalies_sharded(y, enc, dec, cx, S, rec_error_type="dtw", comb="mult", as_tensor=t)
            want = n + S - 1
        rates = {}
        for as_tensor in (True, False):                                   # device result (stays in HBM) / NumPy result (the reference's return type)
            fn(as_tensor)
            best = float("inf")
            for _ in range(3):                                            # three rounds of `reps` calls, the fastest round counts: one allocator or
                torch.cuda.synchronize()                                  # collective set-up hiccup of a few ms inside a 6 ms round once halved the figure
                t0 = time.perf_counter()
                for _ in range(reps):
                    scores = fn(as_tensor)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / reps)
            rates[as_tensor] = n / best
        assert scores.shape == (want,) and np.isfinite(scores).all()
        # the same call replayed as ONE hipGraph (parallel.replay_scorer): the ~25 small launches behind the two large kernels stop costing
        # their enqueue / dispatch gaps; reported next to `value`, same scores
        graph_rate, graph_same = None, None
        try:
            eager = fn(True).clone()
            arenas = [m.arena() for m in mods]
            src = series if hyperbolic else y
            rep = lambda: par.replay_scorer(lambda: fn(True), src, *arenas, key=("bench", hyperbolic))
            rep(); rep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                got = rep()
            torch.cuda.synchronize()
            graph_rate = n / ((time.perf_counter() - t0) / reps)
            graph_same = bool(torch.equal(got, eager))
        except Exception as e:                                           # (e.g. a collective that cannot be captured on this stack: reported, not fatal)
            graph_rate = f"{type(e).__name__}: {e}"[:200]
        chk = torch.tensor([float(np.sum(scores))], device=device, dtype=torch.float64)      # every rank must hold the same scores
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert float(lo) == float(hi), "ranks disagree on the gathered scores"
        out["hyperbolic" if hyperbolic else "euclidean_dtw"] = {"windows": n, "value": rates[True], "unit": "windows/s",
                                                                "numpy_result_value": rates[False], "graph_replay_value": graph_rate,
                                                                "graph_replay_same_scores": graph_same}
    out["what"] = ("hyperbolic: forward + row-wise Poincare distance + KDE critic modes by window range, all-gather, global steps on every "
                   "rank; euclidean_dtw: forward + un-roll median + DTW(11) + rolling mean by timestep range, all-gather, z-score + KDE critic "
                   "scores + 'mult' on every rank; value = scores left on the device, numpy_result_value = returned as NumPy (D2H included)")
    if own_group:
        dist.destroy_process_group()
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this process has not touched the GPU yet (importing torch and counting
    devices does not initialise it), so it starts `torch.distributed.run` with one fresh rank per GPU as a CHILD process,
    relays its output (rank 0 prints the JSON line) and returns its exit code.  Never an exec: a process that has
    initialised the GPU must not replace itself."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _claim_stdout():
    """stdout carries ONE JSON line.  Libraries write there too (RCCL prints a version banner from C when its first communicator
    is created): point file descriptor 1 at stderr for the whole run and keep the real stdout for the JSON line."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


HEADLINE_MAX_BYTES = 4096           # the driver keeps a bounded tail of stdout: the LAST line must be small and whole

_HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_algorithmic", "traffic_ratio", "launch_ms",
              "launches_per_step", "iterations_per_launch", "flop_per_launch")
_CPU_KEYS = ("value", "unit", "cores", "host_cores", "kind", "sample")


def _finite(o):
    """The same structure with every non-finite float replaced by None (strict JSON has no NaN / Infinity)."""
    if isinstance(o, float):
        return o if np.isfinite(o) else None
    if isinstance(o, (np.floating, np.integer)):
        return _finite(o.item())
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def headline(full, provisional=False):
    """The compact strict-JSON object the driver parses: the contract's keys, `roofline` of the dominant kernel, `cpu_baseline`, and one
    figure per detail section (each section whole: bench_detail.json in the working directory, and one line per section on stderr).
    Never raises over its size: what is optional is dropped step by step until the line fits, the contract's keys always stay."""
    full = _finite(full)
    head = {k: full[k] for k in _HEAD_KEYS if k in full}
    cfg = full.get("config", {})
    head["config"] = {k: cfg[k] for k in ("workload", "signals_per_gpu", "iterations_per_step", "launch", "rccl_world_size") if k in cfg}
    if "roofline" in full:
        head["roofline"] = {k: full["roofline"][k] for k in _ROOF_KEYS if k in full["roofline"]}
    if isinstance(full.get("cpu_baseline"), dict):
        head["cpu_baseline"] = {k: full["cpu_baseline"][k] for k in _CPU_KEYS if k in full["cpu_baseline"]}
        if head["cpu_baseline"].get("value") and full.get("value"):
            head["vs_cpu_baseline"] = full["value"] / head["cpu_baseline"]["value"]
    also, failed = {}, []
    for name, sec in full.items():                     # one number per section, so the line says what the detail file holds
        if isinstance(sec, dict) and name not in ("config", "roofline", "cpu_baseline", "final_losses"):
            if isinstance(sec.get("value"), (int, float)):
                also[name] = round(sec["value"], 1)
            elif "error" in sec:
                failed.append(name)
    head["also_windows_per_s"] = also
    if failed:
        head["sections_failed"] = failed
    if provisional:
        head["provisional"] = True                     # (printed before the detail sections ran: the same run's final line follows)
    head["detail"] = "bench_detail.json"
    dumps = lambda: json.dumps(head, allow_nan=False, separators=(",", ":"))
    line = dumps()
    # never grow past what the driver keeps: drop the optional parts in this order, keep the contract
    _ROOF_CORE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel")
    steps = [lambda: head.pop("also_windows_per_s", None), lambda: head.pop("sections_failed", None),
             lambda: head.__setitem__("roofline", {k: v for k, v in head.get("roofline", {}).items() if k in _ROOF_CORE}),
             lambda: head.__setitem__("cpu_baseline", {k: (v[:120] if isinstance(v, str) else v) for k, v in head.get("cpu_baseline", {}).items()})
             if "cpu_baseline" in head else None,
             lambda: head["config"].__setitem__("workload", str(head["config"].get("workload", ""))[:160]),
             lambda: head.__setitem__("config", {"workload": str(head["config"].get("workload", ""))[:80]}),
             lambda: head.pop("roofline", None), lambda: head.pop("cpu_baseline", None)]
    for shrink in steps:
        if len(line) < HEADLINE_MAX_BYTES:
            break
        shrink()
        line = dumps()
    return line


def emit(full, json_out, detail_path=None, final=True):
    """Every section whole -> bench_detail.json (in the working directory unless a path is given) and, with the final line, stderr (one
    section per line); the compact headline -> a new LAST line of stdout (`provisional` until the final one)."""
    full = _finite(full)
    detail_path = detail_path or os.path.join(os.getcwd(), "bench_detail.json")
    try:
        with open(detail_path, "w") as f:
            json.dump(full, f, allow_nan=False, indent=1)
    except OSError as e:                               # (a read-only checkout must not cost the run its line)
        print("bench_detail.json not written: %s" % e, file=sys.stderr)
    if final:
        for name, sec in full.items():
            if isinstance(sec, (dict, list)):
                print(json.dumps({name: sec}, allow_nan=False), file=sys.stderr)
        sys.stderr.flush()
    json_out.write(headline(full, provisional=not final) + "\n")
    json_out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--signals-per-gpu", type=int, default=1, help="independent signals (models) trained side by side on each GPU")
    ap.add_argument("--euclidean", action="store_true", help="configs[0]-style hyperbolic=False instead of configs[1]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scoring", action="store_true", help="skip the anomaly-score windows/s section")
    ap.add_argument("--no-graph", action="store_true", help="launch every epoch eagerly instead of replaying its captured hipGraph")
    ap.add_argument("--host-shuffle", action="store_true", help="draw the epoch's shuffles with torch (rand + argsort) instead of inside the captured epoch")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the euclidean / multivariate / signals32 sections")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2] (8 signals per GPU) secondary line")
    ap.add_argument("--strict", action="store_true", help="fail (instead of reporting) when the per-kernel times add up to more than the step")
    ap.add_argument("--no-drop-in", action="store_true", help="skip timing the reference-style loop over hypad_amd.train's iteration functions")
    ap.add_argument("--no-sharded-scoring", action="store_true", help="skip configs[4]-style scoring sharded over all ranks (RCCL collectives; "
                                                                      "a one-rank nccl group at --gpus 1)")
    ap.add_argument("--sharded-scoring", action="store_true", help="run that section at --gpus > 1 too (it is on by default at one GPU)")
    ap.add_argument("--detail", default=None, help="where every section goes whole (default: bench_detail.json in the working directory)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    json_out = _claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1 or (os.environ.get("HYPAD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ):
        # (HYPAD_BENCH_FORCE_DIST=1 under the launcher at ONE rank: the RCCL barrier / max-over-ranks path of the multi-GPU line, exercised
        # on the one GPU a box has -- scripts/check_launcher.sh)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from hypad_amd import build as hb
    hb.build()                      # (rank-safe: file lock; prints nothing -- stdout carries the ONE JSON line)
    hyperbolic = not args.euclidean
    spg = args.signals_per_gpu
    cfg_main = Cfg("configs[1]" if hyperbolic else "configs[0] on the GPU", hyperbolic=hyperbolic)
    eng, x = build_engine(spg, rank, hyperbolic, device, cfg_main)
    gen = torch.Generator(device=device).manual_seed(100 + rank)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step, losses = make_step(eng, x, spg, gen, device, graph=not args.no_graph, host_shuffle=args.host_shuffle, cfg=cfg_main)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    eng.check_status()              # (a resident critic launch that gave up would have raised here: the timed epochs were complete)
    last = losses.float().mean(dim=(0, 1)).cpu().tolist()
    assert all(np.isfinite(last)), "training diverged"

    # ---- per-kernel durations, HIP events on the launch stream (same workload, after the timed region)
    prof = profile_kernels(eng, x, spg, device, cfg=cfg_main)
    mac_main = cfg_main.mac()
    per_launch, launches, epoch_share, dom = prof["per_launch"], prof["launches"], prof["epoch_share_ms"], prof["dominant"]
    persistent, producers, n_it = prof["persistent"], prof["producers"], N_CRITICS * N_BATCHES
    flop = prof["flop_per_launch"][dom]                  # algorithmic FLOPs of ONE launch of the dominant kernel
    achieved = flop / (per_launch[dom] * 1e-3) / 1e12
    # the per-kernel times must add up to no more than the step they were taken from (+ the three small launches not listed:
    # shuffle, pack, decay): they are kernel times, not event-path times
    share_sum = float(sum(epoch_share.values()))
    ms_step = 1e3 * elapsed / args.steps
    share_ok = share_sum <= 1.01 * ms_step
    if args.strict:
        assert share_ok, (share_sum, ms_step)

    # memory-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside the run, so this is the figure
    # of the committed rocprofv3 --pmc passes of this same command (profiles/r0N_pmc_traffic.json, gfx950 FETCH_SIZE correction
    # of MI355X_MICROARCH.md) -- but only while the kernel sources are the ones that profile was taken on (sha256 recorded in the
    # profile); null otherwise
    traffic, traffic_note = None, "no committed PMC profile matches these kernel sources"
    try:
        from hypad_amd.build import source_digest
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            pmc = json.load(open(path))
            if pmc.get("source_sha256") == source_digest() and dom in pmc["kernels"] and hyperbolic and spg == 1:
                traffic = pmc["kernels"][dom]["hbm_bytes_per_launch"]
                traffic_note = f"profiles/{name} (same kernel sources: sha256 matches)"
                break
    except (OSError, KeyError, ValueError):
        pass

    # matrix-pipe utilisation by counter (SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles of the dispatch, scripts/profile_r03.sh): the
    # committed rocprofv3 --pmc pass, under the same source-digest rule as `traffic`
    roofline_mfma = None
    try:
        mfname = next(n for n in ("r06_mfma_util.json", "r05_mfma_util.json", "r04_mfma_util.json", "r03_mfma_util.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        path = os.path.join(ROOT, "profiles", mfname)
        mf = json.load(open(path))
        same = mf.get("source_sha256") == source_digest()
        roofline_mfma = {"source": "profiles/" + mfname + (" (same kernel sources: sha256 matches)" if same else
                                                                     " (kernel sources changed since: indicative only)"),
                         "formula": mf.get("formula"),
                         "kernels": {k: {"mfma_util": v.get("mfma_util"), "algorithmic_util": v.get("algorithmic_util"),
                                         "issued_over_algorithmic": v.get("issued_over_algorithmic")} for k, v in mf["kernels"].items()}}
    except (OSError, KeyError, ValueError, NameError, StopIteration):
        pass

    # algorithmic memory-side bytes of ONE resident critic launch (the kernel table): every record written once by
    # its producer and read once by its critic; every chunk's merged gradient share -- the valid accumulator quads of its critic:
    # (Q (in + 1) + (nh - 1) Q (L + 1) + L + 1) x 16 bytes, Q = ceil(L / 4) -- written once and read by each of the B/16 chunks
    traffic_algorithmic = None
    if dom == "critic_persistent_kernel":
        nchunks, q = B // 16, (L + 3) // 4
        rec = sum(eng.epoch_records(N_BATCHES, N_CRITICS, c)[1].record_floats for c in (0, 1)) * 4 * nchunks
        share_bytes = ((q * (S + 1) + 3 * q * (L + 1) + L + 1) + (q * (L + 1) + q * (L + 1) + L + 1)) * 16 * nchunks     # critic_x (4 hidden layers), critic_z (2)
        # (where every critic's chunks share an XCD -- counters[5], read from the hardware by the kernel -- the shares never leave that
        # XCD's L2: the memory side then sees the records only)
        shares_in_l2 = int(eng.counters[5]) == 2 * spg
        traffic_algorithmic = float(spg * n_it * ((2 if producers else 1) * rec + (0 if shares_in_l2 else share_bytes * (1 + nchunks))))

    # algorithmic memory-side bytes of ONE generator launch and ONE dW + Adam launch (DESIGN.md §3 layout): gen_kernel reads the packed
    # generator weights once (forward and transposed copies) and the gathered windows, and writes the iteration scratch once (operand
    # rows of every weight gradient, saved gates / activations, dropout masks); dw_adam_kernel reads that scratch once, moves 28 bytes
    # of optimiser traffic per updated parameter (p, m, v read; p, m, v written; + the gradient never leaves the accumulators) and
    # writes every updated matrix element to its two packed positions
    traffic_alg_all = {dom: traffic_algorithmic} if traffic_algorithmic else {}
    try:
        import ctypes
        from hypad_amd import _C
        o_, st_, c_ = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _C.check(_C.lib.hypad_packed_region(ctypes.byref(eng.dims), ctypes.byref(o_), ctypes.byref(st_), ctypes.byref(c_)), "packed_region")
        scratch_bytes, packed_bytes = 4.0 * o_.value, 4.0 * c_.value
        upd = mat = 0
        for net in ("enc", "dec"):
            for name, off, shape in eng.catalogue(net):
                n = int(np.prod(shape))
                if "weight_hh" in name:
                    continue                                        # never read, decayed once per epoch (decay_steps_kernel)
                if "lstm." in name:
                    n = n * 3 // 4                                  # the f-gate rows meet c0 = 0: decay only
                upd += n
                mat += n if len(shape) == 2 else 0
        traffic_alg_all["gen_kernel"] = spg * (packed_bytes + scratch_bytes + 2.0 * B * S * 4)
        traffic_alg_all["dw_adam_kernel"] = spg * (scratch_bytes + 28.0 * upd + 8.0 * mat)
    except Exception as e:                                          # (reported, never fatal for the line)
        traffic_alg_all["error"] = f"{type(e).__name__}: {e}"[:200]
    traffic_all = {}
    try:
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                pmc = json.load(open(path))
                same = pmc.get("source_sha256") == source_digest()
                for k in ("critic_persistent_kernel", "gen_kernel", "dw_adam_kernel"):
                    if k in pmc["kernels"] and hyperbolic and spg == 1:
                        b_ = pmc["kernels"][k]["hbm_bytes_per_launch"]
                        traffic_all[k] = {"pmc_bytes_per_launch": b_, "algorithmic": traffic_alg_all.get(k),
                                          "ratio": (b_ / traffic_alg_all[k]) if traffic_alg_all.get(k) else None,
                                          "source": f"profiles/{name}" + ("" if same else " (kernel sources changed since: indicative only)")}
                break
    except (OSError, KeyError, ValueError, NameError):
        pass

    # ---- the headline object: everything the contract asks for is known here, so rank 0 prints it NOW (a provisional last line) and
    # again after the detail sections with their figures filled in: nothing that happens later -- a detail section that raises, hangs
    # in a collective or takes the process down -- can cost the run its line (main.py:32-70 is the loop these sections stand for)
    guard = RankGuard(dist, device, world)
    out = None
    if rank == 0:
        windows = world * spg * N_BATCHES * B * args.steps
        out = {
            "metric": "training windows/sec (seq_len=100)",
            "value": windows / elapsed,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (("configs[1]" if hyperbolic else "configs[0] on the GPU") + ": univariate synthetic (sine+noise+jump), hyperbolic=%s, batch=64, window=100, "
                                    "latent=20, 1916 windows/signal, %d signal(s) per GPU; step = 1 epoch = 29 x "
                                    "(5 critic_x + 5 critic_z + 1 decoder) iterations") % (hyperbolic, spg),
                       "signals_per_gpu": spg, "iterations_per_step": (2 * N_CRITICS + 1) * N_BATCHES,
                       "iteration_windows_per_s": windows * (2 * N_CRITICS + 1) / elapsed,
                       "critic_phase": ("one resident launch per epoch (critic_persistent_kernel)" + (", records produced by that launch's own "
                                        "producer workgroups" if producers else " behind a precompute launch")) if persistent else "one launch per iteration",
                       "launch": "eager" if args.no_graph else "hipGraph replay of the captured epoch",
                       "shuffles": "torch rand + argsort per epoch (host-driven)" if args.host_shuffle else
                                   "drawn inside the epoch's launch sequence (hypad_epoch_shuffles)", "rccl_world_size": world},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_algorithmic": traffic_algorithmic, "traffic_algorithmic_by_kernel": traffic_alg_all,
                         "traffic_by_kernel": traffic_all,
                         "traffic_ratio": (traffic / traffic_algorithmic) if traffic and traffic_algorithmic else None,
                         "launch_ms": per_launch[dom], "launches_per_step": launches[dom],
                         "iterations_per_launch": n_it if dom == "critic_persistent_kernel" else 1,
                         "us_per_critic_iteration": 1e3 * prof["kern_ms"]["critic_iteration"],
                         "kernel_ms": per_launch, "epoch_share_ms": epoch_share, "epoch_share_sum_ms": share_sum,
                         "epoch_share_le_step": bool(share_ok), "flop_per_launch": flop,
                         "flop_per_launch_parts": ({"critic_iterations": 2.0 * mac_main["critic_iteration"] * n_it * B * spg,
                                                    "record_producers": 2.0 * (mac_main["critic_precompute"] if producers else 0) * n_it * B * spg}
                                                   if dom == "critic_persistent_kernel" else None)},
            "final_losses": {"loss": last[0], "aux": last[1]},
        }
        if roofline_mfma is not None:
            out["roofline_mfma"] = roofline_mfma
        out["config"]["critics_on_one_xcd"] = int(eng.counters[5])          # placement census of the last resident critic launch (of 2 per signal)

    def checkpoint(final=False):
        if rank == 0:
            emit(out, json_out, detail_path=args.detail, final=final)

    def put(name, res):
        if rank == 0 and res is not None:
            out[name] = res
        return res

    checkpoint()
    local = RankGuard().run                 # sections only rank 0 runs: guarded, no collective
    # ---- BASELINE.json configs[2]'s per-GPU share as a secondary line: 8 signals (models) trained side by side on this GPU,
    # replayed as a captured hipGraph like the headline (and once more eagerly, so a host-bound launch path is visible)
    if spg == 1 and hyperbolic and not args.no_secondary:
        put("secondary", guard.run(bench_signals, 8, rank, device, gen, guard=guard, world=world))

    # ---- the other BASELINE.json configs as sections of the same line, each the graph-replayed epoch of its shape with its own roofline:
    # configs[0] on the GPU (hyperbolic=False), configs[3] (5 channels x 30 = window 150, batch 256, 20 480 windows: the compile-time
    # <150, 20, 256> kernels), the reference's shipped multivariate.yaml shapes (WADI: window 123, SWAT: window 51; batch 64) and 32 signals
    # (models) per GPU -- every CU holds a critic workgroup
    if spg == 1 and hyperbolic and not args.no_extra_configs:
        put("euclidean", guard.run(bench_signals, 1, rank, device, gen, warmup=3, steps=20, cfg=Cfg("configs[0]", hyperbolic=False), eager=False, guard=guard, world=world,
                                   what="configs[0] on the GPU: univariate, hyperbolic=False, batch 64, window 100, 1 916 windows, 1 signal"))
        put("multivariate", guard.run(bench_signals, 1, rank, device, gen, warmup=2, steps=8, cfg=Cfg("configs[3]", S=150, B=256, n_windows=20480, data="uniform"),
                                      eager=False, guard=guard, world=world,
                                      what="configs[3]: multivariate stand-in (SURVEY.md 8d config 4): window 150 = 5 channels x 30, batch 256, "
                                           "20 480 windows U(-1, 1), hyperbolic=True; step = 1 epoch = 80 x (5 + 5 + 1) iterations"))
        # the shapes the reference's own configs/multivariate.yaml:5-7 ships (batch_size 64; signal_shape 123 = WADI, 51 = SWAT,
        # utils/dataloader_multivariate.py): compile-time <123, 20, 64> / <51, 20, 64> instantiations of the critic / generator / dW kernels
        for tag, width in (("multivariate_wadi", 123), ("multivariate_swat", 51)):
            put(tag, guard.run(bench_signals, 1, rank, device, gen, warmup=2, steps=6, cfg=Cfg(tag, S=width, B=64, n_windows=20480, data="uniform"),
                               eager=False, guard=guard, world=world,
                               what="the reference's configs/multivariate.yaml as shipped (%s): signal_shape %d, batch 64, 20 480 windows U(-1, 1), "
                                    "hyperbolic=True; step = 1 epoch = 320 x (5 + 5 + 1) iterations" % ("WADI" if width == 123 else "SWAT", width)))
        s32 = put("signals32", guard.run(bench_signals, 32, rank, device, gen, warmup=2, steps=8, eager=False, guard=guard, world=world,
                                         what="32 signals (models) per GPU, otherwise as configs[1]: 4x configs[2]'s per-GPU share"))
        if rank == 0 and world == 1 and "error" not in s32:       # (one process: the 32-model product loop with its set-up and checkpoint files)
            s32["product_loop"] = local(bench_signals_product, 32, device)
    if spg == 1 and hyperbolic and rank == 0 and not args.no_drop_in:
        put("call_level", local(bench_call_level, device))

    # ---- the drop-in call surface (train.py:315-356 -> hypad_amd/train.py): the reference's own epoch loop over the same 29
    # minibatches with the three iteration functions swapped for hypad_amd's (host NumPy / torch RNG, one H2D of noise per call)
    if spg == 1 and rank == 0 and not args.no_drop_in:
        drop_in = put("drop_in", local(bench_drop_in, hyperbolic, device))
        if "value" in drop_in:                                    # the reference's call surface against the device-RNG path and the CPU
            drop_in["vs_resident_path"] = drop_in["value"] / out["value"] * world
    if rank == 0 and not args.no_scoring:
        def scoring_sections():
            sc, hbm, roof = bench_scoring(device, cpu_sample=0 if args.no_cpu_baseline or world > 1 else 40000)
            put("scoring", sc); put("roofline_hbm", hbm); put("roofline_scoring", roof)
            # configs[4] whole on ONE GPU: 10^6 windows (400 MB of windows; the four (N, S) outputs 1.6 GB), errors smoothed over 10^4
            torch.cuda.empty_cache()
            put("scoring_1e6", bench_scoring(device, n=1_000_000, reps=3, smooth=10_000, kernels=False)[0])
            torch.cuda.empty_cache()
            put("roofline_lstm", bench_lstm_layers(device))
        err = local(scoring_sections)
        if isinstance(err, dict):
            put("scoring_error", err)

    # ---- the CPU leg: rank 0 at N = 1 only (task contract).  Behind every GPU section of this process: it leaves torch's host thread pool
    # and allocator in another state (call_level's set-up, 20 pinned staging buffers, took 870 ms behind it against 10-16 ms in front)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        put("cpu_baseline", local(cpu_baseline, hyperbolic))
        drop_in = out.get("drop_in")
        if isinstance(drop_in, dict) and "value" in drop_in and out["cpu_baseline"].get("value"):
            drop_in["vs_cpu_baseline"] = drop_in["value"] / out["cpu_baseline"]["value"]

    # ---- the sections whose collectives sit inside the PRODUCT path (train_signals_resident's gather of the metrics, the sharded
    # scoring pass's all-reduce / all-gather): they run last, behind one more print of the line, because a rank that fails inside
    # them leaves the others in a collective this script does not own
    checkpoint()
    # configs[2] through the product path under a process group (train_signals_resident + plan_signal_groups): every rank, by default
    if spg == 1 and hyperbolic and not args.no_secondary:
        put("signals_sharded", guard.run(bench_signals_sharded, device, world, rank))
    # (scoring_sharded -- one GPU: on by default, through a one-rank RCCL group; several GPUs: only on request)
    if not args.no_sharded_scoring and not args.no_scoring and (world == 1 or args.sharded_scoring):
        put("scoring_sharded", guard.run(bench_scoring_sharded, device, world, rank))
    checkpoint(final=True)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                  # (the line is out: a failing tear-down is reported, not fatal)
            print("process group tear-down: %s: %s" % (type(e).__name__, e), file=sys.stderr)


if __name__ == "__main__":
    sys.exit(main() or 0)
