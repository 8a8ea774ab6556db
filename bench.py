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

The JSON line also carries
  roofline     -- the kernel holding the largest share of the epoch, its algorithmic FLOPs per launch over its mean
                  duration measured with HIP events on the launch stream (hypad_profile_iteration);
  cpu_baseline -- the CPU oracle (oracle/train_iters.py: the reference's nn.LSTM / autograd / Adam structure)
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

# Algorithmic MACs per window, SURVEY.md §8(d) accounting (1 MAC = 2 FLOP), split by the kernel that does the work.
F_ENC = 2 * (4 * 50) * S + 100 * L                                  # 42 000
F_DEC = L * 50 + 2 * 256 * 50 + 2 * 256 * 128 + 128 * S             # 104 936
F_CX = L * S + 3 * L * L + L                                        # 3 220
F_CZ = 2 * L * L + L                                                # 820
MAC_PER_WINDOW = {                                                  # hyperbolic=True
    # critic phase as hypad_train_epoch runs it (critic_fused.hip): the frozen generator's forwards of ALL iterations in
    # one precompute launch, then one launch per (critic_x || critic_z) iteration
    "critic_precompute": (F_DEC + S * S) + F_ENC,                           # decoder(z) + head, encoder(x)
    "critic_iteration": 10 * (F_CX + F_CZ),                                 # 3 fwd + 3 backward-data + second-order chain + 3 weight-gradient passes
    "gen": 2 * F_ENC + 4 * (F_DEC + S * S) + S * S + 2 * F_CX + 2 * F_CZ,   # fwd + backward-data of decoder_iteration
    "dw_gen": F_ENC + 2 * (F_DEC + S * S) + S * S,                          # its weight-gradient third
}
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


def build_engine(spg, rank, hyperbolic, device):
    from hypad_amd.engine import Engine
    from hypad_amd.models import tadgan
    eng = Engine(S, L, B, hyperbolic, n_signals=spg, device=device, lr=5e-4, seed=1234 + rank)
    xs = []
    for s in range(spg):
        sid = rank * spg + s
        torch.manual_seed(sid)      # random-init weights of the reference architecture (train.py:415-426 order)
        mods = dict(enc=tadgan.Encoder(S, L), dec=tadgan.Decoder(S, L, hyperbolic), cx=tadgan.CriticX(S, L), cz=tadgan.CriticZ(L))
        for k, m in mods.items():
            eng.load_state_dict(k, m.state_dict(), s)
        xs.append(synth_windows(N_WINDOWS, S, sid))
    x = torch.from_numpy(np.stack(xs)).to(device, torch.float32).contiguous()
    return eng, x


def cpu_baseline(hyperbolic, budget_s=24.0):
    """The oracle's epoch (same iteration mix) on a bounded number of minibatches, at 1 thread and at all cores."""
    from types import SimpleNamespace
    from oracle import tadgan as ot
    from oracle import train_iters as oi
    P = SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=L, lr=5e-4, hyperbolic=hyperbolic)
    data = torch.from_numpy(synth_windows(4 * B, S, 0)[:, :, None])
    ncores = os.cpu_count() or 1
    best = None
    # 1 thread and a modest intra-op pool: these layer sizes (<= 256 x 128) do not scale past a few cores, and a pool of
    # every core of a 256-core host spends minutes in thread hand-offs alone
    for threads in sorted({1, min(ncores, 8)}):
        torch.set_num_threads(threads)
        enc, dec, cx, cz = ot.build_models(S, L, hyperbolic, seed=0)
        opt = oi.make_optimizers(enc, dec, cx, cz, P)
        np.random.seed(0)
        batches = [data[i * B:(i + 1) * B] for i in range(4)]
        oi.train_epoch(batches[:1], enc, dec, cx, cz, opt, P)                  # warm-up: one minibatch's 11 iterations
        t0 = time.perf_counter()
        nb = 0
        while nb < 1 or (time.perf_counter() - t0 < budget_s / 2 and nb < 4096):
            oi.train_epoch(batches[nb % 4: nb % 4 + 1], enc, dec, cx, cz, opt, P)
            nb += 1
        dt = time.perf_counter() - t0
        rate = nb * B / dt
        if best is None or rate > best["value"]:
            best = dict(value=rate, cores=threads, sample=f"{nb} minibatches x (5 critic_x + 5 critic_z + 1 decoder) iterations, "
                                                               f"B={B}, window={S}, train-mode dropout, {dt:.1f} s")
    torch.set_num_threads(ncores)
    best.update(unit="windows/s", kind="port", host_cores=ncores)
    return best


def bench_scoring(device, n=125_000, reps=5):
    """BASELINE.json's second metric, anomaly-score windows/s, on this GPU's share of configs[4] (10^6 windows over 8
    GPUs): the test-loop forward with the hyperbolic row distance (anomaly_detection.py:67-113), then un-roll median +
    point and DTW errors + rolling mean + z-score (utils/anomaly_detection_utils.py:866-962, 516-524); and the HBM
    rate of the row-wise Poincare-ball kernels (algorithmic bytes per row)."""
    from hypad_amd import _C
    from hypad_amd.hyperspace import gmath
    from hypad_amd.models import tadgan
    from hypad_amd.utils import anomaly_detection_utils as adu
    torch.manual_seed(0)
    enc, dec, cx = tadgan.Encoder(S, L).to(device).eval(), tadgan.Decoder(S, L, True).to(device).eval(), tadgan.CriticX(S, L).to(device).eval()
    g = torch.Generator(device=device).manual_seed(3)
    x = (torch.rand(n, S, device=device, generator=g) * 2 - 1).contiguous()
    new = lambda *shape: torch.empty(*shape, device=device, dtype=torch.float32)
    hyper, eucl, hreal, critic, dist = new(n, S), new(n, S), new(n, S), new(n), new(n)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    ws_bytes = _C.lib.hypad_score_workspace_bytes(S, L, 1)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=device)

    def forward():
        _C.check(_C.lib.hypad_score_forward_packed(_C.ptr(enc.arena()), _C.ptr(dec.arena()), _C.ptr(cx.arena()), _C.ptr(x), 0, _C.ptr(hyper),
                                                   _C.ptr(eucl), _C.ptr(hreal), _C.ptr(critic), _C.ptr(dist), n, S, L, 1, ws.data_ptr(),
                                                   ws_bytes, _C.stream()), "score_forward_packed")

    def numerics():
        true = adu.unroll_true(x)
        pred, _ = adu.unroll_predictions(eucl, False)
        e1 = adu.rolling_mean(adu._point_wise_error(true, pred), 200)
        e2 = adu.rolling_mean(adu._dtw_error(true, pred, 10), 200)
        return adu.zscore_clip(e1), adu.zscore_clip(e2)

    t_fwd, t_num = timed(forward), timed(numerics)
    # row-wise ball kernels on 2 * 10^6 rows (0.8 GB per operand: well past the 256 MB Infinity Cache, so the rate is HBM's)
    m = 2_000_000
    xb = (torch.rand(m, S, device=device, generator=g) * 2 - 1).contiguous()
    ball = gmath.expmap0(0.03 * torch.randn(m, S, device=device, generator=g))
    other = gmath.expmap0(0.03 * torch.randn(m, S, device=device, generator=g))
    ops = {"expmap0": (lambda: gmath.expmap0(xb), 8 * S), "logmap0": (lambda: gmath.logmap0(ball), 8 * S),
           "project": (lambda: gmath.project(xb), 8 * S), "mobius_add": (lambda: gmath.mobius_add(ball, other), 12 * S),
           "poincare_rowdist": (lambda: gmath.poincare_rowdist(ball, other), 8 * S + 4)}
    gbps = {k: m * b / timed(f) / 1e9 for k, (f, b) in ops.items()}
    return {"windows": n, "value": n / (t_fwd + t_num), "unit": "windows/s",
            "forward_windows_per_s": n / t_fwd, "forward_tflops": n * 340312 / t_fwd / 1e12,
            "numerics_windows_per_s": n / t_num,
            "numerics": "un-roll median, point + DTW(11) errors, rolling mean(200), z-score",
            "hyperbolic_ops_GBps": gbps, "hyperbolic_ops_rows": m, "hbm_peak_GBps": 8000.0,
            "note": "row-wise ball ops: algorithmic bytes (800-1204 B/row at S=100) / time; includes the output allocation of the torch-facing wrappers"}


def bench_scoring_sharded(device, world, per_gpu=125_000, reps=3):
    """configs[4] across the ranks (opt-in: --sharded-scoring): 125 000 windows per GPU of one long series, every rank scoring
    its window range (+ halo) and all-gathering the per-window / per-timestep vectors (hypad_amd/parallel.py).  All ranks call this."""
    from hypad_amd import parallel as par
    from hypad_amd.models import tadgan
    torch.manual_seed(0)                                                  # the same weights on every rank
    enc, dec, cx = tadgan.Encoder(S, L).to(device).eval(), tadgan.Decoder(S, L, True).to(device).eval(), tadgan.CriticX(S, L).to(device).eval()
    n = per_gpu * world
    g = torch.Generator(device=device).manual_seed(3)
    series = (torch.rand(n + S - 1, device=device, generator=g) * 2 - 1).contiguous()
    par.score_windows_sharded(series, enc, dec, cx, S, "mult", x_row_stride=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        scores = par.score_windows_sharded(series, enc, dec, cx, S, "mult", x_row_stride=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert scores.shape == (n,) and np.isfinite(scores).all()
    return {"windows": n, "value": n / dt, "unit": "windows/s", "what": "forward + row-wise Poincare distance + KDE critic modes sharded by window "
            "range; all-gather; quantile z-score, rolling mean, combination 'mult' on the full vectors (every rank)"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--signals-per-gpu", type=int, default=1, help="independent signals (models) trained side by side on each GPU")
    ap.add_argument("--euclidean", action="store_true", help="configs[0]-style hyperbolic=False instead of configs[1]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scoring", action="store_true", help="skip the anomaly-score windows/s section")
    ap.add_argument("--sharded-scoring", action="store_true", help="also time configs[4]-style scoring sharded over all ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    import __graft_entry__
    __graft_entry__.build()
    hyperbolic = not args.euclidean
    spg = args.signals_per_gpu
    eng, x = build_engine(spg, rank, hyperbolic, device)
    gen = torch.Generator(device=device).manual_seed(100 + rank)
    losses = torch.empty(spg, (2 * N_CRITICS + 1) * N_BATCHES, 4, device=device)

    def step():
        # the DataLoader's shuffles: a fresh permutation for each of the 5 critic passes and the generator pass
        # (argsort of uniform keys: six independent uniform permutations from one batched sort instead of six randperm calls)
        perm = torch.rand(N_CRITICS + 1, N_WINDOWS, device=device, generator=gen).argsort(dim=1)[:, : N_BATCHES * B]
        eng.train_epoch(x, perm.to(torch.int32).contiguous(), N_BATCHES, N_CRITICS, train_mode=True, losses=losses)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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
    last = losses.float().mean(dim=(0, 1)).cpu().tolist()
    assert all(np.isfinite(last)), "training diverged"

    # ---- per-kernel durations, HIP events on the launch stream (same workload, after the timed region)
    # kind 4 = nine iterations of the critic phase exactly as train_epoch launches them (precompute of their records, the
    # first critic_x || critic_z iteration launch, the mean of eight steady-state launches back to back -- event overhead
    # amortised); kind 2 = decoder_iteration
    names = {4: ["critic_precompute", "critic_iteration_first", "critic_iteration"], 2: ["gen", "dw_gen"]}
    acc = {n: [] for v in names.values() for n in v}
    idx = torch.arange(B, device=device, dtype=torch.int32)
    for rep in range(60):
        for kind in (4, 2):
            ms = eng.profile_iteration(kind, x, idx, train_mode=True)
            if rep >= 10:
                for n, v in zip(names[kind], ms):
                    acc[n].append(v)
    kern_ms = {n: float(np.mean(v)) for n, v in acc.items() if n != "critic_iteration_first"}
    # the precompute runs ONCE per epoch for all 145 iterations; profiled here for nine iterations' rows (a lower bound
    # on its efficiency), so its epoch share is not extrapolated from this number
    launches = {"gen": N_BATCHES, "dw_gen": N_BATCHES, "critic_iteration": N_CRITICS * N_BATCHES + 1, "critic_precompute": 1}
    share = {n: kern_ms[n] * launches[n] for n in kern_ms}
    dom = max(share, key=share.get)
    flop = 2.0 * MAC_PER_WINDOW[dom] * B * spg
    achieved = flop / (kern_ms[dom] * 1e-3) / 1e12

    # memory-side bytes per launch of the dominant kernel: PMC counters cannot be read from inside the run, so this is the
    # figure of the committed rocprofv3 --pmc passes of this same command (profiles/r01_pmc_traffic.json, with the
    # gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); null when the profile does not cover the kernel / config
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
        key = {"gen": "gen_kernel", "dw_gen": "dw_adam_kernel", "critic_iteration": "critic_iteration_kernel"}.get(dom)
        if key in pm and hyperbolic and spg == 1:
            traffic = pm[key]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass

    sharded = bench_scoring_sharded(device, world) if args.sharded_scoring else None
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
                       "iteration_windows_per_s": windows * (2 * N_CRITICS + 1) / elapsed},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "kernel_ms": kern_ms, "epoch_share_ms": share, "flop_per_launch": flop},
            "final_losses": {"loss": last[0], "aux": last[1]},
        }
        if not args.no_scoring:
            out["scoring"] = bench_scoring(device)
        if sharded is not None:
            out["scoring_sharded"] = sharded
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hyperbolic)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
