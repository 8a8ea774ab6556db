"""The shapes the reference's shipped multivariate configuration runs: /root/reference/configs/multivariate.yaml:5-7 is
`signal_shape: 123, batch_size: 64` (WADI; SWAT is 51, same file's comment; windows from utils/dataloader_multivariate.py).  Round 6 gives
them compile-time instantiations -- critic_persistent_kernel / critic_iteration_kernel / gen_kernel / dw_adam_kernel <123, 20, 64> and
<51, 20, 64> -- and a timed section each (bench.py `multivariate_wadi`, `multivariate_swat`: 20 480 windows U(-1, 1), 320 x (5 + 5 + 1)
iterations per epoch).  Here the captured epoch AS TIMED is teacher-forced against oracle.train_iters (train.py:18-249 on CPU autograd), with
the method of tests/test_gpu_timed_shape_r5.py: eval mode, injected z / alpha planes, the weights an iteration starts from read out of a
prefix run that is asserted bit-identical to the long run."""
import numpy as np
import pytest
import torch

from test_gpu_timed_shape_r5 import L, NC, check_against_oracle, cu, generator_against_oracle, make_engine, oracle_modules, states

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,N,critic_its,gen_launches", [(123, 20480, (0, 511, 512, 1599), (0, 319)),      # WADI as timed: 1 600 iterations = slices of 512
                                                          (51, 6400, (0, 263, 499), (0, 99))])                # SWAT: 500 iterations, one slice
def test_shipped_multivariate_epoch_against_the_oracle(S, N, critic_its, gen_launches):
    B = 64
    nb = N // B
    nit = nb * NC
    mods = oracle_modules(S, True, 7 + S)
    w0 = states(mods)
    xw = np.random.default_rng(S).uniform(-1, 1, (1, N, S))
    rng = np.random.default_rng(5 + S)
    planes = dict(z_cx=rng.standard_normal((nit, 1, B, L)).astype(np.float32), alpha_cx=rng.uniform(size=(nit, 1, B, S)).astype(np.float32),
                  z_cz=rng.standard_normal((nit, 1, B, L)).astype(np.float32), alpha_cz=rng.uniform(size=(nit, 1, B, L)).astype(np.float32),
                  z_gen=rng.standard_normal((nb, 1, B, L)).astype(np.float32))
    eng = make_engine(S, B, [w0])
    assert eng.critic_phase_persistent()                        # the timed form: ONE resident critic launch (critic_persistent_kernel<S, 20, 64>)
    g = torch.Generator(device="cuda").manual_seed(3)
    perm = torch.rand(NC + 1, N, device="cuda", generator=g).argsort(dim=1)[:, : nb * B].to(torch.int32).contiguous()      # bench.make_step's host_shuffle branch
    dpl = {n: cu(v) for n, v in planes.items()}
    full = eng.train_epoch_graph(cu(xw), perm, nb, NC, False, noise=dpl).cpu().numpy()
    torch.cuda.synchronize()
    assert eng.status() == 0 and np.isfinite(full).all() and full.shape == (1, 11 * nb, 4)
    # the per-iteration launches (critic_iteration_kernel<S, 20, 64>) agree with the resident form to rounding
    from hypad_amd import _C
    eng2 = make_engine(S, B, [w0])
    per_it = eng2.train_epoch(cu(xw), perm, nb, NC, False, noise=dpl, flags=_C.EPOCH_PER_ITERATION).cpu().numpy()
    assert eng2.status() == 0 and np.abs(per_it[:, :20] - full[:, :20]).max() < 1e-4
    ri = perm.cpu().numpy()
    crit_rows, gen_rows, dpl, x = check_against_oracle(S, B, N, nb, xw, [w0], planes, full, ri, critic_its, None, (0,))
    final_critics = [{k: {n: v.cpu() for n, v in eng.state_dict(k, 0).items()} for k in ("cx", "cz")}]
    generator_against_oracle(S, B, nb, xw, [w0], planes, full, gen_rows, dpl, x, final_critics, gen_launches, (0,))


@pytest.mark.parametrize("S", [100, 123])
def test_queued_replays_of_a_sliced_phase_equal_eager_launches(S):
    """Round 6 regression.  A critic phase of more than 512 iterations runs as several resident launches (slices), each behind the zeroing of its
    exchange block.  Captured into the epoch's hipGraph that zeroing was a memset NODE, and graph replays queued back to back (bench.py's timed
    loop) went non-finite from the ninth epoch on at 20 480 windows -- window 100 and 123 alike; eager launches, and replays with a host
    synchronisation in front of each, never did.  The zeroing is a kernel node now: twelve queued replays == twelve eager epochs, bit for bit."""
    import bench
    N, B, nb = 20480, 64, 320
    dev = torch.device("cuda", 0)
    cfg = bench.Cfg("sliced", S=S, B=B, n_windows=N, data="uniform")
    results = []
    for graph in (True, False):
        gen = torch.Generator(device=dev).manual_seed(100)
        eng, x = bench.build_engine(1, 0, True, dev, cfg)
        step, losses = bench.make_step(eng, x, 1, gen, dev, graph=graph, cfg=cfg)
        for _ in range(12):
            step()                                     # no host synchronisation between the epochs
        torch.cuda.synchronize()
        assert eng.status() == 0
        assert bool(torch.isfinite(losses).all()), ("graph" if graph else "eager")
        results.append((losses.clone(), {net: eng.params[net].clone() for net in ("enc", "dec", "cx", "cz")}))
    assert torch.equal(results[0][0], results[1][0])
    for net in ("enc", "dec", "cx", "cz"):
        assert torch.equal(results[0][1][net], results[1][1][net]), net
