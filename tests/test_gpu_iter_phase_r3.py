"""Round 3: hypad_critic_x_iteration / hypad_critic_z_iteration as ONE-ITERATION PHASES of the hoisted critic form (pack of the frozen
generator half, record precompute, iteration launch, finalising launch: critic_fused.hip) -- the form the entry points take when the
caller's workspace has room for it (Engine.iteration_phase).  The per-iteration parity tests of test_gpu_parity.py -- reference
fixtures (train.py:18-186: losses, every gradient through Adam's first moment, trajectories), injected dropout against the manual
oracle, the host-RNG drop-in functions, three models side by side -- are run again in that form."""
import pytest
import torch

import test_gpu_parity as tp

pytestmark = pytest.mark.gpu


@pytest.fixture()
def phase(monkeypatch):
    from hypad_amd import _C
    from hypad_amd.engine import Engine
    monkeypatch.setattr(Engine, "iteration_phase", True)
    calls = {"n": 0}
    grow = Engine._room_for_iteration_phase

    def counted(self):
        grow(self)
        # the library takes the phase form exactly when the workspace has room: check that it has
        if self.iteration_phase:
            need = _C.lib.hypad_epoch_workspace_bytes(__import__("ctypes").byref(self.dims), 1, 1)
            assert self._ws_bytes >= need
            calls["n"] += 1
    monkeypatch.setattr(Engine, "_room_for_iteration_phase", counted)
    yield calls
    assert calls["n"] > 0, "no critic iteration went through the engine"


@pytest.mark.parametrize("tag,hyper", [("hyper_S100", True), ("eucl_S100", False)])
def test_reference_fixtures_in_the_phase_form(phase, tag, hyper):
    tp.test_training_iterations_match_reference_fixtures(torch.device("cuda"), tag, hyper)


@pytest.mark.parametrize("hyper", [True, False])
def test_injected_dropout_in_the_phase_form(phase, hyper):
    tp.test_training_iterations_with_injected_dropout_match_manual_oracle(torch.device("cuda"), hyper)


def test_drop_in_functions_in_the_phase_form(phase):
    tp.test_drop_in_iteration_functions_follow_host_rng(torch.device("cuda"))


def test_the_phase_form_counts_steps_like_the_stand_alone_launches(phase):
    """Same step / tick accounting as the stand-alone launches (counters) and the same trajectory.  The GPU time of the two forms is
    printed, not asserted: a timing claim does not belong in a parity suite the driver runs with -x (scripts/check_phase_form_speed.py
    asserts it on request)."""
    import bench
    from hypad_amd.engine import Engine
    dev = torch.device("cuda", 0)
    out = {}
    for on in (True, False):
        Engine.iteration_phase = on
        eng, x = bench.build_engine(1, 0, True, dev)
        xb = x[:, :64].contiguous()
        z = torch.randn(1, 64, 20, device=dev); ax = torch.rand(1, 64, 100, device=dev); az = torch.rand(1, 64, 20, device=dev)
        for _ in range(3):
            eng.critic_x_iteration(xb, None, z, ax, train_mode=False); eng.critic_z_iteration(xb, None, z, az, train_mode=False)
        best = float("inf")
        for _ in range(5):                              # (GPU time behind a sleeping stream; the best of five: a host hiccup must not count)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(20_000_000)
            a.record()
            for _ in range(10):
                eng.critic_x_iteration(xb, None, z, ax, train_mode=False)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 10 * 1e3)
        out[on] = (best, eng.counters.cpu().tolist()[:4], eng.params["cx"].clone())
    print("critic_x GPU us per call: phase form %.1f, stand-alone launches %.1f" % (out[True][0], out[False][0]))
    assert out[True][1] == out[False][1] == [53, 3, 0, 56]
    assert float((out[True][2] - out[False][2]).abs().max()) < 5e-3        # 53 Adam steps on the same data: same trajectory to rounding
