"""bench.py's last stdout line is what the driver parses: compact, strict JSON, with the contract's keys (round 4's 22.8 KB line was not)."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _stub_full(big=False):
    sec = lambda v: {"value": v, "unit": "windows/s", "what": "x" * (3000 if big else 200), "roofline": {"kernel_ms": {"a": 1.0, "b": float("nan")}}}
    full = {
        "metric": "training windows/sec (seq_len=100)", "value": 651944.2, "unit": "windows/s", "n_gpus": 1, "steps": 20, "warmup": 3,
        "ms_per_step": 2.8469, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: " + "w" * 300, "signals_per_gpu": 1, "iterations_per_step": 319, "iteration_windows_per_s": 7e6,
                   "critic_phase": "c" * 200, "launch": "hipGraph replay of the captured epoch", "shuffles": "s" * 100, "rccl_world_size": 1},
        "roofline": {"bound": "mfma", "kernel": "critic_persistent_kernel", "achieved": 2.47, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.0157,
                     "traffic": 62.7e6, "traffic_source": "t" * 100, "traffic_algorithmic": 59.1e6, "traffic_ratio": 1.06, "launch_ms": 1.48,
                     "launches_per_step": 1, "iterations_per_launch": 145, "flop_per_launch": 3.663e9,
                     "traffic_algorithmic_by_kernel": {"k%d" % i: float(i) for i in range(50)}, "kernel_ms": {"k%d" % i: float(i) for i in range(50)},
                     "us_per_critic_iteration": float("inf")},
        "final_losses": {"loss": 0.1, "aux": float("nan")},
        "cpu_baseline": {"value": 3112.0, "unit": "windows/s", "cores": 1, "host_cores": 256, "kind": "port", "sample": "4 minibatches x ...",
                         "all_cores": {"value": None, "cores": 256, "sample": "y" * 300}},
    }
    for name in ("secondary", "euclidean", "multivariate", "signals32", "drop_in", "scoring", "scoring_1e6", "signals_sharded", "call_level"):
        full[name] = sec(1234.5)
    for name in ("roofline_hbm", "roofline_scoring", "roofline_lstm", "roofline_mfma", "scoring_sharded"):
        full[name] = {"kernels": {"k%d" % i: {"ms": 0.1 * i, "frac": 0.5} for i in range(30)}}
    return full


@pytest.mark.parametrize("big", [False, True])
def test_headline_is_compact_strict_json_with_the_contract_keys(big):
    line = bench.headline(_stub_full(big))
    assert "\n" not in line and len(line) < 4096
    d = json.loads(line, parse_constant=lambda c: pytest.fail("non-strict constant " + c))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in d, k
    assert d["config"]["workload"].startswith("configs[1]") and "model" not in d["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_algorithmic", "launch_ms"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for k in ("value", "cores", "host_cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["detail"] == "bench_detail.json"


def test_emit_writes_detail_file_sections_to_stderr_and_one_stdout_line(tmp_path, capsys):
    out = io.StringIO()
    path = str(tmp_path / "bench_detail.json")
    bench.emit(_stub_full(), out, detail_path=path)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    head = json.loads(lines[0])
    detail = json.load(open(path))
    assert detail["value"] == head["value"] and "roofline_scoring" in detail and "signals_sharded" in detail
    assert detail["final_losses"]["aux"] is None                     # NaN -> null: strict JSON everywhere
    err = capsys.readouterr().err.splitlines()
    names = [next(iter(json.loads(ln))) for ln in err if ln.startswith("{")]
    assert "secondary" in names and "roofline" in names and "cpu_baseline" in names
    assert capsys.readouterr().out == ""
