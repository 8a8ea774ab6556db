"""bench.py under a process group cannot lose its headline: world-size-2 gloo, CPU only.  One rank's detail section
raises -- in the middle of its collectives, behind its last one, in front of its first -- and the job must still finish with rc 0 on
every rank, the ranks in step for the next section, and a LAST stdout line on rank 0 that parses and says n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    rank, world, where = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["FAIL_AT"]
    json_out = bench._claim_stdout()
    print("a library banner on stdout must not reach the line")            # (lands on stderr now)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    guard = bench.RankGuard(dist, torch.device("cpu"), world)
    out = {"metric": "training windows/sec (seq_len=100)", "value": 1.0e6 * world, "unit": "windows/s", "n_gpus": world, "steps": 2, "warmup": 1,
           "ms_per_step": 2.7, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[1]: stub", "rccl_world_size": world},
           "roofline": {"bound": "mfma", "kernel": "k", "achieved": 2.5, "peak": 157.3, "unit": "TFLOP/s", "frac": 2.5 / 157.3, "traffic": None,
                        "traffic_source": "none"}}
    detail = os.path.join(os.environ["OUT_DIR"], "bench_detail.json")
    if rank == 0:
        bench.emit(out, json_out, detail_path=detail, final=False)

    def section(name, fail):
        def body():
            if fail == "front" and rank == 1:
                raise RuntimeError("injected in front of the first collective of " + name)
            guard.barrier()
            ms = guard.max(3.0 + 2.0 * rank)
            if fail == "middle" and rank == 1:
                raise RuntimeError("injected between the collectives of " + name)
            guard.barrier()
            ms2 = guard.max(1.0 + rank)
            if fail == "behind" and rank == 1:
                raise RuntimeError("injected behind the last collective of " + name)
            return {"value": 1000.0 * ms + ms2, "unit": "windows/s"}
        res = guard.run(body)
        if rank == 0:
            out[name] = res
        return res

    a = section("first", None)
    b = section("second", where)
    c = section("third", None)
    assert "error" not in a and "error" not in c, (a, c)                   # the ranks are in step again behind the failed section
    assert a["value"] == c["value"] == 5002.0
    assert "error" in b, b                                                 # ... and the failed one is skipped on EVERY rank
    if rank == 0:
        bench.emit(out, json_out, detail_path=detail, final=True)
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("where", ["front", "middle", "behind"])
def test_a_failing_section_on_one_rank_costs_neither_the_line_nor_the_job(tmp_path, where):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAIL_AT=where, OUT_DIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    lines = outs[0][0].splitlines()
    assert len(lines) == 2 and outs[1][0] == ""                            # provisional + final on rank 0, nothing else on anybody's stdout
    first, last = (json.loads(ln, parse_constant=lambda c: pytest.fail(c)) for ln in lines)
    assert first["provisional"] is True and "provisional" not in last
    for d in (first, last):
        assert d["n_gpus"] == 2 and d["value"] == 2.0e6 and d["roofline"]["traffic_source"] == "none"
    assert last["sections_failed"] == ["second"] and last["also_windows_per_s"] == {"first": 5002.0, "third": 5002.0}
    assert "injected" in outs[1][1] and "banner" in outs[0][1]              # the traceback and the stray print went to stderr
    detail = json.load(open(tmp_path / "bench_detail.json"))
    assert "error" in detail["second"] and ("skipped on all ranks" in detail["second"]["error"])      # rank 0 did not fail itself: it was told


def test_headline_shrinks_instead_of_raising():
    sys.path.insert(0, ROOT)
    import bench
    full = {"metric": "m", "value": 1.0, "unit": "windows/s", "n_gpus": 8, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "w" * 5000},
            "roofline": {"bound": "mfma", "kernel": "k" * 3000, "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5, "traffic": None,
                         "traffic_source": "t" * 3000},
            "cpu_baseline": {"value": 1.0, "unit": "windows/s", "cores": 1, "kind": "port", "sample": "s" * 5000}}
    for i in range(400):
        full["section%d" % i] = {"value": float(i)}
    line = bench.headline(full)
    assert len(line) < bench.HEADLINE_MAX_BYTES
    d = json.loads(line)
    for k in bench._HEAD_KEYS:
        assert k in d
    assert d["config"]["workload"].startswith("www")
