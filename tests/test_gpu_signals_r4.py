"""One model per signal for a LIST of signals of different lengths (train.train_signals_resident; BASELINE configs[2]):
== the same signals trained one at a time with train_resident, bit for bit; two processes sharing the GPU (gloo) own disjoint
signals and the merged result equals the one-process run; per-signal checkpoint directories; first_signal at the engine level."""
import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
S, B = 100, 64
COUNTS = [3 * B + 5, 2 * B, 3 * B + 40, 2 * B + 63, 5 * B + 1]          # 3, 2, 3, 2, 5 minibatches per epoch


def windows(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n + S - 1)
    series = np.clip(np.sin(2 * np.pi * t / (40.0 + 7 * seed)) + 0.05 * rng.standard_normal(len(t)), -1, 1)
    return series[np.arange(n)[:, None] + np.arange(S)[None, :]]


def P_(epochs=3, hyper=True):
    return SimpleNamespace(batch_size=B, signal_shape=S, latent_space_dim=20, lr=5e-4, hyperbolic=hyper, resume=False, resume_epoch=0,
                           epochs=epochs, dataset="T", signal="x")


def flat(mods):
    return [{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in mods]


@pytest.mark.parametrize("hyper,group", [(True, None), (False, None), (True, 2)])
def test_signals_of_different_lengths_equal_single_signal_runs(tmp_path, monkeypatch, hyper, group):
    from hypad_amd import train as ht
    monkeypatch.chdir(tmp_path)
    data = [windows(n, i) for i, n in enumerate(COUNTS)]
    names = [f"sig{i}" for i in range(len(COUNTS))]
    res = ht.train_signals_resident(data, P_(hyper=hyper), names=names, seed=77, init_seed=500, group=group, log=None)
    assert sorted(res) == sorted(names)
    streams = {n: res[n]["stream"] for n in names}
    assert sorted(streams.values()) == list(range(5)) and streams["sig1"] == 0 and streams["sig3"] == 1 and streams["sig4"] == 4
    for i, name in enumerate(names):
        kind = "hyper" if hyper else "eucl"
        d = f"./trained_models/models_{kind}_T_3_0.0005/T/{name}"
        assert res[name]["path"] == d and sorted(os.listdir(d)) == sorted(f"{m}{sfx}.pt" for m in ("encoder", "decoder", "critic_x", "critic_z") for sfx in ("", "_2"))
        P = P_(hyper=hyper)
        P.signal = name + "_single"
        torch.manual_seed(500 + i)
        enc, dec, cx, cz, path, hist = ht.train_resident(data[i], P, seed=77, log=None, first_signal=streams[name])
        h = res[name]["history"]
        for k in ("cx", "cz", "dec", "hyper" if hyper else "mse"):
            assert h[k] == getattr(hist, k), (name, k)
        for a, b in zip(flat(res[name]["modules"]), flat([enc, dec, cx, cz])):
            for key in a:
                assert torch.equal(a[key], b[key]), (name, key)
        saved = torch.load(os.path.join(d, "encoder.pt"), weights_only=False)
        assert all(torch.equal(v.cpu(), flat([enc])[0][k]) for k, v in saved.state_dict().items())
    # different signals really trained differently
    assert res["sig0"]["history"]["dec"] != res["sig2"]["history"]["dec"]


def test_long_signals_with_host_drawn_shuffles_equal_single_signal_runs(tmp_path, monkeypatch):
    """Signals beyond the in-graph sort's 4 096 windows draw their shuffles with a torch device generator: ONE seeding rule
    (train._host_shuffle_generator: run seed x stream number) for the group loop and the single-signal loop, so the equivalence above
    holds there too -- and two long signals of a run do not share their permutations."""
    from hypad_amd import train as ht
    from hypad_amd.engine import Engine
    monkeypatch.chdir(tmp_path)
    counts = [Engine.SHUFFLE_MAX_WINDOWS + 4, Engine.SHUFFLE_MAX_WINDOWS + 70, Engine.SHUFFLE_MAX_WINDOWS + 30]      # 64, 65, 64 minibatches
    data = [windows(n, 20 + i) for i, n in enumerate(counts)]
    names = ["long0", "long1", "long2"]
    res = ht.train_signals_resident(data, P_(epochs=1), names=names, seed=31, init_seed=900, log=None, save=False)
    for i, name in enumerate(names):
        P = P_(epochs=1)
        P.signal = name + "_single"
        torch.manual_seed(900 + i)
        enc, dec, cx, cz, path, hist = ht.train_resident(data[i], P, seed=31, log=None, first_signal=res[name]["stream"])
        h = res[name]["history"]
        for k in ("cx", "cz", "dec", "hyper"):
            assert h[k] == getattr(hist, k), (name, k)
        assert h["repairs"] == 0 and hist.repairs == 0
        for a, b in zip(flat(res[name]["modules"]), flat([enc, dec, cx, cz])):
            for key in a:
                assert torch.equal(a[key], b[key]), (name, key)
    g0, g1 = (ht._host_shuffle_generator("cuda", 31, s) for s in (0, 1))
    assert not torch.equal(torch.rand(8, device="cuda", generator=g0), torch.rand(8, device="cuda", generator=g1))


def test_first_signal_makes_a_model_independent_of_its_group():
    """Engine level: slot k of a group whose first_signal is f == a single model with first_signal = f + k (device Philox noise and
    dropout, shuffles drawn in the captured epoch from the model's own window count), bit for bit."""
    from hypad_amd.engine import Engine
    from oracle import tadgan as ot
    nb, nc, k, f = 2, 2, 3, 4
    counts = [2 * B + 3, 2 * B + 50, 2 * B]
    x = torch.zeros(k, max(counts), S, device="cuda")
    for s in range(k):
        x[s, : counts[s]] = torch.from_numpy(windows(counts[s], 10 + s)).float()
    def load(eng, s, slot):
        torch.manual_seed(40 + s)
        for net, m in dict(enc=ot.Encoder(S, 20), dec=ot.Decoder(S, 20, True), cx=ot.CriticX(S, 20), cz=ot.CriticZ(20)).items():
            eng.load_state_dict(net, m.state_dict(), slot)
    grp = Engine(S, 20, B, True, n_signals=k, seed=9, first_signal=f)
    for s in range(k):
        load(grp, s, s)
    ri = torch.empty(k, nc + 1, nb * B, dtype=torch.int32, device="cuda")
    lg = [grp.train_epoch_graph(x, ri, nb, nc, True, shuffle_windows=counts).clone() for _ in range(2)]
    assert grp.check_status() == 0
    for s in range(k):
        assert int(ri[s].max()) < counts[s]
        one = Engine(S, 20, B, True, n_signals=1, seed=9, first_signal=f + s)
        load(one, s, 0)
        r1 = torch.empty(nc + 1, nb * B, dtype=torch.int32, device="cuda")
        l1 = [one.train_epoch_graph(x[s: s + 1, : counts[s]].contiguous(), r1, nb, nc, True, shuffle_windows=counts[s]).clone() for _ in range(2)]
        assert torch.equal(r1, ri[s])
        for e in range(2):
            assert torch.equal(l1[e][0], lg[e][s]), (s, e)
        for net in ("enc", "dec", "cx", "cz"):
            assert torch.equal(one.params[net][0], grp.params[net][s]), (s, net)
    assert not torch.equal(ri[0], ri[2])


@pytest.mark.parametrize("k,per_signal,series,nb,nc", [(3, True, False, 2, 3), (16, False, False, 2, 3), (2, False, True, 2, 3),
                                                       (32, False, False, 29, 5)])      # (the last: bench.py's `signals32` epoch, precompute launch in front)
def test_encoder_table_equals_the_encoder_run_per_pass(k, per_signal, series, nb, nc):
    """hypad_epoch_io.enc_table (ABI 6): encoder(x) evaluated once per window row in front of the critic phase and gathered by critic_z's
    record producers == the producers running the encoder on their rows in every pass, bit for bit (losses, weights): signals of
    different lengths with their own shuffle planes, 16 models (the resident launch with its own producers), the series view; as a
    replayed graph and eagerly."""
    from hypad_amd.engine import Engine
    from oracle import tadgan as ot
    counts = [2 * B + 3, 2 * B + 50, 2 * B][:k] if per_signal else [nb * B + 7] * k
    if series:
        x = torch.stack([torch.from_numpy(windows(counts[0], 20 + s)[:, 0].copy()) for s in range(k)]).float().cuda().contiguous()      # any series will do
        x = torch.cat([x, x[:, : S - 1]], 1).contiguous()                       # (k, counts[0] + S - 1): window n = x[n : n + S]
        stride = 1
    else:
        x = torch.zeros(k, max(counts), S, device="cuda")
        for s_ in range(k):
            x[s_, : counts[s_]] = torch.from_numpy(windows(counts[s_], 10 + s_)).float()
        stride = 0
    results = []
    for table in (False, True):
        eng = Engine(S, 20, B, True, n_signals=k, seed=11, first_signal=2)
        eng.enc_table = table
        for s_ in range(k):
            torch.manual_seed(60 + s_)
            for net, m in dict(enc=ot.Encoder(S, 20), dec=ot.Decoder(S, 20, True), cx=ot.CriticX(S, 20), cz=ot.CriticZ(20)).items():
                eng.load_state_dict(net, m.state_dict(), s_)
        ri = torch.empty((k, nc + 1, nb * B) if per_signal else (nc + 1, nb * B), dtype=torch.int32, device="cuda")
        sw = counts if per_signal else counts[0]
        ls = [eng.train_epoch_graph(x, ri, nb, nc, True, x_row_stride=stride, shuffle_windows=sw).clone() for _ in range(2)]
        eng.draw_shuffles(ri, sw)
        ls.append(eng.train_epoch(x, ri, nb, nc, True, x_row_stride=stride).clone())                                                     # eagerly
        assert eng.check_status() == 0
        assert (eng.__dict__.get("_enc_table") is not None) == table
        results.append((ls, {n: eng.params[n].clone() for n in ("enc", "dec", "cx", "cz")}))
    (la, pa), (lb, pb) = results
    for e in range(3):
        assert torch.isfinite(la[e]).all() and torch.equal(la[e], lb[e]), e
    for n in pa:
        assert torch.equal(pa[n], pb[n]), n


def test_the_resident_loops_repair_a_launch_that_gave_up(tmp_path, monkeypatch):
    """train_tadgan_resident and train_signals_resident read an epoch's losses when the next epoch is already queued and write their
    checkpoints from device copies taken between epochs.  A resident critic launch that gives up in epoch 0 (injected: the test bits
    of hypad_epoch_io.flags) makes that epoch AND the one queued behind it no-ops: both are repeated with per-iteration launches, the
    losses of each repeat are kept and a checkpoint epoch among them is copied again behind its repeat -- histories, final weights
    and the checkpoint files equal those of a run in the per-iteration form from the start."""
    from hypad_amd import _C
    from hypad_amd import train as ht
    from hypad_amd.models import tadgan
    monkeypatch.chdir(tmp_path)
    def files(d):
        out = {}
        for f in sorted(os.listdir(d)):
            if f.endswith(".pt"):
                out[f] = {k: v.cpu() for k, v in torch.load(os.path.join(d, f), weights_only=False).state_dict().items()}
        return out
    def same(a, b):
        assert sorted(a) == sorted(b) and len(a) >= 4, (sorted(a), sorted(b))
        for f in a:
            assert all(torch.equal(a[f][k], b[f][k]) for k in a[f]), f
    # one model
    got = []
    for tag, flags in (("ref", _C.EPOCH_PER_ITERATION), ("hit", 3 << _C.EPOCH_TEST_GIVE_UP_SHIFT)):
        torch.manual_seed(21)
        mods = [m.cuda().train() for m in (tadgan.Encoder(S, 20), tadgan.Decoder(S, 20, True), tadgan.CriticX(S, 20), tadgan.CriticZ(20))]
        d = tmp_path / ("one_" + tag)
        d.mkdir()
        P = P_(2)
        P.epoch_flags = flags
        h = ht.train_tadgan_resident(windows(2 * B + 5, 3), *mods, n_epochs=2, params=P, path=str(d), seed=5, log=None)
        got.append((vars(h), [{k: v.detach().cpu() for k, v in m.state_dict().items()} for m in mods], files(d)))
    assert got[0][0].pop("repairs") == 0 and got[1][0].pop("repairs") == 1        # the give-up is COUNTED in the history it repaired (not only logged)
    assert got[0][0] == got[1][0] and all(np.isfinite(got[1][0]["cx"])) and len(got[1][0]["cx"]) == 2
    for wa, wb in zip(got[0][1], got[1][1]):
        assert all(torch.equal(wa[k], wb[k]) for k in wa)
    same(got[0][2], got[1][2])                      # encoder_1.pt ...: the checkpoint of the failed epoch holds ITS weights
    # three models of different lengths in one group
    res = []
    for tag, flags in (("ref", _C.EPOCH_PER_ITERATION), ("hit", 3 << _C.EPOCH_TEST_GIVE_UP_SHIFT)):
        P = P_(2)
        P.epoch_flags = flags
        P.dataset = "grp_" + tag
        data = [windows(n, 30 + i) for i, n in enumerate((2 * B + 1, 2 * B + 40, 2 * B))]
        r = ht.train_signals_resident(data, P, names=["a", "b", "c"], seed=9, init_seed=100, log=None)
        res.append(r)
    for n in ("a", "b", "c"):
        assert res[0][n]["history"].pop("repairs") == 0 and res[1][n]["history"].pop("repairs") == 1, n
        assert res[0][n]["history"] == res[1][n]["history"], n
        for ma, mb in zip(res[0][n]["modules"], res[1][n]["modules"]):
            sa, sb = ma.state_dict(), mb.state_dict()
            assert all(torch.equal(sa[k], sb[k]) for k in sa), n
        same(files(res[0][n]["path"]), files(res[1][n]["path"]))


def test_small_groups_on_lanes_equal_the_groups_one_after_the_other(tmp_path, monkeypatch):
    """Signals of several different lengths leave the planner with small groups; train_signals_resident deals them over lanes (streams
    that run beside each other, hypad_amd/streams.py) -- the same histories, final weights and checkpoint files as with every group
    on one stream (a signal's training depends on its seed, stream number and data only), losses read an epoch late on both."""
    from hypad_amd import train as ht
    monkeypatch.chdir(tmp_path)
    counts = [2 * B + 1, 2 * B + 9, 3 * B + 4, 3 * B, 4 * B + 2, 5 * B + 3, 5 * B]
    data = [windows(n, 70 + i) for i, n in enumerate(counts)]
    names = [f"s{i}" for i in range(len(counts))]
    plan, _ = ht.plan_signal_groups(counts, B)
    assert len(plan) == 4 and max(len(m) for _, m in plan) == 2
    runs = []
    for lanes in (1, None):
        P = P_(11)
        P.lanes = lanes
        P.dataset = "lanes_%s" % lanes
        runs.append(ht.train_signals_resident(data, P, names=names, seed=3, init_seed=40, log=None))
    for n in names:
        assert runs[0][n]["history"] == runs[1][n]["history"], n
        assert runs[0][n]["history"]["repairs"] == 0 and runs[1][n]["history"]["repairs"] == 0, n      # co-resident lanes: no resident launch gave up
        assert len(runs[1][n]["history"]["dec"]) == 11 and np.isfinite(runs[1][n]["history"]["dec"]).all()
        for ma, mb in zip(runs[0][n]["modules"], runs[1][n]["modules"]):
            sa, sb = ma.state_dict(), mb.state_dict()
            assert all(torch.equal(sa[k], sb[k]) for k in sa), n
        fa, fb = sorted(os.listdir(runs[0][n]["path"])), sorted(os.listdir(runs[1][n]["path"]))
        assert fa == fb and "encoder_10.pt" in fa and "encoder.pt" in fa
        for f in fa:
            a = torch.load(os.path.join(runs[0][n]["path"], f), weights_only=False).state_dict()
            b = torch.load(os.path.join(runs[1][n]["path"], f), weights_only=False).state_dict()
            assert all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a), (n, f)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_worker(rank, world, port, cwd, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.chdir(cwd)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hypad_amd import train as ht
        data = [windows(n, i) for i, n in enumerate(COUNTS)]
        res = ht.train_signals_resident(data, P_(), names=[f"sig{i}" for i in range(len(COUNTS))], seed=77, init_seed=500, log=None)
        ret[rank] = {n: {"history": r["history"], "rank": r["rank"], "local": "modules" in r,
                         "enc": ({k: v.cpu() for k, v in r["modules"][0].state_dict().items()} if "modules" in r else None)} for n, r in res.items()}
    finally:
        dist.destroy_process_group()


def test_two_processes_own_disjoint_signals_and_agree_with_one(tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    from hypad_amd import train as ht
    (tmp_path / "two").mkdir(); (tmp_path / "one").mkdir()
    ret = mp.Manager().dict()
    mp.spawn(_rank_worker, args=(2, _free_port(), str(tmp_path / "two"), ret), nprocs=2, join=True)
    monkeypatch.chdir(tmp_path / "one")
    data = [windows(n, i) for i, n in enumerate(COUNTS)]
    names = [f"sig{i}" for i in range(len(COUNTS))]
    one = ht.train_signals_resident(data, P_(), names=names, seed=77, init_seed=500, log=None)
    owners = {n: ret[0][n]["rank"] for n in names}
    assert set(owners.values()) == {0, 1} and owners == {n: ret[1][n]["rank"] for n in names}
    for n in names:
        assert ret[0][n]["history"] == ret[1][n]["history"] == one[n]["history"], n           # gathered: every rank holds every signal's metrics
        assert ret[owners[n]][n]["local"] and not ret[1 - owners[n]][n]["local"]
        enc = ret[owners[n]][n]["enc"]
        for k, v in one[n]["modules"][0].state_dict().items():
            assert torch.equal(v.cpu(), enc[k]), (n, k)
        for sfx in ("", "_2"):                                                                  # the owner wrote the signal's checkpoints
            assert os.path.exists(tmp_path / "two" / "trained_models" / "models_hyper_T_3_0.0005" / "T" / n / f"encoder{sfx}.pt")


def _nccl_worker(rank, port, cwd, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.chdir(cwd)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from hypad_amd import train as ht
        data = [windows(n, i) for i, n in enumerate(COUNTS[:3])]
        res = ht.train_signals_resident(data, P_(epochs=2), names=["a", "b", "c"], seed=5, init_seed=9, log=None)
        ret["hist"] = {n: r["history"] for n, r in res.items()}
        ret["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_signal_metrics_gather_through_a_one_rank_rccl_group(tmp_path, monkeypatch):
    """The end-of-run gather of train_signals_resident (the only collective of multi-signal training) through RCCL: a world-size-1
    `nccl` group in a fresh process == no process group at all."""
    import torch.multiprocessing as mp
    from hypad_amd import train as ht
    (tmp_path / "nccl").mkdir(); (tmp_path / "plain").mkdir()
    ret = mp.Manager().dict()
    mp.spawn(_nccl_worker, args=(_free_port(), str(tmp_path / "nccl"), ret), nprocs=1, join=True)
    assert ret["backend"] == "nccl"
    monkeypatch.chdir(tmp_path / "plain")
    data = [windows(n, i) for i, n in enumerate(COUNTS[:3])]
    plain = ht.train_signals_resident(data, P_(epochs=2), names=["a", "b", "c"], seed=5, init_seed=9, log=None)
    assert {n: r["history"] for n, r in plain.items()} == dict(ret["hist"])


def test_run_signals_end_to_end_on_two_csv_signals(tmp_path, monkeypatch, capsys):
    """``python -m hypad_amd.main --config cfg.yaml --signals siga,sigb`` (main.run_signals; /root/reference/main.py:32-70 once per
    signal, train.py:428-437 directories): two synthetic CSV signals -> SignalDataset -> train_signals_resident (one group of two
    models) -> per signal the test loop, the scoring kernels, intervals and the overlap-segment counts -> metrics of both signals.
    Through the argument parser and through the function; the per-signal models equal train_resident on that signal alone."""
    import json
    import yaml
    from hypad_amd import main as hmain
    from hypad_amd import train as ht
    from hypad_amd.utils import data as od
    d = tmp_path / "data"
    d.mkdir()
    n = 700
    t0 = 1_400_000_000
    rows = []
    for k, name in enumerate(("siga", "sigb")):
        rng = np.random.default_rng(50 + k)
        tt = np.arange(n)
        v = np.sin(2 * np.pi * tt / (60.0 + 11 * k)) + 0.05 * rng.standard_normal(n)
        v[400:430] += 1.5
        with open(d / f"{name}.csv", "w") as f:
            f.write("timestamp,value\n" + "\n".join(f"{t0 + 600 * i},{x:.6f}" for i, x in zip(tt, v)) + "\n")
        rows.append('%s,"[[%d, %d]]"' % (name, t0 + 600 * 395, t0 + 600 * 435))
    with open(d / "anomalies.csv", "w") as f:
        f.write("signal,events\n" + "\n".join(rows) + "\n")
    cfg = dict(dataset="NAB", signal="siga", epochs=2, hyperbolic=True, signal_shape=100, lr=5e-4, batch_size=64, save_result=False, filename="",
               rec_error="dtw", combination="mult", interval=600, unique_dataset=True, resume=False, resume_epoch=0, load=False)
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(9)
    res = hmain.main(["--config", str(tmp_path / "cfg.yaml"), "--data-dir", str(d), "--signals", "siga, sigb"])
    out = capsys.readouterr().out
    assert sorted(res) == ["siga", "sigb"] and "siga" in out and "sigb" in out
    for name in ("siga", "sigb"):
        r = res[name]
        assert r["path"] == f"./trained_models/models_hyper_NAB_2_0.0005/NAB/{name}" and r["rank"] == 0
        assert len(r["confusion"]) == 4 and r["n_intervals"] >= 0 and np.isfinite(r["final"]["dec"])
        assert {"critic_x.pt", "critic_z.pt", "decoder.pt", "encoder.pt"} <= set(os.listdir(r["path"]))
        assert os.path.exists(os.path.join(r["path"], "recons_signal.pt"))          # the test loop's artefacts (anomaly_detection.py:116-131)
        json.dumps(r)                                                                # plain data: what all_gather_object carries
    # the function form with the same seed trains the same models, and each equals its single-signal run
    P = SimpleNamespace(**cfg)
    torch.manual_seed(9)
    again = hmain.run_signals(P, ["siga", "sigb"], None, str(d), log=lambda s_: None)
    assert {k: again[k]["final"] for k in again} == {k: res[k]["final"] for k in res}
    sets = []
    for name in ("siga", "sigb"):
        p = SimpleNamespace(**cfg); p.signal = name
        sets.append(od.dataset_selection(p, str(d))[0])
    torch.manual_seed(9)
    both = ht.train_signals_resident(sets, SimpleNamespace(**cfg), names=["siga", "sigb"], log=None, save=False)
    for i, name in enumerate(("siga", "sigb")):
        assert both[name]["final"] == res[name]["final"]
        p = SimpleNamespace(**cfg); p.signal = name + "_alone"
        torch.manual_seed(9 + i)
        enc, dec, cx, cz, _, hist = ht.train_resident(sets[i], p, seed=9, log=None, first_signal=both[name]["stream"])
        assert hist.dec == both[name]["history"]["dec"] and hist.hyper == both[name]["history"]["hyper"]
        saved = torch.load(os.path.join(res[name]["path"], "decoder.pt"), weights_only=False)
        for k, v in dec.state_dict().items():
            assert torch.equal(v.cpu(), saved.state_dict()[k].cpu()), (name, k)
