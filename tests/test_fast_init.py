"""hypad_amd/models/init.py makes the draws of nn.LSTM / nn.Linear without building the modules: same values, same final generator
state -- pinned against torch's own modules, and the four networks against a construction that does build them (the reference's
construction order, models/tadgan.py:11-21,31-56,71-89,110-121)."""
import pytest
import torch
from torch import nn

from hypad_amd.models import init as hi


@pytest.mark.parametrize("seed", [0, 7, 123456])
@pytest.mark.parametrize("a,b", [(100, 20), (20, 50), (128, 100), (20, 1), (150, 20), (3, 5), (20, 20)])
def test_linear_init_equals_nn_linear(seed, a, b):
    torch.manual_seed(seed)
    ref = nn.Linear(a, b)
    s_ref = torch.get_rng_state()
    torch.manual_seed(seed)
    got = hi.linear_init(a, b)
    assert torch.equal(got["weight"], ref.weight.detach()) and torch.equal(got["bias"], ref.bias.detach())
    assert torch.equal(torch.get_rng_state(), s_ref)


@pytest.mark.parametrize("seed", [0, 11])
@pytest.mark.parametrize("inp,hid,layers", [(100, 50, 1), (50, 64, 2), (150, 50, 1), (123, 50, 1), (7, 3, 3)])
def test_lstm_init_equals_nn_lstm(seed, inp, hid, layers):
    torch.manual_seed(seed)
    ref = nn.LSTM(input_size=inp, hidden_size=hid, num_layers=layers, dropout=0.2 if layers > 1 else 0.0, bidirectional=True)
    s_ref = torch.get_rng_state()
    torch.manual_seed(seed)
    got = hi.lstm_init(inp, hid, layers, True)
    sd = ref.state_dict()
    assert list(got) == list(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    assert torch.equal(torch.get_rng_state(), s_ref)


@pytest.mark.parametrize("S,hyper", [(100, True), (100, False), (150, True)])
def test_networks_start_from_the_weights_the_reference_construction_draws(S, hyper):
    """One manual_seed, the four networks in train.py:415-426's order: hypad_amd's modules against the oracle's (which build nn.LSTM /
    nn.Linear as the reference does)."""
    from hypad_amd.models import tadgan
    from oracle import tadgan as ot
    torch.manual_seed(3)
    mine = [tadgan.Encoder(S, 20), tadgan.Decoder(S, 20, hyper), tadgan.CriticX(S, 20), tadgan.CriticZ(20)]
    s_mine = torch.get_rng_state()
    torch.manual_seed(3)
    ref = [ot.Encoder(S, 20), ot.Decoder(S, 20, hyper), ot.CriticX(S, 20), ot.CriticZ(20)]
    assert torch.equal(torch.get_rng_state(), s_mine)
    for m, r in zip(mine, ref):
        sm, sr = m.state_dict(), r.state_dict()
        assert list(sm) == list(sr)
        for k in sr:
            assert torch.equal(sm[k], sr[k]), k


def test_one_host_thread_is_reentrant_and_restores_the_setting():
    """overlapping uses (nested, or from two threads) must leave torch's intra-op thread count as they found it."""
    import threading
    import torch
    from hypad_amd import train as ht
    before = torch.get_num_threads()
    torch.set_num_threads(max(2, before))
    want = torch.get_num_threads()
    try:
        with ht._one_host_thread():
            assert torch.get_num_threads() == 1
            with ht._one_host_thread():
                assert torch.get_num_threads() == 1
            assert torch.get_num_threads() == 1                   # the inner exit does not restore under the outer one
        assert torch.get_num_threads() == want
        gate_in, gate_out, seen = threading.Event(), threading.Event(), []

        def other():
            with ht._one_host_thread():                           # enters while the main thread's use is in flight, leaves after it
                seen.append(torch.get_num_threads())
                gate_in.set()
                gate_out.wait(10)
            seen.append(torch.get_num_threads())
        t = threading.Thread(target=other)
        with ht._one_host_thread():
            t.start()
            assert gate_in.wait(10)
        assert torch.get_num_threads() == want                    # this thread's outermost use ended: restored, whatever the other thread does
        gate_out.set()
        t.join()
        assert torch.get_num_threads() == want and seen == [1, want]      # the late one went back to the value taken by the FIRST use, not to 1
    finally:
        torch.set_num_threads(before)
