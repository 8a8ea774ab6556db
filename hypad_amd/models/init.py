"""Initial weights of ``nn.LSTM`` / ``nn.Linear`` without building the modules.

The reference builds its networks from ``nn.LSTM`` and ``nn.Linear`` (models/tadgan.py:15-21,34-41,77-89,113-119) and trains whatever
their default initialisation drew from torch's global CPU generator.  hypad_amd's networks keep their weights in one flat arena
(arena.py) and need those modules only for the numbers: constructing them costs 3-7 ms per network (``Module.__init__``, parameter
registration, ``flatten_parameters``) -- 49 ms per model, 1.6 s of set-up for 32 models in ``train_signals_resident``.  The functions
below make the SAME draws -- the same ``uniform_`` calls on tensors of the same shapes in the same order, so the generator is left in
the same state -- and return them under the modules' ``state_dict`` names.  Pinned against torch itself by
tests/test_fast_init.py (bit for bit, several seeds and shapes)."""
import math

import torch


def linear_init(in_features, out_features):
    """``nn.Linear(in_features, out_features)``: kaiming_uniform_(weight, a=sqrt(5)), then bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    (torch/nn/modules/linear.py reset_parameters; init.py kaiming_uniform_ / calculate_gain, spelled out in the same floating-point steps)."""
    w = torch.empty(out_features, in_features)
    gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
    std = gain / math.sqrt(in_features)
    bound = math.sqrt(3.0) * std
    w.uniform_(-bound, bound)
    b = torch.empty(out_features)
    bb = 1 / math.sqrt(in_features) if in_features > 0 else 0
    b.uniform_(-bb, bb)
    return {"weight": w, "bias": b}


def lstm_init(input_size, hidden_size, num_layers=1, bidirectional=True):
    """``nn.LSTM(input_size, hidden_size, num_layers, bidirectional=...)``: every tensor ~ U(-1/sqrt(hidden), 1/sqrt(hidden)), drawn in
    registration order -- layer, direction, (weight_ih, weight_hh, bias_ih, bias_hh) (torch/nn/modules/rnn.py RNNBase)."""
    stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
    out = {}
    dirs = 2 if bidirectional else 1
    for layer in range(num_layers):
        lin = input_size if layer == 0 else hidden_size * dirs
        for d in range(dirs):
            sfx = "_reverse" if d == 1 else ""
            for name, shape in ((f"weight_ih_l{layer}{sfx}", (4 * hidden_size, lin)), (f"weight_hh_l{layer}{sfx}", (4 * hidden_size, hidden_size)),
                                (f"bias_ih_l{layer}{sfx}", (4 * hidden_size,)), (f"bias_hh_l{layer}{sfx}", (4 * hidden_size,))):
                out[name] = torch.empty(*shape).uniform_(-stdv, stdv)
    return out
