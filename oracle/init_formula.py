"""ORACLE - TEST INFRASTRUCTURE ONLY (never imported by the product path).

Deterministic, torch-RNG-free parameter formula shared by tests/golden/make_golden.py
(which applies it to the reference's RandLANet state_dict) and by the tests (which apply
it to the build's state_dict and to the oracle restatement), so golden fixtures need to
store only inputs and outputs, never weights.

Keys are visited in state_dict order -- the order of the reference's
randlanet.utils.modules.RandLANet (modules.py:494-530), committed as
tests/golden/state_dict_*.json.
"""
from collections import OrderedDict

import numpy as np
import torch


def formula_state_dict(items, seed=1234):
    """items: iterable of (key, shape) in state_dict order -> OrderedDict of fp32 tensors."""
    rs = np.random.RandomState(seed)
    out = OrderedDict()
    for key, shape in items:
        shape = tuple(int(s) for s in shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        is_bn = ".batch_norm." in key or key.startswith("bn_start.")
        if leaf == "running_var":
            a = rs.uniform(0.8, 1.2, shape)
        elif leaf == "running_mean":
            a = rs.uniform(-0.1, 0.1, shape)
        elif is_bn and leaf == "weight":
            a = rs.uniform(0.8, 1.2, shape)
        elif leaf == "bias":
            a = rs.uniform(-0.1, 0.1, shape)
        else:
            # conv (Cout,Cin,1,1) / linear (out,in): fan_in = shape[1];
            # decoder ConvTranspose2d (Cin,Cout,1,1) (modules.py:512-523): fan_in = shape[0]
            fan_in = shape[0] if (key.startswith("decoder.") and leaf == "weight") else shape[1]
            a = rs.uniform(-1.0, 1.0, shape) * np.sqrt(3.0 / fan_in)
        out[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out
