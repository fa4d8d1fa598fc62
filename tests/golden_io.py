"""Reader for tests/golden/*.npz (written by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    keys = json.loads(str(z["__keys__"]))
    meta = json.loads(str(z["__meta__"]))
    return {k: z[f"t{i}"] for i, k in enumerate(keys)}, meta


def t(a):
    return torch.from_numpy(np.asarray(a))


def unpack_mask(packed, n):
    return torch.from_numpy(np.unpackbits(packed)[:n].astype(bool))


def checksum(x):
    x = x.detach().double()
    return torch.stack([x.sum(), x.abs().sum(), (x * x).sum()])


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
