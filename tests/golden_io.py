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


def rel_close(a, b, rtol=1e-4, atol_frac=1e-6):
    """Element-wise  |a - b| <= rtol * |b| + atol  with  atol = atol_frac * max|b|  -- north_star's "within 1e-4 relative
    fp32" per VALUE (small predictions are not hidden behind the largest one); the absolute floor only covers entries that
    are rounding noise at the tensor's scale.  -> (ok, worst excess ratio)."""
    a, b = a.double(), b.double()
    atol = atol_frac * float(b.abs().max().clamp_min(1e-30))
    ratio = (a - b).abs() / (rtol * b.abs() + atol)
    worst = float(ratio.max()) if ratio.numel() else 0.0
    return worst <= 1.0, worst
