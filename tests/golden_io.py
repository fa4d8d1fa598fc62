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


def grad_close(a, b, tie=None, rtol=2e-4, row_frac=1e-4, floor=0.0):
    """Gradient check per ELEMENT:  |a - b| <= rtol * |b| + row_frac * max|b[row]| + floor (+ tie[i]).
    floor: an ABSOLUTE allowance, 1e-6 of the largest gradient of the whole model in the callers -- a parameter whose true
    gradient is 0 (a bias in front of a BatchNorm: the mean subtraction cancels it) carries pure rounding noise on both
    sides, which only a scale from outside the tensor can bound.
    The absolute part scales with the element's own ROW (a vector is one row), so a row of small magnitude -- the
    embedding gradient of a patient with one supervised pair next to one with fifty -- has to be right at its own scale;
    the max-norm bar (2e-4 * max|b| everywhere) lets such a row be wholly wrong.  tie (same shape, >= 0): extra slack
    for elements behind a ReLU input that sits within rounding of 0 (see _train_step_vs_oracle).
    -> (ok, worst ratio, number of elements that pass ONLY thanks to the tie slack)."""
    a, b = a.double(), b.double()
    b2 = b.reshape(1, -1) if b.dim() < 2 else b.reshape(b.shape[0], -1)
    a2 = a.reshape(b2.shape)
    rowmax = b2.abs().amax(dim=1, keepdim=True) if b2.numel() else b2.abs()
    base = rtol * b2.abs() + row_frac * rowmax + float(floor) if b2.numel() else b2
    tol = base if tie is None else base + tie.double().reshape(b2.shape)
    ratio = (a2 - b2).abs() / tol.clamp_min(1e-300)
    worst = float(ratio.max()) if ratio.numel() else 0.0
    n_tied = 0 if tie is None else int(((a2 - b2).abs() > base).sum())
    return worst <= 1.0, worst, n_tied
