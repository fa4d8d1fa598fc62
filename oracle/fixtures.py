"""Closed-form (RNG-free) test inputs shared by gen_golden.py and the tests (TEST INFRASTRUCTURE).

Inputs are pure integer-hash functions of their indices, so they are bit-identical on
any machine / torch version; the golden fixtures therefore only need to store the
reference's OUTPUTS for them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from .model import mangle

NODE_TYPES = ["patient", "lab", "diagnosis", "medication"]          # graph_build.py:186-201
EDGE_TYPES = [                                                       # graph_build.py:216-247
    ("patient", "has_lab", "lab"), ("lab", "has_lab_rev", "patient"),
    ("patient", "has_diagnosis", "diagnosis"), ("diagnosis", "has_diagnosis_rev", "patient"),
    ("patient", "has_medication", "medication"), ("medication", "has_medication_rev", "patient"),
]


def _hash_u32(*idx: np.ndarray, salt: int) -> np.ndarray:
    h = np.full(np.broadcast(*idx).shape, (salt * 0x9E3779B1) & 0xFFFFFFFF, dtype=np.uint64)
    for k, a in enumerate(idx):
        h = (h ^ (a.astype(np.uint64) + np.uint64(0x7F4A7C15 + 0x1000193 * k))) * np.uint64(0x01000193)
        h &= np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(15)
        h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(12)
    return h.astype(np.uint32)


def det_uniform(shape: Sequence[int], salt: int, lo=-1.0, hi=1.0) -> torch.Tensor:
    """Deterministic pseudo-uniform fp32 tensor in [lo, hi): 20-bit hash / 2^20."""
    grids = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij") if len(shape) else [np.zeros((), np.int64)]
    u = (_hash_u32(*grids, salt=salt) >> np.uint32(12)).astype(np.float64) / float(1 << 20)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(tuple(shape)))


def det_state(num_nodes: Dict[str, int], hidden_dim=128, num_layers=2,
              edge_types=EDGE_TYPES) -> Dict[str, torch.Tensor]:
    """A full state_dict in the reference layout with non-trivial BN buffers."""
    D = hidden_dim
    sd: Dict[str, torch.Tensor] = {}
    salt = [100]

    def nxt():
        salt[0] += 1
        return salt[0]

    def lin(prefix, o, i, bias=True):
        b = 1.0 / math.sqrt(i)
        sd[prefix + ".weight"] = det_uniform((o, i), nxt(), -b, b)
        if bias:
            sd[prefix + ".bias"] = det_uniform((o,), nxt(), -b, b)

    def bn(prefix):
        sd[prefix + ".weight"] = det_uniform((D,), nxt(), 0.5, 1.5)
        sd[prefix + ".bias"] = det_uniform((D,), nxt(), -0.3, 0.3)
        sd[prefix + ".running_mean"] = det_uniform((D,), nxt(), -0.2, 0.2)
        sd[prefix + ".running_var"] = det_uniform((D,), nxt(), 0.5, 1.5)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(3, dtype=torch.long)

    lin("patient_transform.0", D, D); bn("patient_transform.1")
    lin("patient_transform.4", D, D); bn("patient_transform.5")
    lin("patient_transform.8", D, D)
    for l in range(num_layers):
        for et in edge_types:
            pre = f"convs.{l}.convs.{mangle(tuple(et))}"
            lin(pre + ".lin_l", D, D, True)
            lin(pre + ".lin_r", D, D, False)
    for l in range(num_layers):
        for t in num_nodes:
            bn(f"batch_norms.{l}.{t}")
    for head in ("edge_predictor", "tabular_mlp"):
        lin(f"{head}.mlp.0", 64, 2 * D)
        lin(f"{head}.mlp.3", 32, 64)
        lin(f"{head}.mlp.6", 1, 32)
    for t, n in num_nodes.items():
        b = math.sqrt(6.0 / (n + D))
        sd[f"embeddings.{t}.weight"] = det_uniform((n, D), nxt(), -b, b)
    return sd


def det_frames(n_pat: int, n_lab: int, n_dx: int, n_med: int, *, lab_density=0.67,
               dx_density=0.026, med_density=0.087, salt=7):
    """Closed-form EHR-like frames in the reference's column layout and row order
    (labs lab-major as preprocess.py:141-147 yields; dx/med patient-major).

    ~1 % of patients have no edge at all, ~2 % have < 6 labs (exercise the degree gate,
    model.py:312-315, and the empty-neighbourhood case).  Returns plain python/numpy columns:
      cohort_ids, (lab_sid, lab_item, lab_val), (dx_sid, dx_code), (med_sid, med_drug)
    """
    pid = 10000 + 3 * np.arange(n_pat)                 # non-contiguous SUBJECT_IDs
    item = 50800 + 7 * np.arange(n_lab)                # ITEMIDs
    P = np.arange(n_pat)
    isolated = (P % 97) == 5
    sparse = (P % 53) == 7

    def incidence(n_other, density, s, popularity=True):
        i, j = np.meshgrid(P, np.arange(n_other), indexing="ij")
        thr = density * (1.6 - 1.2 * j / max(n_other - 1, 1)) if popularity else density
        m = (_hash_u32(i, j, salt=s) >> np.uint32(8)) % np.uint32(10000) < (np.clip(thr, 0, 1) * 10000).astype(np.uint32)
        m[isolated] = False
        return m

    a_lab = incidence(n_lab, lab_density, salt + 1)
    a_lab[sparse, 3:] = False                          # at most 3 labs -> tabular head
    a_dx = incidence(n_dx, dx_density, salt + 2, popularity=False)
    a_med = incidence(n_med, med_density, salt + 3, popularity=False)

    jj, ii = np.nonzero(a_lab.T)                       # lab-major row order
    val = ((_hash_u32(ii, jj, salt=salt + 4) >> np.uint32(10)) % np.uint32(4001)).astype(np.float64) / 1000.0 - 2.0
    labs = (pid[ii], item[jj], val)
    ii, jj = np.nonzero(a_dx)
    dx = (pid[ii], np.array([f"{300 + 3 * j}" if j % 4 else f"V{10 + j}" for j in jj], dtype=object))
    ii, jj = np.nonzero(a_med)
    med = (pid[ii], np.array([f"drug_{j:03d}" for j in jj], dtype=object))
    return pid, labs, dx, med


def graph_from_frames(frames):
    from .graph import build_graph
    pid, labs, dx, med = frames
    return build_graph(list(pid), (list(labs[0]), list(labs[1]), list(labs[2])),
                       (list(dx[0]), list(dx[1])), (list(med[0]), list(med[1])))


def det_masks(names_shapes: Dict[str, Tuple[int, ...]], p: float, salt=900) -> Dict[str, torch.Tensor]:
    """Deterministic dropout keep-masks (1 = keep) for injected-mask parity runs."""
    out = {}
    for k, (name, shape) in enumerate(sorted(names_shapes.items())):
        out[name] = (det_uniform(shape, salt + k, 0.0, 1.0) >= p).to(torch.float32)
    return out
