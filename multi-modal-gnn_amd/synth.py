"""Synthetic hetero-graphs of the eICU shape for benchmarks and scale tests.

scale s: N_P = 1,834*s patients; vocab 50 labs / 114 diagnoses / 100 medications (fixed);
exactly 61,484*s has_lab, 5,421*s has_diagnosis, 15,933*s has_medication edges; no duplicate
(patient, item) pair; has_lab degree ~ clipped Normal(33.5, 12) with ~1.5 % of patients forced to 0..5
labs (exercises the degree gate of src/model.py:312-315 and empty neighbourhoods); Zipf-like item
popularity; edge order lab-major for has_lab (what the reference's parquet yields,
src/preprocess.py:141-147) and patient-major for the other two; edge_attr ~ N(0,1).
Runs on any torch device (generation on the GPU at x100/x1000 takes seconds).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .data import HeteroGraph

EICU = dict(patients=1834, labs=50, dx=114, meds=100, e_lab=61484, e_dx=5421, e_med=15933)
MIMIC_LIKE = dict(patients=1834, labs=50, dx=200, meds=100, e_lab=61484, e_dx=5421, e_med=15933)


def _degrees(n, mean, std, lo, hi, total, gen, device, frozen=None):
    d = (torch.randn(n, generator=gen, device=device) * std + mean).round().clamp_(lo, hi).long()
    if frozen is not None:
        d = torch.where(frozen >= 0, frozen, d)
    free = torch.ones(n, dtype=torch.bool, device=device) if frozen is None else frozen < 0
    for _ in range(64):                       # nudge +-1 on random free rows until the total is exact
        diff = int(total - int(d.sum()))
        if diff == 0:
            break
        ok = free & ((d < hi) if diff > 0 else (d > lo))
        idx = torch.nonzero(ok).flatten()
        if idx.numel() == 0:
            raise RuntimeError("cannot reach the requested edge total")
        k = min(abs(diff), idx.numel())
        pick = idx[torch.randperm(idx.numel(), generator=gen, device=device)[:k]]
        d[pick] += 1 if diff > 0 else -1
    if int(d.sum()) != total:
        raise RuntimeError("edge total not reached")
    return d


def _poisson_degrees(n, lam, hi, total, gen, device):
    d = torch.poisson(torch.full((n,), float(lam), device=device), generator=gen).clamp_(0, hi).long()
    for _ in range(64):
        diff = int(total - int(d.sum()))
        if diff == 0:
            break
        ok = (d < hi) if diff > 0 else (d > 0)
        idx = torch.nonzero(ok).flatten()
        k = min(abs(diff), idx.numel())
        pick = idx[torch.randperm(idx.numel(), generator=gen, device=device)[:k]]
        d[pick] += 1 if diff > 0 else -1
    if int(d.sum()) != total:
        raise RuntimeError("edge total not reached")
    return d


def _pick_items(deg, n_items, zipf_a, gen, device, chunk=1 << 18):
    """deg[i] distinct items per row, weighted by popularity (Gumbel top-k). -> (rows, items), row-major."""
    w = 1.0 / torch.arange(1, n_items + 1, device=device, dtype=torch.float32) ** zipf_a
    logw = torch.log(w / w.sum())
    rows, items = [], []
    for s in range(0, deg.numel(), chunk):
        d = deg[s:s + chunk]
        u = torch.rand(d.numel(), n_items, generator=gen, device=device).clamp_(1e-12, 1 - 1e-7)
        score = logw - torch.log(-torch.log(u))
        rank = score.argsort(dim=1, descending=True).argsort(dim=1)
        r, c = torch.nonzero(rank < d[:, None], as_tuple=True)
        rows.append(r + s)
        items.append(c)
    return torch.cat(rows), torch.cat(items)


def make_graph(scale: int = 1, seed: int = 0, device="cpu", shape: Dict = EICU,
               with_reverse: bool = True) -> HeteroGraph:
    device = torch.device(device)
    gen = torch.Generator(device=device).manual_seed(seed)
    P = shape["patients"] * scale
    L, DX, M = shape["labs"], shape["dx"], shape["meds"]
    # ~1.5 % low-connectivity patients with 0..5 labs
    frozen = torch.full((P,), -1, dtype=torch.long, device=device)
    low = torch.rand(P, generator=gen, device=device) < 0.015
    frozen[low] = torch.randint(0, 6, (int(low.sum()),), generator=gen, device=device)
    d_lab = _degrees(P, 33.5, 12.0, 0, L, shape["e_lab"] * scale, gen, device, frozen)
    d_dx = _poisson_degrees(P, shape["e_dx"] / shape["patients"], DX, shape["e_dx"] * scale, gen, device)
    d_med = _poisson_degrees(P, shape["e_med"] / shape["patients"], M, shape["e_med"] * scale, gen, device)

    p_lab, i_lab = _pick_items(d_lab, L, 0.35, gen, device)
    order = torch.sort(i_lab, stable=True).indices            # lab-major, patients ascending inside a lab
    p_lab, i_lab = p_lab[order], i_lab[order]
    p_dx, i_dx = _pick_items(d_dx, DX, 0.8, gen, device)
    p_med, i_med = _pick_items(d_med, M, 0.6, gen, device)

    g = HeteroGraph()
    g["patient"].num_nodes = P
    g["lab"].num_nodes = L
    g["diagnosis"].num_nodes = DX
    g["medication"].num_nodes = M
    ei = torch.stack([p_lab, i_lab]).contiguous()
    ea = torch.randn(ei.shape[1], 1, generator=gen, device=device)
    g["patient", "has_lab", "lab"].edge_index = ei
    g["patient", "has_lab", "lab"].edge_attr = ea
    if with_reverse:
        g["lab", "has_lab_rev", "patient"].edge_index = ei.flip(0).contiguous()
        g["lab", "has_lab_rev", "patient"].edge_attr = ea
    ei = torch.stack([p_dx, i_dx]).contiguous()
    g["patient", "has_diagnosis", "diagnosis"].edge_index = ei
    if with_reverse:
        g["diagnosis", "has_diagnosis_rev", "patient"].edge_index = ei.flip(0).contiguous()
    ei = torch.stack([p_med, i_med]).contiguous()
    g["patient", "has_medication", "medication"].edge_index = ei
    if with_reverse:
        g["medication", "has_medication_rev", "patient"].edge_index = ei.flip(0).contiguous()
    return g


def directed_edges(g) -> int:
    return sum(int(g[et].edge_index.shape[1]) for et in g.edge_types)
