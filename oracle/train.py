"""CPU restatement of the caller-side arithmetic around the hot path (TEST INFRASTRUCTURE).

Follows /root/reference/src/train.py and evaluate.py:
  * edge splits               train.py:98-129   (manual_seed(seed); randperm(E); int() boundaries)
  * supervision mask          train.py:150-176  (rand(n_train) < mask_fraction; injectable generator
                                                 instead of the wall-clock seed of :156)
  * per-lab loss weights      train.py:295-330
  * weighted MAE/MSE step     train.py:366-386
  * regression metrics        evaluate.py:36-82 ; per-lab +-3 sigma winsorisation :417-440
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import model as om


def edge_splits(num_edges: int, train=0.7, val=0.15, test=0.15, seed=42):
    assert abs(train + val + test - 1.0) < 1e-6, "Splits must sum to 1.0"
    torch.manual_seed(seed)
    np.random.seed(seed)
    perm = torch.randperm(num_edges)
    n_tr, n_va = int(train * num_edges), int(val * num_edges)
    masks = [torch.zeros(num_edges, dtype=torch.bool) for _ in range(3)]
    masks[0][perm[:n_tr]] = True
    masks[1][perm[n_tr:n_tr + n_va]] = True
    masks[2][perm[n_tr + n_va:]] = True
    return tuple(masks)


def supervision_mask(n_train: int, mask_fraction: float, generator: Optional[torch.Generator] = None):
    if mask_fraction > 0:
        return torch.rand(n_train, generator=generator) < mask_fraction
    return torch.ones(n_train, dtype=torch.bool)


def lab_weights(lab_indices: torch.Tensor, values: torch.Tensor, num_labs: int, eps=1e-6):
    var = torch.zeros(num_labs, dtype=values.dtype)
    for j in range(num_labs):
        m = lab_indices == j
        var[j] = values[m].var() if int(m.sum()) > 1 else 1.0
    w = 1.0 / (var + eps)
    return w * num_labs / w.sum()


def weighted_loss(pred, target, lab_idx, weights, sup_mask, loss_type="mae"):
    p, t = pred[sup_mask], target[sup_mask]
    if loss_type == "mae":
        per = (p - t).abs()
    elif loss_type == "mse":
        per = (p - t) ** 2
    else:  # train.py:378-383 falls back to the unweighted loss
        return om.compute_regression_loss(p, t, loss_type)
    return (per * weights[lab_idx[sup_mask]]).mean()


def train_step_grads(sd, g, pi, li, target, weights, sup_mask, *, p=0.0, masks=None, loss_type="mae",
                     num_layers=2, degree_threshold=6, use_batch_norm=True, activation="relu"):
    """One fwd + weighted loss + bwd (train.py:347-392 minus optimizer.step).
    Returns (loss, pred, grads dict over every floating parameter incl. embeddings, bufs)."""
    leaf = {k: (v.detach().clone().requires_grad_(True)
                if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
            for k, v in sd.items()}
    pred, bufs = om.predict_lab_values(leaf, g, pi, li, num_layers=num_layers, training=True, p=p, masks=masks,
                                       degree_threshold=degree_threshold, use_batch_norm=use_batch_norm,
                                       activation=activation)
    loss = weighted_loss(pred, target, li, weights, sup_mask, loss_type)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v))
             for k, v in leaf.items() if torch.is_tensor(v) and v.requires_grad}
    return loss.detach(), pred.detach(), grads, bufs


# --------------------------------------------------------------------------------------
# evaluate.py
# --------------------------------------------------------------------------------------
def regression_metrics(pred: np.ndarray, target: np.ndarray) -> Dict[str, float]:
    err = pred - target
    mae = float(np.mean(np.abs(err)))
    mse = float(np.mean(err ** 2))
    ss_res = float(np.sum((target - pred) ** 2))
    ss_tot = float(np.sum((target - target.mean()) ** 2))
    r2 = 1.0 - ss_res / ss_tot if ss_tot > 0 else (1.0 if ss_res == 0 else 0.0)
    nz = target != 0
    mape = float(np.mean(np.abs((target[nz] - pred[nz]) / target[nz])) * 100) if nz.sum() > 0 else float("nan")
    return {"mae": mae, "rmse": float(np.sqrt(mse)), "r2": float(r2), "mape": mape}


def winsorise_per_lab(pred: np.ndarray, target: np.ndarray, lab_idx: np.ndarray) -> Tuple[np.ndarray, int]:
    pred = pred.copy()
    res = pred - target
    capped = 0
    for j in np.unique(lab_idx):
        m = lab_idx == j
        r = res[m]
        if len(r) > 1:
            mu, sd = np.mean(r), np.std(r)
            rc = np.clip(r, mu - 3 * sd, mu + 3 * sd)
            capped += int(np.sum(rc != r))
            pred[m] = target[m] + rc
    return pred, capped
