"""Evaluation reducers around ``HeteroRGCN.predict_lab_values`` -- the caller on the far side of the hot path
(reference ``src/evaluate.py``; SURVEY.md section 8 "next" row f4).  Same function names, argument meaning, result keys
and files written as the reference:

  compute_regression_metrics  :36-86     MAE / RMSE / R^2 / MAPE(non-zero targets) in float64
  compute_per_lab_metrics     :89-141    one row per lab with >= 2 samples, sorted by MAE
  GlobalMeanBaseline / PerLabMeanBaseline / evaluate_baselines  :152-230
  stratify_by_patient_degree  :237-287   1-5 / 6-15 / 16+ observed labs
  stratify_by_lab_frequency   :290-342   quartiles of the non-zero lab counts
  evaluate_model              :349-570   predict -> per-lab +-3 sigma winsorisation -> metrics -> json / csv

The predictions come from the HIP path (one gather / head launch over all test pairs).  On a HIP device ``evaluate_model``
reduces them there (``device_*`` below: every reported figure is a segment sum over the pairs -- mmg_seg_moments /
mmg_seg_metrics -- so only [segments, 8] doubles leave the device instead of the predictions); the host reducers below are
the same arithmetic in numpy, the checker of the device path (tests/test_evaluate_gpu.py).  The host reducers are O(pairs)
float64 host arithmetic exactly as in the reference (sklearn's MAE/MSE/R^2 are restated in closed form: sklearn is
not a dependency of this package).  The per-lab loops are segment reductions over a stable sort instead of one boolean
mask per lab, which keeps the x1000 scale (92 M test pairs) linear.  Golden parity: tests/test_evaluate_cpu.py.
"""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Dict, Tuple

import numpy as np
import pandas as pd
import torch


def compute_regression_metrics(predictions: np.ndarray, targets: np.ndarray) -> Dict[str, float]:
    predictions = np.asarray(predictions)
    targets = np.asarray(targets)
    err = targets - predictions
    mae = float(np.average(np.abs(err), axis=0))
    mse = float(np.average(err ** 2, axis=0))        # sklearn hands back a Python float: the sqrt below is float64
    # r2_score: 1 - SS_res/SS_tot in the INPUT dtype (sklearn >= 1.5 sums fp32 inputs in fp32); a zero SS_res gives 1.0,
    # a constant target with a non-zero SS_res gives 0.0 (sklearn's force_finite)
    ss_res = np.sum(err ** 2, axis=0)
    ss_tot = np.sum((targets - np.average(targets, axis=0)) ** 2, axis=0)
    if ss_res == 0:
        r2 = 1.0
    elif ss_tot != 0:
        r2 = 1 - ss_res / ss_tot
    else:
        r2 = 0.0
    nz = targets != 0
    if nz.sum() > 0:
        mape = np.mean(np.abs((targets[nz] - predictions[nz]) / targets[nz])) * 100
    else:
        mape = np.nan
    return {"mae": float(mae), "rmse": float(np.sqrt(mse)), "r2": float(r2), "mape": float(mape)}


def _segments(keys: np.ndarray):
    """Stable sort by key -> (order, unique keys, segment starts, segment ends).  Within a segment the original
    element order is kept, so per-segment sums see the same operand order as ``x[keys == k]``."""
    order = np.argsort(keys, kind="stable")
    sk = keys[order]
    if len(sk) == 0:
        return order, sk, np.empty(0, np.int64), np.empty(0, np.int64)
    starts = np.flatnonzero(np.concatenate(([True], sk[1:] != sk[:-1])))
    ends = np.concatenate((starts[1:], [len(sk)]))
    return order, sk[starts], starts, ends


def compute_per_lab_metrics(predictions, targets, lab_indices, lab_names: Dict[int, str]) -> pd.DataFrame:
    order, labs, starts, ends = _segments(np.asarray(lab_indices))
    p, t = np.asarray(predictions)[order], np.asarray(targets)[order]
    results = []
    for lab_idx, a, b in zip(labs.tolist(), starts.tolist(), ends.tolist()):
        if b - a < 2:
            continue
        m = compute_regression_metrics(p[a:b], t[a:b])
        m["lab_index"] = int(lab_idx)
        m["lab_name"] = lab_names.get(int(lab_idx), f"Lab_{lab_idx}")
        m["num_samples"] = int(b - a)
        results.append(m)
    df = pd.DataFrame(results)
    return df.sort_values("mae")


def winsorize_residuals(predictions: np.ndarray, targets: np.ndarray, lab_indices: np.ndarray,
                        n_sigma: float = 3.0) -> Tuple[np.ndarray, int]:
    """evaluate.py:417-440: clip every lab's residuals to mean +- 3 std (labs with > 1 sample); returns the adjusted
    predictions (``targets + clipped residual``, in the predictions' dtype) and the number of capped residuals."""
    predictions = np.array(predictions, copy=True)
    residuals = predictions - targets
    order, _, starts, ends = _segments(np.asarray(lab_indices))
    num_capped = 0
    for a, b in zip(starts.tolist(), ends.tolist()):
        if b - a > 1:
            sel = order[a:b]
            r = residuals[sel]
            sd, mu = np.std(r), np.mean(r)
            rc = np.clip(r, mu - n_sigma * sd, mu + n_sigma * sd)
            num_capped += int(np.sum(r != rc))
            predictions[sel] = targets[sel] + rc
    return predictions, num_capped


class GlobalMeanBaseline:
    def __init__(self):
        self.mean = None

    def fit(self, values: np.ndarray):
        self.mean = np.mean(values)

    def predict(self, n: int) -> np.ndarray:
        return np.full(n, self.mean)


class PerLabMeanBaseline:
    def __init__(self):
        self.lab_means = {}

    def fit(self, values: np.ndarray, lab_indices: np.ndarray):
        order, labs, starts, ends = _segments(np.asarray(lab_indices))
        v = np.asarray(values)[order]
        for k, a, b in zip(labs.tolist(), starts.tolist(), ends.tolist()):
            self.lab_means[k] = v[a:b].mean()
        self.global_mean = np.mean(values)

    def predict(self, lab_indices: np.ndarray) -> np.ndarray:
        return np.array([self.lab_means.get(i, self.global_mean) for i in np.asarray(lab_indices).tolist()])


def evaluate_baselines(train_data, test_data) -> Dict[str, Dict[str, float]]:
    train_values, train_lab_indices = train_data
    test_values, test_lab_indices, _ = test_data
    results = {}
    g = GlobalMeanBaseline()
    g.fit(train_values)
    results["global_mean"] = compute_regression_metrics(g.predict(len(test_values)), test_values)
    pl = PerLabMeanBaseline()
    pl.fit(train_values, train_lab_indices)
    results["per_lab_mean"] = compute_regression_metrics(pl.predict(test_lab_indices), test_values)
    return results


def _groups(groups, predictions, targets):
    out = {}
    for name, mask in groups.items():
        if mask.sum() > 0:
            m = compute_regression_metrics(predictions[mask], targets[mask])
            m["num_samples"] = int(mask.sum())
            out[name] = m
    return out


def stratify_by_patient_degree(predictions, targets, patient_indices, graph) -> Dict[str, Dict]:
    ei = graph["patient", "has_lab", "lab"].edge_index
    degrees = torch.bincount(ei[0], minlength=graph["patient"].num_nodes).cpu().numpy()
    d = degrees[patient_indices]
    return _groups({"low (1-5 labs)": (d >= 1) & (d <= 5),
                    "medium (6-15 labs)": (d >= 6) & (d <= 15),
                    "high (16+ labs)": d >= 16}, predictions, targets)


def stratify_by_lab_frequency(predictions, targets, lab_indices, graph) -> Dict[str, Dict]:
    ei = graph["patient", "has_lab", "lab"].edge_index
    lab_counts = torch.bincount(ei[1], minlength=graph["lab"].num_nodes).cpu().numpy()
    f = lab_counts[lab_indices]
    q25 = np.percentile(lab_counts[lab_counts > 0], 25)
    q75 = np.percentile(lab_counts[lab_counts > 0], 75)
    return _groups({"rare (bottom 25%)": f < q25,
                    "common (middle 50%)": (f >= q25) & (f <= q75),
                    "very common (top 25%)": f > q75}, predictions, targets)


# ---------------------------------------------------------------------------------------------- device reducers
DEVICE_MAX_SEGMENTS = 2048       # EV_MAXSEG of csrc/evalred.hip (mmg_seg_reduce_ws_bytes returns 0 beyond it)


def metrics_from_sums(s) -> Dict[str, float]:
    """MAE / RMSE / R^2 / MAPE from one row (or the sum of rows) of ops.seg_sums:
    (n, sum|e|, sum e^2, sum t, sum t^2, sum|e/t| over t != 0, count(t != 0), clipped) -- evaluate.py:36-82."""
    n, s_abs, s_sq, s_t, s_t2, s_ape, n_nz = (float(s[i]) for i in range(7))
    if n <= 0:
        return {"mae": float("nan"), "rmse": float("nan"), "r2": float("nan"), "mape": float("nan")}
    ss_tot = s_t2 - s_t * s_t / n
    if n < 2:
        r2 = float("nan")                          # sklearn's r2_score: not defined for fewer than two samples
    elif s_sq == 0:
        r2 = 1.0
    elif ss_tot > 1e-12 * max(s_t2, 1e-300):
        r2 = 1.0 - s_sq / ss_tot
    else:
        r2 = 0.0                                   # constant target, non-zero residual (sklearn's force_finite)
    mape = s_ape / n_nz * 100.0 if n_nz > 0 else float("nan")
    return {"mae": s_abs / n, "rmse": float(np.sqrt(s_sq / n)), "r2": r2, "mape": mape}


def device_evaluate(predictions: torch.Tensor, targets: torch.Tensor, patient_indices: torch.Tensor,
                    lab_indices: torch.Tensor, graph, lab_names: Dict[int, str], stratify=(), n_sigma: float = 3.0):
    """Everything evaluate_model reports, from device-side segment sums (one D2H copy of a few [segments, 8] doubles):
    -> (overall metrics, number of clipped residuals, per-lab DataFrame, stratified results)."""
    from . import ops
    n_labs = int(graph["lab"].num_nodes)
    li64 = lab_indices.to(torch.int64).contiguous()
    # per-lab +-3 sigma winsorisation, then the per-lab sums of the ADJUSTED predictions; overall = their total
    sums, adj = ops.seg_sums(predictions.contiguous(), targets.contiguous(), li64, n_labs, n_sigma, want_adjusted=bool(stratify))
    per_lab = sums.cpu().numpy()
    overall = metrics_from_sums(per_lab.sum(0))
    num_capped = int(round(per_lab[:, 7].sum()))
    rows = []
    for j in range(n_labs):
        if per_lab[j, 0] < 2:
            continue
        m = metrics_from_sums(per_lab[j])
        m["lab_index"], m["lab_name"], m["num_samples"] = j, lab_names.get(j, f"Lab_{j}"), int(per_lab[j, 0])
        rows.append(m)
    per_lab_df = pd.DataFrame(rows)
    if len(per_lab_df):
        per_lab_df = per_lab_df.sort_values("mae")
    strat = {}
    ei = graph["patient", "has_lab", "lab"].edge_index
    if "num_labs" in stratify:                       # evaluate.py:237-287: 1-5 / 6-15 / 16+ observed labs
        deg = torch.bincount(ei[0], minlength=int(graph["patient"].num_nodes))[patient_indices]
        seg = torch.full_like(deg, -1)
        seg[(deg >= 1) & (deg <= 5)] = 0
        seg[(deg >= 6) & (deg <= 15)] = 1
        seg[deg >= 16] = 2
        g = ops.seg_sums(adj, targets.contiguous(), seg.contiguous(), 3)[0].cpu().numpy()
        strat["by_patient_degree"] = {nm: dict(metrics_from_sums(g[i]), num_samples=int(g[i, 0]))
                                      for i, nm in enumerate(("low (1-5 labs)", "medium (6-15 labs)", "high (16+ labs)"))
                                      if g[i, 0] > 0}
    if "lab_frequency" in stratify:                  # evaluate.py:290-342: quartiles of the non-zero lab counts
        counts = torch.bincount(ei[1], minlength=n_labs)
        cn = counts.cpu().numpy()
        q25, q75 = np.percentile(cn[cn > 0], 25), np.percentile(cn[cn > 0], 75)
        f = counts[li64].double()
        seg = torch.ones_like(li64)
        seg[f < q25] = 0
        seg[f > q75] = 2
        g = ops.seg_sums(adj, targets.contiguous(), seg.contiguous(), 3)[0].cpu().numpy()
        strat["by_lab_frequency"] = {nm: dict(metrics_from_sums(g[i]), num_samples=int(g[i, 0]))
                                     for i, nm in enumerate(("rare (bottom 25%)", "common (middle 50%)",
                                                             "very common (top 25%)")) if g[i, 0] > 0}
    return overall, num_capped, per_lab_df, strat


@torch.no_grad()
def evaluate_model(model, graph, test_edges, config: Dict, output_dir) -> Dict:
    """evaluate.py:349-570.  ``model`` is this package's HeteroRGCN (or anything with ``eval()``, ``parameters()`` and
    ``predict_lab_values``); writes ``per_lab_metrics.csv`` and ``evaluation_results.json`` into ``output_dir``."""
    output_dir = Path(output_dir)
    model.eval()
    device = next(model.parameters()).device
    graph = graph.to(device)
    edge_indices, edge_values = test_edges
    edge_indices = edge_indices.to(device)
    edge_values = edge_values.to(device)
    patient_indices, lab_indices = edge_indices[0], edge_indices[1]

    predictions = model.predict_lab_values(graph, patient_indices, lab_indices)

    ev = config["evaluation"]
    # the device reducers keep one [segments, 8] fp64 table in LDS: up to EV_MAXSEG = 2048 lab segments; a larger lab
    # vocabulary is reduced by the host reducers below (numpy, as the reference does it), not refused
    if predictions.is_cuda and int(graph["lab"].num_nodes) <= DEVICE_MAX_SEGMENTS:
        # HIP device: every reducer runs there (segment sums); the predictions never travel to the host
        lab_store = graph["lab"]
        meta = getattr(lab_store, "metadata", None) if "metadata" in lab_store else None
        lab_names = ({idx: m["label"] for idx, m in meta.items()} if meta
                     else {i: f"Lab_{i}" for i in range(graph["lab"].num_nodes)})
        overall_metrics, num_capped, per_lab_df, stratified_results = device_evaluate(
            predictions, edge_values.reshape(-1).float(), patient_indices, lab_indices, graph, lab_names,
            stratify=tuple(ev.get("stratify_by") or ()))
        logging.info(f"  Capped {num_capped}/{predictions.numel()} outlier residuals")
        logging.info(f"Overall: MAE {overall_metrics['mae']:.4f} RMSE {overall_metrics['rmse']:.4f} "
                     f"R2 {overall_metrics['r2']:.4f} MAPE {overall_metrics['mape']:.2f}%")
        if ev.get("per_lab_metrics", True):
            per_lab_df.to_csv(output_dir / "per_lab_metrics.csv", index=False)
        all_results = {"overall_metrics": overall_metrics, "num_test_samples": int(predictions.numel()),
                       "stratified_results": stratified_results}
        with open(output_dir / "evaluation_results.json", "w") as f:
            json.dump(all_results, f, indent=2)
        return all_results

    predictions_np = predictions.cpu().numpy()
    targets_np = edge_values.cpu().numpy()
    patient_indices_np = patient_indices.cpu().numpy()
    lab_indices_np = lab_indices.cpu().numpy()

    predictions_np, num_capped = winsorize_residuals(predictions_np, targets_np, lab_indices_np)
    logging.info(f"  Capped {num_capped}/{len(predictions_np)} outlier residuals")

    overall_metrics = compute_regression_metrics(predictions_np, targets_np)
    logging.info(f"Overall: MAE {overall_metrics['mae']:.4f} RMSE {overall_metrics['rmse']:.4f} "
                 f"R2 {overall_metrics['r2']:.4f} MAPE {overall_metrics['mape']:.2f}%")

    if ev.get("per_lab_metrics", True):
        lab_store = graph["lab"]
        meta = getattr(lab_store, "metadata", None) if "metadata" in lab_store else None
        if meta:
            lab_names = {idx: m["label"] for idx, m in meta.items()}
        else:
            lab_names = {i: f"Lab_{i}" for i in range(graph["lab"].num_nodes)}
        per_lab_df = compute_per_lab_metrics(predictions_np, targets_np, lab_indices_np, lab_names)
        per_lab_df.to_csv(output_dir / "per_lab_metrics.csv", index=False)

    stratified_results = {}
    if ev.get("stratify_by"):
        if "num_labs" in ev["stratify_by"]:
            stratified_results["by_patient_degree"] = stratify_by_patient_degree(
                predictions_np, targets_np, patient_indices_np, graph)
        if "lab_frequency" in ev["stratify_by"]:
            stratified_results["by_lab_frequency"] = stratify_by_lab_frequency(
                predictions_np, targets_np, lab_indices_np, graph)

    all_results = {"overall_metrics": overall_metrics, "num_test_samples": len(predictions_np),
                   "stratified_results": stratified_results}
    with open(output_dir / "evaluation_results.json", "w") as f:
        json.dump(all_results, f, indent=2)
    return all_results
