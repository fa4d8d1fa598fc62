#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own host code (build container only).

Runs only where /root/reference exists.  The reference's modules are imported from where
they lie (nothing is copied); torch_geometric -- a third-party dependency that is absent
from this image -- is provided by oracle/pyg_min.py (PyG's published SAGEConv/HeteroConv/
HeteroData semantics).  Outputs are DATA ONLY: inputs are the closed-form generators of
oracle/fixtures.py, so a fixture stores what the reference computed for them.

Pins:
  edges_*.npz      graph_build.build_heterogeneous_graph  -> edge_index x6, edge_attr, indexers
  splits.npz       train.EdgeMasker._create_splits         -> packed bool masks, E in {10, 61484}
  model_small.npz  model.HeteroRGCN on a 300-patient graph, D=64: encode_nodes / forward /
                   predict_lab_values in eval mode, train mode with dropout 0 (incl. the double
                   BN update of F7), Trainer._compute_lab_weights, Trainer.train_epoch loss +
                   every parameter / embedding gradient
  model_eicu.npz   the same at the eICU shape, D=128 (predictions for the 9,224 test pairs,
                   checksums of everything else)
  metrics.npz      evaluate.compute_regression_metrics + the winsorisation loop semantics
  eval_small.npz   evaluate.evaluate_model end to end on fixed predictions (results json, per-lab csv, baselines)

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np
import pandas as pd
import torch

from oracle import fixtures as fx
from oracle import pyg_min

pyg_min.install_as_torch_geometric()
sys.path.insert(0, os.path.join(REF, "src"))

import logging
logging.disable(logging.CRITICAL)

import graph_build as ref_gb      # noqa: E402
import model as ref_model         # noqa: E402
import train as ref_train         # noqa: E402
import evaluate as ref_eval       # noqa: E402
import utils as ref_utils         # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)  # sequential fp32 sums: reproducible fixtures


def save(name, tensors: dict, meta: dict = None):
    keys = list(tensors.keys())
    arrs = {f"t{i}": (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for i, v in enumerate(tensors.values())}
    np.savez_compressed(os.path.join(OUT, name), __keys__=np.array(json.dumps(keys)),
                        __meta__=np.array(json.dumps(meta or {})), **arrs)
    print(f"wrote {name}: {len(keys)} arrays, {os.path.getsize(os.path.join(OUT, name)) / 1024:.1f} KiB")


def ref_config(dropout=0.0, hidden=128):
    cfg = ref_utils.load_config(os.path.join(REF, "conf", "config.yaml"))
    cfg["model"]["dropout"] = dropout
    cfg["model"]["hidden_dim"] = hidden
    cfg["train"]["lr_scheduler"]["enabled"] = False   # verbose kwarg removed in torch>=2.7
    cfg["train"]["device"] = "cpu"
    cfg["logging"]["save_to_file"] = False
    return cfg


def frames_to_pandas(frames):
    pid, labs, dx, med = frames
    cohort = pd.DataFrame({"SUBJECT_ID": pid})
    labs_df = pd.DataFrame({"SUBJECT_ID": labs[0], "ITEMID": labs[1], "VALUE_NORMALIZED": labs[2]})
    dx_df = pd.DataFrame({"SUBJECT_ID": dx[0], "ICD3_CODE": dx[1]})
    med_df = pd.DataFrame({"SUBJECT_ID": med[0], "DRUG": med[1]})
    labitems = pd.DataFrame({"ITEMID": np.unique(labs[1]) if len(labs[1]) else np.array([], dtype=np.int64)})
    labitems["LABEL"] = [f"lab_{i}" for i in range(len(labitems))]
    return cohort, labs_df, dx_df, med_df, pd.DataFrame({"SUBJECT_ID": pid}), labitems


def ref_graph(frames, cfg):
    return ref_gb.build_heterogeneous_graph(*frames_to_pandas(frames), cfg)


def graph_tensors(g, prefix=""):
    out = {}
    for et in g.edge_types:
        out[prefix + "edge_index/" + "|".join(et)] = g[et].edge_index
        if "edge_attr" in g[et]:
            out[prefix + "edge_attr/" + "|".join(et)] = g[et].edge_attr
    out[prefix + "num_nodes"] = torch.tensor([g[t].num_nodes for t in g.node_types])
    return out


# ------------------------------------------------------------------------------------------
# 1. edge_index construction (graph_build.py)
# ------------------------------------------------------------------------------------------
def gen_edges():
    cfg = ref_config()
    # (a) closed-form frames, int ids
    g = ref_graph(fx.det_frames(60, 9, 11, 8), cfg)
    t = graph_tensors(g)
    meta = {"indexers": {k: v["id_to_index"] for k, v in g.indexers.items()}, "node_types": g.node_types,
            "edge_types": ["|".join(e) for e in g.edge_types]}
    save("edges_small.npz", t, meta)

    # (b) hand-made quirks: float ids (10006.0), string ids, ids unknown to the cohort, repeated rows,
    #     labs whose ITEMID never matches, and an EMPTY medication frame
    cohort = pd.DataFrame({"SUBJECT_ID": [10006.0, 10011.0, 10013.0, 10017.0, 10019.0]})
    labs = pd.DataFrame({"SUBJECT_ID": [10011, 10006, 99999, 10019, 10006, 10013, 10011],
                         "ITEMID": [50912, 50912, 50912, 50971, 50971, 50983, 50983],
                         "VALUE_NORMALIZED": [0.5, -1.25, 3.0, 0.0, 2.5, -0.75, 1.0]})
    dx = pd.DataFrame({"SUBJECT_ID": ["10006", "10017", "10017", "10020", "10011"],
                       "ICD3_CODE": ["428", "V45", "428", "250", 401]})
    med = pd.DataFrame({"SUBJECT_ID": pd.Series([], dtype=np.int64), "DRUG": pd.Series([], dtype=object)})
    labitems = pd.DataFrame({"ITEMID": [50912, 50971, 50983], "LABEL": ["a", "b", "c"]})
    g = ref_gb.build_heterogeneous_graph(cohort, labs, dx, med, cohort, labitems, cfg)
    t = graph_tensors(g)
    meta = {"indexers": {k: v["id_to_index"] for k, v in g.indexers.items()},
            "inputs": {"cohort": cohort["SUBJECT_ID"].tolist(),
                       "labs": [labs[c].tolist() for c in labs.columns],
                       "dx": [dx[c].tolist() for c in dx.columns]},
            "edge_types": ["|".join(e) for e in g.edge_types]}
    save("edges_quirks.npz", t, meta)


# ------------------------------------------------------------------------------------------
# 2. edge splits (train.py:98-129)
# ------------------------------------------------------------------------------------------
def gen_splits():
    out = {}
    for E in (10, 61484):
        g = pyg_min.HeteroData()
        g["patient"].num_nodes = 4
        g["lab"].num_nodes = 4
        g["patient", "has_lab", "lab"].edge_index = torch.zeros(2, E, dtype=torch.long)
        g["patient", "has_lab", "lab"].edge_attr = torch.zeros(E, 1)
        m = ref_train.EdgeMasker(g, 0.7, 0.15, 0.15, 0.2, seed=42)
        for nm, mask in (("train", m.train_mask), ("val", m.val_mask), ("test", m.test_mask)):
            out[f"E{E}/{nm}"] = np.packbits(mask.numpy())
            out[f"E{E}/{nm}_count"] = int(mask.sum())
    save("splits.npz", out)


# ------------------------------------------------------------------------------------------
# 3. model + trainer
# ------------------------------------------------------------------------------------------
def checksum(t):
    t = t.detach().double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()])


def run_model(tag, frames, hidden, full_tensors: bool):
    cfg = ref_config(0.0, hidden)
    g = ref_graph(frames, cfg)
    num_nodes = {t: g[t].num_nodes for t in g.node_types}
    sd = fx.det_state(num_nodes, hidden)

    model = ref_model.build_model(cfg, (g.node_types, g.edge_types), None)
    n_before = sum(p.numel() for p in model.parameters())
    model._init_embeddings(g)                                  # evaluate.py:629-630 order
    missing = model.load_state_dict(sd, strict=True)
    n_after = sum(p.numel() for p in model.parameters())
    out, meta = {}, {"params_before_embeddings": n_before, "params_after": n_after,
                     "state_keys": list(model.state_dict().keys()), "num_nodes": num_nodes,
                     "hidden": hidden}

    masker = ref_train.EdgeMasker(g, 0.7, 0.15, 0.15, 0.2, seed=42)
    ei_te, y_te, _, _ = masker.get_masked_data("test")
    meta["n_test"] = int(ei_te.shape[1])

    # ---- eval mode --------------------------------------------------------------------
    model.eval()
    with torch.no_grad():
        enc = model.encode_nodes(g)
        fwd = model(g)
        pred = model.predict_lab_values(g, ei_te[0], ei_te[1])
    out["eval/pred_test"] = pred
    for t in g.node_types:
        if full_tensors:
            out[f"eval/enc/{t}"] = enc[t]
            out[f"eval/fwd/{t}"] = fwd[t]
        else:
            out[f"eval/enc_rows/{t}"] = enc[t][:: max(1, enc[t].shape[0] // 16)][:16]
            out[f"eval/fwd_rows/{t}"] = fwd[t][:: max(1, fwd[t].shape[0] // 16)][:16]
        out[f"eval/enc_sum/{t}"] = checksum(enc[t])
        out[f"eval/fwd_sum/{t}"] = checksum(fwd[t])
    # evaluate.py metrics on this prediction (winsorised as evaluate_model does, :417-440)
    met = ref_eval.compute_regression_metrics(pred.numpy().copy(), y_te.numpy())
    meta["eval_metrics_raw"] = met

    # ---- Trainer: lab weights + one train epoch with a fixed wall clock (F8) -------------
    import time as _time
    real_time = _time.time
    _time.time = lambda: 1234.0                               # train.py:156 seeds from the clock
    try:
        model2 = ref_model.build_model(cfg, (g.node_types, g.edge_types), None)
        model2._init_embeddings(g)
        model2.load_state_dict(sd, strict=True)
        cfg["train"]["optimizer"]["lr"] = 1e-3
        trainer = ref_train.Trainer(model2, g, masker, cfg, torch.device("cpu"))
        out["lab_weights"] = trainer.lab_weights
        # Adam must see the embeddings for the post-step pin to cover them too?  No: keep the
        # reference behaviour (optimizer was built from model2.parameters() AFTER _init_embeddings
        # here, unlike train.py where it is built before -> F5).  Only loss/grads/buffers are pinned.
        ei_tr, y_tr, _, sup = masker.get_masked_data("train")
        out["train/sup_mask"] = np.packbits(sup.numpy())
        meta["n_train"] = int(ei_tr.shape[1]); meta["n_sup"] = int(sup.sum())
        loss = trainer.train_epoch()
        meta["train_loss"] = float(loss)
        with torch.no_grad():
            pass
        for k, p in model2.named_parameters():
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            if full_tensors or gr.numel() <= 4096:
                out[f"train/grad/{k}"] = gr
            else:
                out[f"train/grad_rows/{k}"] = gr.reshape(gr.shape[0], -1)[:: max(1, gr.shape[0] // 8)][:8]
            out[f"train/grad_sum/{k}"] = checksum(gr)
        for k, b in model2.named_buffers():
            out[f"train/buf/{k}"] = b
        # train-mode predictions (dropout 0) on the train pairs with the ORIGINAL weights
        model3 = ref_model.build_model(cfg, (g.node_types, g.edge_types), None)
        model3._init_embeddings(g); model3.load_state_dict(sd, strict=True); model3.train()
        with torch.no_grad():
            ptr = model3.predict_lab_values(g, ei_tr[0], ei_tr[1])
        out["train/pred_train" if full_tensors else "train/pred_train_head"] = ptr if full_tensors else ptr[:2048]
        out["train/pred_train_sum"] = checksum(ptr)
    finally:
        _time.time = real_time
    save(f"model_{tag}.npz", out, meta)


def gen_metrics():
    p = fx.det_uniform((500,), 5, -2, 2).numpy().astype(np.float64)
    t = fx.det_uniform((500,), 6, -2, 2).numpy().astype(np.float64)
    t[::50] = 0.0
    m = ref_eval.compute_regression_metrics(p, t)
    save("metrics.npz", {"pred": p, "target": t}, {"metrics": m})


class _FixedModel(torch.nn.Module):
    """Stands where a trained model would: evaluate_model only calls eval(), parameters() and
    predict_lab_values(), so fixed predictions pin everything AFTER the hot path."""

    def __init__(self, pred):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.pred = pred

    def predict_lab_values(self, graph, patient_indices, lab_indices):
        assert len(patient_indices) == len(self.pred)
        return self.pred.clone()


def gen_eval():
    """evaluate.evaluate_model end to end (winsorisation, overall, per-lab csv, both stratifications)."""
    import tempfile
    from pathlib import Path
    cfg = ref_config()
    g = ref_graph(fx.det_frames(300, 12, 15, 10), cfg)
    ei = g["patient", "has_lab", "lab"].edge_index
    ea = g["patient", "has_lab", "lab"].edge_attr.squeeze()
    E = ei.shape[1]
    sel = torch.nonzero(fx.det_uniform((E,), 21, 0, 1) < 0.3).squeeze(1)       # ~30 % "test" pairs
    tgt = ea[sel].clone()
    pred = tgt + fx.det_uniform((len(sel),), 22, -0.5, 0.5)
    pred[::37] += 6.0                                                           # outliers the 3-sigma guard caps
    pred[5::91] -= 4.0
    tgt[::29] = 0.0                                                             # exercises the MAPE mask
    with tempfile.TemporaryDirectory() as d:
        res = ref_eval.evaluate_model(_FixedModel(pred), g, (ei[:, sel], tgt), cfg, Path(d))
        per_lab = pd.read_csv(os.path.join(d, "per_lab_metrics.csv"))
    base = ref_eval.evaluate_baselines((ea.numpy().astype(np.float64), ei[1].numpy()),
                                       (tgt.numpy().astype(np.float64), ei[1, sel].numpy(), None))
    t = {"sel": sel, "pred": pred, "target": tgt}
    for c in ("mae", "rmse", "r2", "mape", "lab_index", "num_samples"):
        t["per_lab/" + c] = per_lab[c].to_numpy()
    save("eval_small.npz", t, {"results": res, "per_lab_names": per_lab["lab_name"].tolist(), "baselines": base,
                               "frames": [300, 12, 15, 10]})


if __name__ == "__main__":
    gen_edges()
    gen_splits()
    gen_metrics()
    gen_eval()
    run_model("small", fx.det_frames(300, 12, 15, 10), 64, full_tensors=True)
    run_model("eicu", fx.det_frames(1834, 50, 114, 100), 128, full_tensors=False)
    print("done")
