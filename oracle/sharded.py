"""Patient-sharded CPU restatement (TEST INFRASTRUCTURE): proves that the decomposition the product uses on
N GPUs -- contiguous patient ranges, SUM all-reduce of the patient->vocab partial sums, Sync-BatchNorm
statistics over the patient axis, replicated vocab tables and weights -- reproduces the single-process
oracle (oracle/model.py) exactly up to fp32 re-association.  Runs under torch.distributed (gloo) with
autograd-aware collectives, so gradients are checked too.

Each rank holds: embeddings.patient rows [lo, hi), the edges of those patients (local patient ids), its
supervision pairs.  Everything else is replicated.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.distributed as dist
import torch.distributed.nn.functional as dfn
import torch.nn.functional as F

from . import model as om


def _allreduce(t):
    return dfn.all_reduce(t, op=dist.ReduceOp.SUM)


def _sync_bn(x, sd, prefix, training, bufs, n_global, sharded):
    """BatchNorm1d whose batch is the GLOBAL patient axis (train: stats all-reduced)."""
    if not sharded:
        return om._batch_norm(x, sd, prefix, training, bufs)
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm_k, rv_k, nb_k = prefix + ".running_mean", prefix + ".running_var", prefix + ".num_batches_tracked"
    if training:
        s = _allreduce(torch.stack([x.sum(0), (x * x).sum(0)]))
        mean = s[0] / n_global
        var = (s[1] / n_global - mean * mean).clamp_min(0)
        with torch.no_grad():
            bufs[rm_k] = (1 - om.BN_MOMENTUM) * bufs[rm_k] + om.BN_MOMENTUM * mean.detach()
            bufs[rv_k] = (1 - om.BN_MOMENTUM) * bufs[rv_k] + om.BN_MOMENTUM * var.detach() * (n_global / (n_global - 1))
            bufs[nb_k] = bufs[nb_k] + 1
    else:
        mean, var = bufs[rm_k], bufs[rv_k]
    return (x - mean) / torch.sqrt(var + om.BN_EPS) * w + b


def _encode(sd, g, n_global, training, bufs):
    x = {t: sd[f"embeddings.{t}.weight"] for t in g.node_types}
    h = x["patient"]
    h = om._linear(h, sd, "patient_transform.0")
    h = F.relu(_sync_bn(h, sd, "patient_transform.1", training, bufs, n_global, True))
    h = om._linear(h, sd, "patient_transform.4")
    h = F.relu(_sync_bn(h, sd, "patient_transform.5", training, bufs, n_global, True))
    h = om._linear(h, sd, "patient_transform.8")
    x["patient"] = F.normalize(h, p=2, dim=1, eps=om.L2_EPS)
    return x


def _layer(sd, g, x, layer, global_cnt):
    outs: Dict[str, list] = {}
    for et in g.edge_types:
        s, _, d = et
        pre = f"convs.{layer}.convs.{om.mangle(et)}"
        ei = g.edge_index[et]
        if d == "patient":        # vocab -> patient: every edge of a patient is local
            agg = om.scatter_mean(x[s], ei, x[d].shape[0])
        else:                     # patient -> vocab: local partial SUM, all-reduce, divide by the GLOBAL count
            part = torch.zeros(x[d].shape[0], x[s].shape[1], dtype=x[s].dtype).index_add_(0, ei[1], x[s][ei[0]])
            agg = _allreduce(part) / global_cnt[et].clamp(min=1).to(part.dtype).unsqueeze(1)
        out = F.linear(agg, sd[pre + ".lin_l.weight"], sd[pre + ".lin_l.bias"]) + F.linear(x[d], sd[pre + ".lin_r.weight"])
        outs.setdefault(d, []).append(out)
    return {d: torch.stack(v, 0).sum(0) for d, v in outs.items()}


def predict_sharded(sd, g: om.GraphView, pi, li, n_global: int, *, training=True, num_layers=2, bufs=None,
                    degree_threshold=6):
    """predict_lab_values on this rank's shard (dropout 0).  Returns (pred_local, bufs)."""
    bufs = om._bufs_of(sd) if bufs is None else bufs
    global_cnt = {}
    for et in g.edge_types:
        if et[2] != "patient":
            c = torch.bincount(g.edge_index[et][1], minlength=g.num_nodes[et[2]])
            dist.all_reduce(c)
            global_cnt[et] = c
    init = _encode(sd, g, n_global, training, bufs)
    deg = torch.bincount(g.edge_index[("patient", "has_lab", "lab")][0], minlength=g.num_nodes["patient"])
    x = _encode(sd, g, n_global, training, bufs)
    for l in range(num_layers):
        x = _layer(sd, g, x, l, global_cnt)
        x = {t: F.relu(_sync_bn(v, sd, f"batch_norms.{l}.{t}", training, bufs, n_global, t == "patient"))
             for t, v in x.items()}
    low = deg[pi] < degree_threshold
    pred = torch.zeros(len(pi), dtype=init["patient"].dtype)
    if low.any():
        z = torch.cat([init["patient"][pi][low], init["lab"][li][low]], 1)
        pred[low] = om.edge_head(sd, "tabular_mlp", z, training=training, p=0.0, masks=None, rows=low).squeeze(-1)
    if (~low).any():
        z = torch.cat([x["patient"][pi][~low], x["lab"][li][~low]], 1)
        pred[~low] = om.edge_head(sd, "edge_predictor", z, training=training, p=0.0, masks=None, rows=~low).squeeze(-1)
    return pred, bufs
