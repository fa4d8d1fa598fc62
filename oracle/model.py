"""Functional CPU restatement of the reference's HeteroRGCN hot path (TEST INFRASTRUCTURE).

Works on a plain ``state_dict`` in the reference's key layout so that
the same weights can be pushed through (a) the reference's own ``model.py`` (golden
fixtures, oracle/gen_golden.py), (b) this restatement and (c) the HIP product.

Every function cites the reference lines it follows (paths relative to /root/reference).
Pure PyTorch CPU ops; dtype follows the state (fp32, or fp64 via ``cast_state``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .pyg_min import scatter_mean

EdgeType = Tuple[str, str, str]
BN_MOMENTUM = 0.1   # nn.BatchNorm1d default (model.py:95,99,136)
BN_EPS = 1e-5
L2_EPS = 1e-12      # F.normalize default (model.py:232)


def mangle(edge_type: EdgeType) -> str:
    """PyG >= 2.4 ModuleDict key for a relation."""
    return "<" + "___".join(edge_type) + ">"


class GraphView:
    """The attributes of HeteroData the hot path reads (model.py:193-226,256,297-298)."""

    def __init__(self, data):
        self.node_types: List[str] = list(data.node_types)
        self.edge_types: List[EdgeType] = [tuple(e) for e in data.edge_types]
        self.num_nodes: Dict[str, int] = {t: int(data[t].num_nodes) for t in self.node_types}
        self.edge_index: Dict[EdgeType, torch.Tensor] = {
            e: data[e].edge_index for e in self.edge_types}


def cast_state(sd: Dict[str, torch.Tensor], dtype) -> Dict[str, torch.Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def _activation(name: str):
    # model.py:145-152
    if name == "relu":
        return F.relu
    if name == "elu":
        return F.elu
    if name == "leaky_relu":
        return F.leaky_relu
    raise ValueError(f"Unknown activation: {name}")


def _batch_norm(x, sd, prefix, training, bufs):
    """nn.BatchNorm1d: train -> batch mean / biased var, running stats with momentum 0.1
    and UNBIASED var; eval -> running stats.  ``bufs`` carries the running buffers so two
    calls in one step (F7: encode_nodes runs twice, model.py:294,301->251) chain."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm_k, rv_k, nb_k = prefix + ".running_mean", prefix + ".running_var", prefix + ".num_batches_tracked"
    if training:
        n = x.shape[0]
        if n <= 1:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {list(x.shape)}")
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        with torch.no_grad():
            bufs[rm_k] = (1 - BN_MOMENTUM) * bufs[rm_k] + BN_MOMENTUM * mean.detach()
            bufs[rv_k] = (1 - BN_MOMENTUM) * bufs[rv_k] + BN_MOMENTUM * var.detach() * (n / (n - 1))
            bufs[nb_k] = bufs[nb_k] + 1
    else:
        mean, var = bufs[rm_k].to(x.dtype), bufs[rv_k].to(x.dtype)
    return (x - mean) / torch.sqrt(var + BN_EPS) * w + b


def _dropout(x, p, training, masks, name, rows=None):
    """F.dropout; with ``masks`` given, use the injected keep-mask (1 = keep) instead of
    the torch RNG so a device implementation with another RNG can be compared exactly."""
    if not training or p == 0.0:
        return x
    if masks is None:
        return F.dropout(x, p=p, training=True)
    m = masks[name]
    if rows is not None:
        m = m[rows]
    return x * m.to(x.dtype) / (1.0 - p)


def _linear(x, sd, prefix, bias=True):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"] if bias else None)


def _bufs_of(sd):
    return {k: v.detach().clone() for k, v in sd.items()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


# --------------------------------------------------------------------------------------
# model.py:206-234  encode_nodes
# --------------------------------------------------------------------------------------
def encode_nodes(sd, g: GraphView, *, training=False, p=0.0, masks=None, bufs=None, call=0):
    bufs = _bufs_of(sd) if bufs is None else bufs
    x = {}
    for t in g.node_types:
        # Embedding(arange(N)) == the whole table (model.py:222-226)
        x[t] = sd[f"embeddings.{t}.weight"][: g.num_nodes[t]]
    if "patient" in x:
        h = x["patient"]
        # patient_transform = Linear,BN,ReLU,Drop,Linear,BN,ReLU,Drop,Linear (model.py:93-103)
        h = _linear(h, sd, "patient_transform.0")
        h = _batch_norm(h, sd, "patient_transform.1", training, bufs)
        h = _dropout(F.relu(h), p, training, masks, f"enc{call}.drop0")
        h = _linear(h, sd, "patient_transform.4")
        h = _batch_norm(h, sd, "patient_transform.5", training, bufs)
        h = _dropout(F.relu(h), p, training, masks, f"enc{call}.drop1")
        h = _linear(h, sd, "patient_transform.8")
        x["patient"] = F.normalize(h, p=2, dim=1, eps=L2_EPS)   # model.py:232
    return x


# --------------------------------------------------------------------------------------
# PyG SAGEConv / HeteroConv (call sites model.py:125-131,256)
# --------------------------------------------------------------------------------------
def hetero_sage_layer(sd, g: GraphView, x: Dict[str, torch.Tensor], layer: int):
    outs: Dict[str, list] = {}
    for et in g.edge_types:
        s, _, d = et
        pre = f"convs.{layer}.convs.{mangle(et)}"
        agg = scatter_mean(x[s], g.edge_index[et], x[d].shape[0])
        out = F.linear(agg, sd[pre + ".lin_l.weight"], sd[pre + ".lin_l.bias"]) \
            + F.linear(x[d], sd[pre + ".lin_r.weight"])
        outs.setdefault(d, []).append(out)
    return {d: torch.stack(v, 0).sum(0) for d, v in outs.items()}


# --------------------------------------------------------------------------------------
# model.py:236-271  forward
# --------------------------------------------------------------------------------------
def forward(sd, g: GraphView, *, num_layers=2, training=False, p=0.0, masks=None, bufs=None,
            use_batch_norm=True, activation="relu", call=0):
    bufs = _bufs_of(sd) if bufs is None else bufs
    act = _activation(activation)
    x = encode_nodes(sd, g, training=training, p=p, masks=masks, bufs=bufs, call=call)
    for l in range(num_layers):
        x = hetero_sage_layer(sd, g, x, l)                                 # model.py:256
        if use_batch_norm:                                                 # model.py:259-261
            x = {t: _batch_norm(v, sd, f"batch_norms.{l}.{t}", training, bufs) for t, v in x.items()}
        x = {t: act(v) for t, v in x.items()}                              # model.py:264
        if l < num_layers - 1:                                             # model.py:267-269
            x = {t: _dropout(v, p, training, masks, f"conv{l}.{t}") for t, v in x.items()}
    return x


# --------------------------------------------------------------------------------------
# model.py:342-396  EdgeRegressionHead
# --------------------------------------------------------------------------------------
def edge_head(sd, prefix, z, *, training, p, masks, rows):
    h = F.relu(_linear(z, sd, f"{prefix}.mlp.0"))
    h = _dropout(h, p, training, masks, "head.drop0", rows)
    h = F.relu(_linear(h, sd, f"{prefix}.mlp.3"))
    h = _dropout(h, p, training, masks, "head.drop1", rows)
    return _linear(h, sd, f"{prefix}.mlp.6")


# --------------------------------------------------------------------------------------
# model.py:273-335  predict_lab_values
# --------------------------------------------------------------------------------------
def predict_lab_values(sd, g: GraphView, patient_indices, lab_indices, *, num_layers=2,
                       training=False, p=0.0, masks=None, bufs=None, degree_threshold=6,
                       use_batch_norm=True, activation="relu"):
    """Returns (predictions [n], bufs) -- ``bufs`` holds the post-call BN running buffers."""
    bufs = _bufs_of(sd) if bufs is None else bufs
    init = encode_nodes(sd, g, training=training, p=p, masks=masks, bufs=bufs, call=0)   # :294
    ei = g.edge_index[("patient", "has_lab", "lab")]
    deg = torch.bincount(ei[0], minlength=g.num_nodes["patient"])                        # :297-298
    fin = forward(sd, g, num_layers=num_layers, training=training, p=p, masks=masks,     # :301
                  bufs=bufs, use_batch_norm=use_batch_norm, activation=activation, call=1)
    low = deg[patient_indices] < degree_threshold                                        # :312-315
    pred = torch.zeros(len(patient_indices), dtype=init["patient"].dtype)
    if low.any():                                                                        # :319-325
        z = torch.cat([init["patient"][patient_indices][low], init["lab"][lab_indices][low]], 1)
        pred[low] = edge_head(sd, "tabular_mlp", z, training=training, p=p, masks=masks, rows=low).squeeze(-1)
    if (~low).any():                                                                     # :327-333
        z = torch.cat([fin["patient"][patient_indices][~low], fin["lab"][lab_indices][~low]], 1)
        pred[~low] = edge_head(sd, "edge_predictor", z, training=training, p=p, masks=masks, rows=~low).squeeze(-1)
    return pred, bufs


# --------------------------------------------------------------------------------------
# model.py:579-612  compute_regression_loss
# --------------------------------------------------------------------------------------
def compute_regression_loss(predictions, targets, loss_type="mae"):
    if loss_type == "mae":
        return (predictions - targets).abs().mean()
    if loss_type == "mse":
        return ((predictions - targets) ** 2).mean()
    if loss_type == "huber":
        return F.huber_loss(predictions, targets)
    raise ValueError(f"Unknown loss type: {loss_type}")


# --------------------------------------------------------------------------------------
# state construction with the reference's initialisers (for tests / bench; not a product API)
# --------------------------------------------------------------------------------------
def init_state(num_nodes: Dict[str, int], edge_types: Sequence[EdgeType], hidden_dim=128,
               num_layers=2, seed=42, generator: Optional[torch.Generator] = None):
    """Fresh state_dict in the reference layout: nn.Linear default init (kaiming-uniform a=sqrt(5)),
    BN weight 1 / bias 0 / running (0,1), embeddings Xavier-uniform (model.py:198-199)."""
    import math
    gen = generator or torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(prefix, fan_out, fan_in, bias=True):
        bound = 1.0 / math.sqrt(fan_in)
        sd[prefix + ".weight"] = (torch.rand(fan_out, fan_in, generator=gen) * 2 - 1) * bound
        if bias:
            sd[prefix + ".bias"] = (torch.rand(fan_out, generator=gen) * 2 - 1) * bound

    def bn(prefix, d):
        sd[prefix + ".weight"] = torch.ones(d)
        sd[prefix + ".bias"] = torch.zeros(d)
        sd[prefix + ".running_mean"] = torch.zeros(d)
        sd[prefix + ".running_var"] = torch.ones(d)
        sd[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    D = hidden_dim
    lin("patient_transform.0", D, D); bn("patient_transform.1", D)
    lin("patient_transform.4", D, D); bn("patient_transform.5", D)
    lin("patient_transform.8", D, D)
    for l in range(num_layers):
        for et in edge_types:
            pre = f"convs.{l}.convs.{mangle(tuple(et))}"
            lin(pre + ".lin_l", D, D, bias=True)
            lin(pre + ".lin_r", D, D, bias=False)
    for l in range(num_layers):
        for t in num_nodes:
            bn(f"batch_norms.{l}.{t}", D)
    for head in ("edge_predictor", "tabular_mlp"):
        lin(f"{head}.mlp.0", 64, 2 * D)
        lin(f"{head}.mlp.3", 32, 64)
        lin(f"{head}.mlp.6", 1, 32)
    for t, n in num_nodes.items():
        bound = math.sqrt(6.0 / (n + D))
        sd[f"embeddings.{t}.weight"] = (torch.rand(n, D, generator=gen) * 2 - 1) * bound
    return sd
