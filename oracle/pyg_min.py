"""Minimal CPU restatement of the three torch_geometric pieces the reference uses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch_geometric is a third-party
dependency of the reference (requirements.txt ``torch-geometric>=2.3.0``; no exact
pin; absent from /root/reference and from this image).  What is restated here is
PyG's *published* behaviour for exactly the surface the reference touches:

  * ``HeteroData``  -- graph_build.py:148-261, model.py:193-226,256,297, train.py:85-86,211
  * ``SAGEConv(in, out, aggr='mean')((x_src, x_dst), edge_index)`` -- model.py:125-129
        out_i = lin_l( mean_{e: dst(e)=i} x_src[src(e)] ) + lin_r( x_dst[i] )
        lin_l has a bias, lin_r has none; an empty neighbourhood aggregates to 0.
  * ``HeteroConv(convs, aggr='sum')(x_dict, edge_index_dict)`` -- model.py:131,256
        per destination type: sum over relations, relations visited in dict order.

Parity at this boundary is UNPINNED (nothing in the reference tests it).
"""
from __future__ import annotations

import copy
from typing import Dict, Tuple

import torch
import torch.nn as nn

Linear = nn.Linear  # torch_geometric.nn.Linear is only imported, never used, by model.py:23


# --------------------------------------------------------------------------------------
# HeteroData
# --------------------------------------------------------------------------------------
class _Store:
    """Attribute bag for one node type or one edge type."""

    def __init__(self):
        object.__setattr__(self, "_d", {})

    def __getattr__(self, k):
        d = object.__getattribute__(self, "_d")
        if k in d:
            return d[k]
        raise AttributeError(k)

    def __setattr__(self, k, v):
        self._d[k] = v

    def __contains__(self, k):
        return k in self._d

    def keys(self):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def _to(self, device):
        for k, v in list(self._d.items()):
            if torch.is_tensor(v):
                self._d[k] = v.to(device)


class HeteroData:
    """Dict-of-stores container: ``data['patient'].num_nodes``, ``data[s, r, d].edge_index``."""

    def __init__(self):
        object.__setattr__(self, "_nodes", {})
        object.__setattr__(self, "_edges", {})
        object.__setattr__(self, "_extra", {})

    # data['patient'] / data['patient','has_lab','lab'] / data[('patient','has_lab','lab')]
    def __getitem__(self, key):
        if isinstance(key, tuple):
            if len(key) != 3:
                raise KeyError(key)
            return self._edges.setdefault(tuple(key), _Store())
        return self._nodes.setdefault(key, _Store())

    def __getattr__(self, k):
        extra = object.__getattribute__(self, "_extra")
        if k in extra:
            return extra[k]
        raise AttributeError(k)

    def __setattr__(self, k, v):
        self._extra[k] = v

    @property
    def node_types(self):
        return list(self._nodes.keys())

    @property
    def edge_types(self):
        return list(self._edges.keys())

    def metadata(self):
        return self.node_types, self.edge_types

    @property
    def edge_index_dict(self) -> Dict[Tuple[str, str, str], torch.Tensor]:
        return {k: s.edge_index for k, s in self._edges.items() if "edge_index" in s}

    def to(self, device):
        for s in list(self._nodes.values()) + list(self._edges.values()):
            s._to(device)
        return self

    def clone(self):
        return copy.deepcopy(self)


# --------------------------------------------------------------------------------------
# Operators
# --------------------------------------------------------------------------------------
def scatter_mean(x_src: torch.Tensor, edge_index: torch.Tensor, num_dst: int) -> torch.Tensor:
    """mean_{e: dst(e)=i} x_src[src(e)]; rows without an incoming edge are 0.

    Sum in edge order via index_add_, divide by clamp(count, min=1).
    """
    src, dst = edge_index[0], edge_index[1]
    out = torch.zeros(num_dst, x_src.shape[1], dtype=x_src.dtype, device=x_src.device)
    out.index_add_(0, dst, x_src.index_select(0, src))
    cnt = torch.bincount(dst, minlength=num_dst).clamp_(min=1).to(x_src.dtype)
    return out / cnt.unsqueeze(1)


class SAGEConv(nn.Module):
    def __init__(self, in_channels, out_channels, aggr: str = "mean"):
        super().__init__()
        if aggr != "mean":
            raise NotImplementedError("only aggr='mean' is on the reference's path (model.py:128)")
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.lin_l = nn.Linear(in_channels[0], out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels[1], out_channels, bias=False)

    def forward(self, x, edge_index):
        if torch.is_tensor(x):
            x = (x, x)
        x_src, x_dst = x
        agg = scatter_mean(x_src, edge_index, x_dst.shape[0])
        return self.lin_l(agg) + self.lin_r(x_dst)


class HeteroConv(nn.Module):
    def __init__(self, convs: Dict[Tuple[str, str, str], nn.Module], aggr: str = "sum"):
        super().__init__()
        if aggr != "sum":
            raise NotImplementedError("only aggr='sum' is on the reference's path (model.py:131)")
        self.aggr = aggr
        self._keys = list(convs.keys())
        # PyG >= 2.4 mangles tuple keys as '<src___rel___dst>' 
        self.convs = nn.ModuleDict({self.mangle(k): m for k, m in convs.items()})

    @staticmethod
    def mangle(key):
        return "<" + "___".join(key) + ">"

    def forward(self, x_dict, edge_index_dict):
        outs: Dict[str, list] = {}
        for key in self._keys:
            src, _, dst = key
            if key not in edge_index_dict or src not in x_dict or dst not in x_dict:
                continue
            out = self.convs[self.mangle(key)]((x_dict[src], x_dict[dst]), edge_index_dict[key])
            outs.setdefault(dst, []).append(out)
        return {dst: torch.stack(v, dim=0).sum(dim=0) for dst, v in outs.items()}


def install_as_torch_geometric():
    """Register this module under the names the reference imports (gen_golden.py only)."""
    import sys
    import types

    tg = types.ModuleType("torch_geometric")
    tg_data = types.ModuleType("torch_geometric.data")
    tg_nn = types.ModuleType("torch_geometric.nn")
    tg_tr = types.ModuleType("torch_geometric.transforms")
    tg_data.HeteroData = HeteroData
    for name in ("HeteroConv", "SAGEConv", "Linear"):
        setattr(tg_nn, name, globals()[name])
    # imported by model.py:23-24 but unused on the RGCN path
    for name in ("GCNConv", "GATConv", "to_hetero", "HGTConv"):
        setattr(tg_nn, name, None)
    tg.data, tg.nn, tg.transforms = tg_data, tg_nn, tg_tr
    sys.modules.update({
        "torch_geometric": tg,
        "torch_geometric.data": tg_data,
        "torch_geometric.nn": tg_nn,
        "torch_geometric.transforms": tg_tr,
    })
