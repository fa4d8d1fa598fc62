"""Graph container + static graph plan for the hot path.

``HeteroGraph`` speaks the subset of PyG's ``HeteroData`` protocol that the reference touches
(src/graph_build.py:148-261, src/model.py:193-226,256,297, src/train.py:85-86,211 of the reference), so
the model accepts either a real PyG ``HeteroData`` or this container.

``GraphPlan`` is what the kernels consume: one CSR-by-patient per relation (built once by the HIP
radix sort ``mmg_csr_build``; the graph is static across epochs), the mean-aggregation
reciprocals, and the has_lab degree that gates the two heads (src/model.py:297-298).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops

EdgeType = Tuple[str, str, str]
ROW_TYPE = "patient"          # the big, sharded axis; every relation has exactly one patient endpoint
LAB_EDGE = ("patient", "has_lab", "lab")


class _Store:
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

    def _to(self, device):
        for k, v in list(self._d.items()):
            if torch.is_tensor(v):
                self._d[k] = v.to(device)


class HeteroGraph:
    """``g['patient'].num_nodes = n``; ``g['patient','has_lab','lab'].edge_index = [2,E] int64``."""

    def __init__(self):
        object.__setattr__(self, "_nodes", {})
        object.__setattr__(self, "_edges", {})
        object.__setattr__(self, "_extra", {})

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
    def node_types(self) -> List[str]:
        return list(self._nodes.keys())

    @property
    def edge_types(self) -> List[EdgeType]:
        return list(self._edges.keys())

    def metadata(self):
        return self.node_types, self.edge_types

    @property
    def edge_index_dict(self) -> Dict[EdgeType, torch.Tensor]:
        return {k: s.edge_index for k, s in self._edges.items() if "edge_index" in s}

    def to(self, device):
        for s in list(self._nodes.values()) + list(self._edges.values()):
            s._to(device)
        return self

    @classmethod
    def from_edges(cls, num_nodes: Dict[str, int], edges: Dict[EdgeType, torch.Tensor],
                   edge_attr: Optional[Dict[EdgeType, torch.Tensor]] = None) -> "HeteroGraph":
        g = cls()
        for t, n in num_nodes.items():
            g[t].num_nodes = int(n)
        for et, ei in edges.items():
            g[et].edge_index = ei
            if edge_attr and et in edge_attr:
                g[et].edge_attr = edge_attr[et]
        return g


# --------------------------------------------------------------------------------------------
@dataclass
class RelCSR:
    """One relation, stored by patient row."""
    edge_type: EdgeType
    other: str                      # the non-patient node type
    patient_is_dst: bool            # True: (other -> patient); False: (patient -> other)
    n_cols: int
    rowptr: torch.Tensor            # int32 [P+1]
    col: torch.Tensor               # int32 [E]
    perm: torch.Tensor              # int32 [E] original edge ids
    inv_row: torch.Tensor           # f32 [P]    1/max(row degree,1)   (mean into patients)
    inv_col: torch.Tensor           # f32 [n_cols] 1/max(col in-degree,1) (mean into vocab nodes)
    col_cnt: torch.Tensor           # int32 [n_cols] LOCAL in-degree (summed across shards for inv_col)
    n_edges: int = 0
    simple: bool = False            # no (patient, item) pair occurs twice (true for the reference's frames)
    mask_t: Optional[torch.Tensor] = None    # bit planes [ceil(P/64)][pad32(n_cols)][2] of a simple relation
    mask_r: Optional[torch.Tensor] = None    # the same row-major: uint16 fields [P][2][pad32(n_cols)/16]


@dataclass
class GraphPlan:
    node_types: List[str]
    edge_types: List[EdgeType]
    num_nodes: Dict[str, int]
    device: torch.device
    rels: Dict[EdgeType, RelCSR] = field(default_factory=dict)
    lab_deg: Optional[torch.Tensor] = None     # int32 [P] has_lab out-degree (full graph)
    row_offset: int = 0                        # global id of local patient 0 (sharding)
    n_rows_global: int = 0
    key: tuple = ()

    @property
    def n_rows(self) -> int:
        return self.num_nodes[ROW_TYPE]

    def rels_into_patient(self) -> List[RelCSR]:
        return [self.rels[e] for e in self.edge_types if self.rels[e].patient_is_dst]

    def rels_from_patient(self) -> List[RelCSR]:
        return [self.rels[e] for e in self.edge_types if not self.rels[e].patient_is_dst]


def _plan_key(data) -> tuple:
    k = []
    for et in data.edge_types:
        ei = data[et].edge_index
        k.append((tuple(et), ei.data_ptr(), ei._version, tuple(ei.shape), str(ei.device)))
    k.append(tuple((t, int(data[t].num_nodes)) for t in data.node_types))
    return tuple(k)


# key -> (plan, the edge_index tensors the key was derived from).  The entry HOLDS those tensors: a key is built from
# their data_ptr(), and the caching allocator hands a freed address to the next graph of the same size -- without the
# reference a stale plan (CSR, masks, degrees of the old graph) would be returned for it.
_PLAN_CACHE: Dict[tuple, tuple] = {}
_PLAN_CACHE_MAX_BYTES = 4 << 30       # pinned edge_index tensors + plans kept alive by the cache (an x1000 graph is ~2.7 GB)


def _plan_bytes(plan: "GraphPlan", eis) -> int:
    n = sum(int(t.numel()) * t.element_size() for t in eis)
    seen = set()
    for r in plan.rels.values():
        for t in (r.rowptr, r.col, r.perm, r.inv_row, r.inv_col, r.col_cnt, r.mask_t, r.mask_r):
            if t is not None and id(t) not in seen:
                seen.add(id(t))
                n += int(t.numel()) * t.element_size()
    return n


def drop_cached_plan(plan: "GraphPlan"):
    """Remove `plan` from the cache (dist.shard_plan rewrites it in place: a later unsharded use of the same graph must
    not get the sharded plan)."""
    ent = _PLAN_CACHE.get(plan.key)
    if ent is not None and ent[0] is plan:
        del _PLAN_CACHE[plan.key]


def build_plan(data, device=None, validate: bool = True, use_cache: bool = True) -> GraphPlan:
    """CSR-by-patient for every relation of ``data`` (a PyG HeteroData or a HeteroGraph)."""
    key = _plan_key(data)
    if use_cache and key in _PLAN_CACHE:
        return _PLAN_CACHE[key][0]
    node_types = list(data.node_types)
    edge_types = [tuple(e) for e in data.edge_types]
    num_nodes = {t: int(data[t].num_nodes) for t in node_types}
    if ROW_TYPE not in num_nodes:
        raise ValueError(f"graph has no '{ROW_TYPE}' node type")
    if device is None:
        device = data[edge_types[0]].edge_index.device if edge_types else torch.device("cuda")
    device = torch.device(device)
    P = num_nodes[ROW_TYPE]
    plan = GraphPlan(node_types, edge_types, num_nodes, device, n_rows_global=P, key=key)
    shared: Dict[Tuple[str, str], Tuple[torch.Tensor, RelCSR]] = {}
    # adjacency bit planes are only built where a kernel can take them: the matrix-core aggregates hold <= 768 padded
    # items per launch (24 tiles of 32; every patient relation of one direction is fused into one launch), so a larger
    # vocabulary -- e.g. 3,000 diagnosis codes -- keeps the CSR kernels and allocates no masks (they would be
    # P * items / 4 bytes each)
    others = {(e[0] if e[2] == ROW_TYPE else e[2]) for e in edge_types
              if (e[0] == ROW_TYPE) != (e[2] == ROW_TYPE)}
    masks_fit = sum((num_nodes[o] + 31) // 32 * 32 for o in others) <= 768
    for et in edge_types:
        s, _, d = et
        if (s == ROW_TYPE) == (d == ROW_TYPE):
            raise NotImplementedError(f"relation {et}: exactly one endpoint must be '{ROW_TYPE}' "
                                      "(the reference's schema, graph_build.py:128-141)")
        ei = data[et].edge_index
        if ei.dim() != 2 or ei.shape[0] != 2:
            raise ValueError(f"Edge type {et} has invalid shape: {tuple(ei.shape)}")     # graph_build.py:615-616
        if ei.dtype != torch.int64:
            raise TypeError(f"Edge type {et}: edge_index must be int64")
        ei = ei.to(device).contiguous()
        patient_is_dst = d == ROW_TYPE
        other = s if patient_is_dst else d
        sort_row = 1 if patient_is_dst else 0
        E = int(ei.shape[1])
        if validate and E > 0:                                                           # graph_build.py:618-633
            mx = ei.max(dim=1).values.tolist()
            mn = int(ei.min())
            lim = (num_nodes[s], num_nodes[d])
            if mn < 0 or mx[0] >= lim[0] or mx[1] >= lim[1]:
                raise ValueError(f"Edge type {et} has out-of-bounds index (max {mx}, sizes {lim})")
        # the reference builds reverse relations as edge_index.flip(0): share the CSR when it is one
        twin = shared.get((other, "dst" if not patient_is_dst else "src"))
        rel = None
        if twin is not None:
            t_ei, t_rel = twin
            if t_ei.shape == ei.shape and torch.equal(t_ei.flip(0), ei):
                rel = RelCSR(et, other, patient_is_dst, t_rel.n_cols, t_rel.rowptr, t_rel.col, t_rel.perm,
                             t_rel.inv_row, t_rel.inv_col, t_rel.col_cnt, E, t_rel.simple, t_rel.mask_t, t_rel.mask_r)
        if rel is None:
            rowptr, col, perm = ops.csr_build(ei, P, sort_row)
            _, inv_row = ops.row_degree(rowptr)
            cnt, inv_col = ops.col_degree(col, num_nodes[other])
            prow = ei[1] if patient_is_dst else ei[0]
            ocol = ei[0] if patient_is_dst else ei[1]
            simple = masks_fit and (E == 0 or int(torch.unique(prow * num_nodes[other] + ocol).numel()) == E)  # one-off
            mask_t, mask_r = ops.rel_mask_build(rowptr, col, num_nodes[other]) if simple and P > 0 else (None, None)
            rel = RelCSR(et, other, patient_is_dst, num_nodes[other], rowptr, col, perm, inv_row, inv_col, cnt, E,
                         simple, mask_t, mask_r)
            shared[(other, "src" if not patient_is_dst else "dst")] = (ei, rel)
        plan.rels[et] = rel
    if LAB_EDGE in plan.rels:
        r = plan.rels[LAB_EDGE]
        plan.lab_deg, _ = ops.row_degree(r.rowptr)
    if use_cache:
        eis = [data[et].edge_index for et in data.edge_types]
        nbytes = _plan_bytes(plan, eis)
        # bounded by entries AND by bytes: oldest entries go first; a plan larger than the budget is not cached at all
        while _PLAN_CACHE and (len(_PLAN_CACHE) >= 8 or
                               sum(e[2] for e in _PLAN_CACHE.values()) + nbytes > _PLAN_CACHE_MAX_BYTES):
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
        if nbytes <= _PLAN_CACHE_MAX_BYTES:
            _PLAN_CACHE[key] = (plan, eis, nbytes)
    return plan
