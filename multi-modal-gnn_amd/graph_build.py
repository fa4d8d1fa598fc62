"""Vectorised, bit-identical counterpart of the reference's ``src/graph_build.py`` edge_index construction
(SURVEY.md section 8 row a1 / "next" row f2).  Same names and semantics:

  * ``NodeIndexer``  (:34-97)   id -> contiguous index in first-seen order; numerics are keyed as ``str(int(id))``
                                 (so 10006.0 and 10006 collide), everything else as ``str(id)``;
  * ``create_patient_{lab,diagnosis,medication}_edges`` (:476-586)  one edge per frame row whose two ids are known,
    ROW ORDER PRESERVED, ``[2,E] int64`` contiguous (+ ``[E,1] float32`` values), empty -> ``[2,0]`` / ``[0,1]``;
  * ``build_heterogeneous_graph`` (:104-273) incl. the ``flip(0)`` reverse relations and ``validate_graph`` (:593-637).

The reference walks every row with ``DataFrame.iterrows`` (~35 us/row: an hour at the x1000 scale); here ids are
factorised once (``pd.factorize`` keeps first-seen order = ``Series.unique()`` order) and only the UNIQUE ids go
through the Python key rule.  Golden parity: tests/test_graph_build_cpu.py against tests/golden/edges_*.npz.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from .data import HeteroGraph


def _key(entity_id) -> str:
    # graph_build.py:66-70: numerics -> int first (a NaN raises ValueError there too), then str
    if isinstance(entity_id, (int, float, np.integer, np.floating)):
        entity_id = int(entity_id)
    return str(entity_id)


class NodeIndexer:
    """graph_build.py:34-97."""

    def __init__(self):
        self.id_to_index: Dict[str, int] = {}
        self.index_to_id: Dict[int, str] = {}
        self.next_index = 0

    def add(self, entity_id) -> int:
        k = _key(entity_id)
        idx = self.id_to_index.get(k)
        if idx is None:
            idx = self.next_index
            self.id_to_index[k] = idx
            self.index_to_id[idx] = k
            self.next_index += 1
        return idx

    def add_many(self, values) -> None:
        """Same as calling add() on every element in order (only first occurrences matter)."""
        for v in pd.unique(pd.Series(values, copy=False)):
            self.add(v)

    def get_index(self, entity_id) -> Optional[int]:
        return self.id_to_index.get(_key(entity_id))

    def get_id(self, index: int) -> Optional[str]:
        return self.index_to_id.get(index)

    def lookup(self, column: pd.Series) -> np.ndarray:
        """Vectorised get_index over a column: int64 array, -1 where the id is unknown."""
        codes, uniques = pd.factorize(column, use_na_sentinel=False)
        table = np.fromiter((self.id_to_index.get(_key(u), -1) for u in uniques), dtype=np.int64, count=len(uniques))
        return table[codes] if len(codes) else np.empty(0, dtype=np.int64)

    def __len__(self) -> int:
        return self.next_index

    def __repr__(self) -> str:
        return f"NodeIndexer(num_entities={len(self)})"


def _edges(frame: pd.DataFrame, src_col: str, dst_col: str, src_ix: NodeIndexer, dst_ix: NodeIndexer,
           value_col: Optional[str] = None):
    if len(frame) == 0:
        ei = torch.empty((2, 0), dtype=torch.long)
        return (ei, torch.empty((0, 1), dtype=torch.float32)) if value_col else ei
    # iterrows() upcasts an all-numeric row to float64 (the lab frame): the key rule makes that a no-op
    s = src_ix.lookup(frame[src_col])
    d = dst_ix.lookup(frame[dst_col])
    keep = (s >= 0) & (d >= 0)
    if not keep.any():
        ei = torch.empty((2, 0), dtype=torch.long)
        return (ei, torch.empty((0, 1), dtype=torch.float32)) if value_col else ei
    ei = torch.from_numpy(np.ascontiguousarray(np.stack([s[keep], d[keep]])))
    if value_col is None:
        return ei
    vals = frame[value_col].to_numpy(dtype=np.float64, copy=False)[keep]
    return ei, torch.from_numpy(vals.astype(np.float32)).unsqueeze(1)


def create_patient_lab_edges(labs, patient_indexer, lab_indexer) -> Tuple[torch.Tensor, torch.Tensor]:
    return _edges(labs, "SUBJECT_ID", "ITEMID", patient_indexer, lab_indexer, "VALUE_NORMALIZED")


def create_patient_diagnosis_edges(diagnoses, patient_indexer, diagnosis_indexer) -> torch.Tensor:
    return _edges(diagnoses, "SUBJECT_ID", "ICD3_CODE", patient_indexer, diagnosis_indexer)


def create_patient_medication_edges(medications, patient_indexer, medication_indexer) -> torch.Tensor:
    return _edges(medications, "SUBJECT_ID", "DRUG", patient_indexer, medication_indexer)


def create_lab_metadata(labitems: pd.DataFrame, indexer: NodeIndexer) -> Dict:
    meta = {}
    idx = indexer.lookup(labitems["ITEMID"]) if len(labitems) else np.empty(0, dtype=np.int64)
    cols = {c: (labitems[c].tolist() if c in labitems.columns else None) for c in ("LABEL", "FLUID", "CATEGORY")}
    items = labitems["ITEMID"].tolist() if len(labitems) else []
    for r, i in enumerate(idx.tolist()):
        if i >= 0:
            meta[i] = {"itemid": items[r],
                       "label": cols["LABEL"][r] if cols["LABEL"] is not None else "Unknown",
                       "fluid": cols["FLUID"][r] if cols["FLUID"] is not None else "Unknown",
                       "category": cols["CATEGORY"][r] if cols["CATEGORY"] is not None else "Unknown"}
    return meta


def validate_graph(data) -> None:
    """graph_build.py:593-637 (same ValueError texts)."""
    for node_type in data.node_types:
        if data[node_type].num_nodes == 0:
            logging.warning(f"Node type '{node_type}' has 0 nodes!")
    for edge_type in data.edge_types:
        edge_index = data[edge_type].edge_index
        if edge_index.shape[0] != 2:
            raise ValueError(f"Edge type {edge_type} has invalid shape: {edge_index.shape}")
        src_type, _, dst_type = edge_type
        if edge_index.shape[1] > 0:
            if edge_index[0].max() >= data[src_type].num_nodes:
                raise ValueError(f"Edge type {edge_type} has out-of-bounds source index: {edge_index[0].max()} >= "
                                 f"{data[src_type].num_nodes}")
            if edge_index[1].max() >= data[dst_type].num_nodes:
                raise ValueError(f"Edge type {edge_type} has out-of-bounds destination index: {edge_index[1].max()} >= "
                                 f"{data[dst_type].num_nodes}")


def build_heterogeneous_graph(cohort, labs, diagnoses, medications, demographics, labitems, config: Dict) -> HeteroGraph:
    """graph_build.py:104-273: same node order, edge order, reverse relations and attached metadata."""
    data = HeteroGraph()
    indexers = {t: NodeIndexer() for t in ("patient", "lab", "diagnosis", "medication")}
    indexers["patient"].add_many(cohort["SUBJECT_ID"])                       # :163-164 cohort order
    indexers["lab"].add_many(labs["ITEMID"])                                 # :166-167 .unique() order
    indexers["diagnosis"].add_many(diagnoses["ICD3_CODE"])
    indexers["medication"].add_many(medications["DRUG"])
    data["patient"].num_nodes = len(indexers["patient"])
    data["lab"].num_nodes = len(indexers["lab"])
    data["lab"].metadata = create_lab_metadata(labitems, indexers["lab"])
    data["diagnosis"].num_nodes = len(indexers["diagnosis"])
    data["medication"].num_nodes = len(indexers["medication"])

    ec = config["graph"]["edge_types"]
    if ec["patient_lab"]["enabled"]:
        ei, ea = create_patient_lab_edges(labs, indexers["patient"], indexers["lab"])
        data["patient", "has_lab", "lab"].edge_index = ei
        data["patient", "has_lab", "lab"].edge_attr = ea
        if ec["patient_lab"]["bidirectional"]:
            data["lab", "has_lab_rev", "patient"].edge_index = ei.flip(0)
            data["lab", "has_lab_rev", "patient"].edge_attr = ea
    if ec["patient_diagnosis"]["enabled"]:
        ei = create_patient_diagnosis_edges(diagnoses, indexers["patient"], indexers["diagnosis"])
        data["patient", "has_diagnosis", "diagnosis"].edge_index = ei
        if ec["patient_diagnosis"]["bidirectional"]:
            data["diagnosis", "has_diagnosis_rev", "patient"].edge_index = ei.flip(0)
    if ec["patient_medication"]["enabled"]:
        ei = create_patient_medication_edges(medications, indexers["patient"], indexers["medication"])
        data["patient", "has_medication", "medication"].edge_index = ei
        if ec["patient_medication"]["bidirectional"]:
            data["medication", "has_medication_rev", "patient"].edge_index = ei.flip(0)
    data.indexers = {t: {"id_to_index": ix.id_to_index, "index_to_id": ix.index_to_id} for t, ix in indexers.items()}
    data.config = config
    validate_graph(data)
    return data


def compute_graph_statistics(data) -> Dict:
    """graph_build.py:644-720: node / edge counts, patient out-degree moments per relation, has_lab density."""
    stats = {"node_counts": {t: data[t].num_nodes for t in data.node_types},
             "edge_counts": {et: int(data[et].edge_index.shape[1]) for et in data.edge_types}}
    for et in data.edge_types:
        s, rel, _ = et
        if s == "patient":
            deg = torch.bincount(data[et].edge_index[0], minlength=data["patient"].num_nodes)
            stats[f"patient_degree_{rel}"] = {"mean": deg.float().mean().item(), "std": deg.float().std().item(),
                                              "min": deg.min().item(), "max": deg.max().item(),
                                              "median": deg.median().item()}
    if ("patient", "has_lab", "lab") in data.edge_types:
        n = data["patient", "has_lab", "lab"].edge_index.shape[1]
        stats["density_patient_lab"] = n / (data["patient"].num_nodes * data["lab"].num_nodes)
    return stats


# ------------------------------------------------------------------------------------------------------------------
# graph file: the reference pickles the PyG object (torch.save(graph), graph_build.py:769; torch.load, train.py:601),
# which needs torch_geometric to read back.  Here a graph file is a plain dict of tensors and builtins, readable with
# torch.load(weights_only=True) and nothing else; load_graph() also accepts a pickled HeteroData when PyG is installed.
# ------------------------------------------------------------------------------------------------------------------
GRAPH_FORMAT = "mmgnn.hetero_graph.v1"


def save_graph(data, path) -> None:
    nodes = {}
    for t in data.node_types:
        nodes[t] = {"num_nodes": int(data[t].num_nodes)}
        md = getattr(data[t], "metadata", None) if hasattr(data[t], "metadata") else None
        if isinstance(md, dict):
            nodes[t]["metadata"] = {int(k): dict(v) for k, v in md.items()}
    edges = []
    for et in data.edge_types:
        rec = {"type": tuple(et), "edge_index": data[et].edge_index.cpu().contiguous()}
        if hasattr(data[et], "edge_attr") and getattr(data[et], "edge_attr", None) is not None:
            rec["edge_attr"] = data[et].edge_attr.cpu().contiguous()
        edges.append(rec)
    blob = {"format": GRAPH_FORMAT, "nodes": nodes, "edges": edges}
    ix = getattr(data, "indexers", None) if hasattr(data, "indexers") else None
    if ix is not None:
        blob["indexers"] = {t: {"ids": list(m["id_to_index"].keys())} for t, m in ix.items()}   # index = position
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(blob, path)


def load_graph(path, map_location=None):
    try:
        blob = torch.load(path, map_location=map_location, weights_only=True)
    except Exception:
        blob = torch.load(path, map_location=map_location, weights_only=False)      # a pickled HeteroData (needs PyG)
    if not (isinstance(blob, dict) and blob.get("format") == GRAPH_FORMAT):
        if hasattr(blob, "edge_types") and hasattr(blob, "node_types"):
            return blob
        raise ValueError(f"{path}: not a {GRAPH_FORMAT} file nor a HeteroData pickle")
    g = HeteroGraph()
    for t, rec in blob["nodes"].items():
        g[t].num_nodes = rec["num_nodes"]
        if "metadata" in rec:
            g[t].metadata = rec["metadata"]
    for rec in blob["edges"]:
        et = tuple(rec["type"])
        g[et].edge_index = rec["edge_index"]
        if "edge_attr" in rec:
            g[et].edge_attr = rec["edge_attr"]
    if "indexers" in blob:
        g.indexers = {t: {"id_to_index": {k: i for i, k in enumerate(m["ids"])},
                          "index_to_id": dict(enumerate(m["ids"]))} for t, m in blob["indexers"].items()}
    return g


def build_graph_from_preprocessed(interim_dir, config: Dict, output_path=None) -> HeteroGraph:
    """graph_build.py:727-772: the six parquet files of preprocess.py -> graph (-> graph file)."""
    interim_dir = Path(interim_dir)
    frames = [pd.read_parquet(interim_dir / f"{n}.parquet")
              for n in ("cohort", "labs_normalized", "diagnoses", "medications", "demographics", "labitems")]
    graph = build_heterogeneous_graph(*frames, config)
    compute_graph_statistics(graph)
    if output_path:
        save_graph(graph, output_path)
    return graph
