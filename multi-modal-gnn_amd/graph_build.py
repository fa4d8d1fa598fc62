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
