"""CPU restatement of the reference's edge_index construction + the CSR definition
(TEST INFRASTRUCTURE; loops are fine here -- small cases only).

Follows /root/reference/src/graph_build.py:
  * NodeIndexer.add / get_index           :34-97   (key = str(int(id)) for numerics, first-seen order)
  * indexing order                        :162-173 (patients in cohort order; others in .unique() order)
  * create_patient_{lab,diagnosis,medication}_edges :476-586 (row order kept, unknown ids dropped,
    empty -> [2,0] int64 / [0,1] f32, else tensor(list).t().contiguous())
  * reverse relations = edge_index.flip(0), has_lab_rev shares edge_attr :216-248
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from .pyg_min import HeteroData


def _key(entity_id) -> str:
    # graph_build.py:66-70 / :84-86
    if isinstance(entity_id, (int, float, np.integer, np.floating)):
        entity_id = int(entity_id)
    return str(entity_id)


class Indexer:
    def __init__(self):
        self.id_to_index: Dict[str, int] = {}
        self.index_to_id: Dict[int, str] = {}

    def add(self, entity_id) -> int:
        k = _key(entity_id)
        if k not in self.id_to_index:
            i = len(self.id_to_index)
            self.id_to_index[k] = i
            self.index_to_id[i] = k
        return self.id_to_index[k]

    def get(self, entity_id) -> Optional[int]:
        return self.id_to_index.get(_key(entity_id))

    def __len__(self):
        return len(self.id_to_index)


def _unique_first_seen(values: Iterable) -> List:
    seen, out = set(), []
    for v in values:
        # pandas .unique() keeps first-seen order and collapses NaNs to one entry
        k = ("nan",) if (isinstance(v, float) and v != v) else v
        if k not in seen:
            seen.add(k)
            out.append(v)
    return out


def build_edges(src_ids, dst_ids, src_ix: Indexer, dst_ix: Indexer, values=None):
    pairs, attrs = [], []
    for r, (a, b) in enumerate(zip(src_ids, dst_ids)):
        i, j = src_ix.get(a), dst_ix.get(b)
        if i is not None and j is not None:
            pairs.append((i, j))
            if values is not None:
                attrs.append(values[r])
    if not pairs:
        ei = torch.empty((2, 0), dtype=torch.long)
        ea = torch.empty((0, 1), dtype=torch.float32)
    else:
        ei = torch.tensor(pairs, dtype=torch.long).t().contiguous()
        ea = torch.tensor(attrs, dtype=torch.float32).unsqueeze(1) if values is not None else None
    return (ei, ea) if values is not None else ei


def build_graph(cohort_ids, labs, diagnoses, medications, bidirectional=True) -> HeteroData:
    """labs = (SUBJECT_ID[], ITEMID[], VALUE_NORMALIZED[]); diagnoses = (SUBJECT_ID[], ICD3_CODE[]);
    medications = (SUBJECT_ID[], DRUG[]).  Mirrors build_heterogeneous_graph :155-248."""
    ix = {t: Indexer() for t in ("patient", "lab", "diagnosis", "medication")}
    for s in cohort_ids:
        ix["patient"].add(s)
    for v in _unique_first_seen(labs[1]):
        ix["lab"].add(v)
    for v in _unique_first_seen(diagnoses[1]):
        ix["diagnosis"].add(v)
    for v in _unique_first_seen(medications[1]):
        ix["medication"].add(v)
    data = HeteroData()
    for t in ix:
        data[t].num_nodes = len(ix[t])
    ei, ea = build_edges(labs[0], labs[1], ix["patient"], ix["lab"], labs[2])
    data["patient", "has_lab", "lab"].edge_index = ei
    data["patient", "has_lab", "lab"].edge_attr = ea
    if bidirectional:
        data["lab", "has_lab_rev", "patient"].edge_index = ei.flip(0)
        data["lab", "has_lab_rev", "patient"].edge_attr = ea
    ei = build_edges(diagnoses[0], diagnoses[1], ix["patient"], ix["diagnosis"])
    data["patient", "has_diagnosis", "diagnosis"].edge_index = ei
    if bidirectional:
        data["diagnosis", "has_diagnosis_rev", "patient"].edge_index = ei.flip(0)
    ei = build_edges(medications[0], medications[1], ix["patient"], ix["medication"])
    data["patient", "has_medication", "medication"].edge_index = ei
    if bidirectional:
        data["medication", "has_medication_rev", "patient"].edge_index = ei.flip(0)
    data.indexers = {t: {"id_to_index": ix[t].id_to_index, "index_to_id": ix[t].index_to_id} for t in ix}
    return data


# --------------------------------------------------------------------------------------
# CSR: the definition the HIP csr_build must match bit for bit
# --------------------------------------------------------------------------------------
def csr_reference(edge_index: torch.Tensor, num_rows: int, sort_row: int = 0):
    """Stable sort of the edge ids by ``edge_index[sort_row]``.

    Returns (rowptr int32 [num_rows+1], col int32 [E], perm int32 [E]) where
    perm[k] is the ORIGINAL edge id at CSR slot k (ties keep original order) and
    col[k] = edge_index[1 - sort_row][perm[k]].
    """
    key = edge_index[sort_row]
    other = edge_index[1 - sort_row]
    perm = torch.sort(key, stable=True).indices
    counts = torch.bincount(key, minlength=num_rows)
    rowptr = torch.zeros(num_rows + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), other[perm].to(torch.int32), perm.to(torch.int32)
