"""The vectorised edge_index builder (mmgnn.graph_build) against what the REFERENCE's iterrows builder produced
(tests/golden/edges_*.npz, oracle/gen_golden.py) -- bit-exact, same order -- and against the oracle's row-by-row
restatement on larger closed-form frames."""
import numpy as np
import pandas as pd
import pytest
import torch

import mmgnn  # noqa: F401
from mmgnn import graph_build as gb
from oracle import fixtures as fx
from golden_io import load, t

CFG = {"graph": {"edge_types": {k: {"enabled": True, "bidirectional": True}
                                for k in ("patient_lab", "patient_diagnosis", "patient_medication")}}}


def frames_to_pandas(frames):
    pid, labs, dx, med = frames
    cohort = pd.DataFrame({"SUBJECT_ID": pid})
    labs_df = pd.DataFrame({"SUBJECT_ID": labs[0], "ITEMID": labs[1], "VALUE_NORMALIZED": labs[2]})
    dx_df = pd.DataFrame({"SUBJECT_ID": dx[0], "ICD3_CODE": dx[1]})
    med_df = pd.DataFrame({"SUBJECT_ID": med[0], "DRUG": med[1]})
    labitems = pd.DataFrame({"ITEMID": np.unique(labs[1])})
    labitems["LABEL"] = [f"lab_{i}" for i in range(len(labitems))]
    return cohort, labs_df, dx_df, med_df, cohort.copy(), labitems


def check(g, gold, meta):
    assert ["|".join(e) for e in g.edge_types] == meta["edge_types"]
    for et in meta["edge_types"]:
        key = tuple(et.split("|"))
        ei = g[key].edge_index
        assert ei.dtype == torch.int64 and ei.is_contiguous()
        assert torch.equal(ei, t(gold["edge_index/" + et])), et
        if "edge_attr/" + et in gold:
            assert g[key].edge_attr.dtype == torch.float32
            assert torch.equal(g[key].edge_attr, t(gold["edge_attr/" + et])), et
    assert [g[n].num_nodes for n in g.node_types] == gold["num_nodes"].tolist()
    for nt, m in meta["indexers"].items():
        assert g.indexers[nt]["id_to_index"] == m
        assert list(g.indexers[nt]["id_to_index"]) == list(m)           # insertion order too


def test_golden_closed_form_frames():
    gold, meta = load("edges_small.npz")
    g = gb.build_heterogeneous_graph(*frames_to_pandas(fx.det_frames(60, 9, 11, 8)), CFG)
    check(g, gold, meta)
    assert g.node_types == meta["node_types"]
    assert g["lab"].metadata[0]["label"] == "lab_0" and g["lab"].metadata[0]["fluid"] == "Unknown"


def test_golden_quirks_float_string_unknown_repeated_empty():
    gold, meta = load("edges_quirks.npz")
    inp = meta["inputs"]
    cohort = pd.DataFrame({"SUBJECT_ID": inp["cohort"]})
    labs = pd.DataFrame(dict(zip(("SUBJECT_ID", "ITEMID", "VALUE_NORMALIZED"), inp["labs"])))
    dx = pd.DataFrame(dict(zip(("SUBJECT_ID", "ICD3_CODE"), inp["dx"])))
    med = pd.DataFrame({"SUBJECT_ID": pd.Series([], dtype=np.int64), "DRUG": pd.Series([], dtype=object)})
    labitems = pd.DataFrame({"ITEMID": [50912, 50971, 50983], "LABEL": ["a", "b", "c"]})
    g = gb.build_heterogeneous_graph(cohort, labs, dx, med, cohort, labitems, CFG)
    check(g, gold, meta)
    assert g["patient", "has_medication", "medication"].edge_index.shape == (2, 0)
    assert g["medication", "has_medication_rev", "patient"].edge_index.shape == (2, 0)


def test_matches_row_by_row_restatement_at_eicu_shape():
    frames = fx.det_frames(1834, 50, 114, 100)
    ref = fx.graph_from_frames(frames)
    g = gb.build_heterogeneous_graph(*frames_to_pandas(frames), CFG)
    assert g.edge_types == ref.edge_types
    for et in ref.edge_types:
        assert torch.equal(g[et].edge_index, ref[et].edge_index), et
    assert torch.equal(g["patient", "has_lab", "lab"].edge_attr, ref["patient", "has_lab", "lab"].edge_attr)
    for nt in ref.node_types:
        assert g.indexers[nt]["id_to_index"] == ref.indexers[nt]["id_to_index"]


def test_disabled_and_unidirectional_relations():
    cfg = {"graph": {"edge_types": {"patient_lab": {"enabled": True, "bidirectional": False},
                                    "patient_diagnosis": {"enabled": False, "bidirectional": True},
                                    "patient_medication": {"enabled": True, "bidirectional": True}}}}
    g = gb.build_heterogeneous_graph(*frames_to_pandas(fx.det_frames(40, 6, 5, 4)), cfg)
    assert g.edge_types == [("patient", "has_lab", "lab"), ("patient", "has_medication", "medication"),
                            ("medication", "has_medication_rev", "patient")]


def test_indexer_key_rule_and_nan():
    ix = gb.NodeIndexer()
    assert [ix.add(v) for v in (10006, 10006.0, "10006", np.int64(10006), np.float32(7.9), "7", "x")] == \
        [0, 0, 0, 0, 1, 1, 2]
    assert ix.get_index(7.2) == 1 and ix.get_index("nope") is None and ix.get_id(2) == "x" and len(ix) == 3
    assert ix.lookup(pd.Series([7.5, "x", 3, 10006])).tolist() == [1, 2, -1, 0]
    with pytest.raises(ValueError):            # int(nan), as in the reference
        ix.add(float("nan"))


def test_validate_graph_rejects_out_of_bounds():
    g = gb.build_heterogeneous_graph(*frames_to_pandas(fx.det_frames(40, 6, 5, 4)), CFG)
    g["patient"].num_nodes = 3
    with pytest.raises(ValueError, match="out-of-bounds source index"):
        gb.validate_graph(g)


def test_graph_file_round_trip_and_parquet_pipeline(tmp_path):
    frames = frames_to_pandas(fx.det_frames(60, 9, 11, 8))
    names = ("cohort", "labs_normalized", "diagnoses", "medications", "demographics", "labitems")
    for n, f in zip(names, frames):
        f.to_parquet(tmp_path / f"{n}.parquet")
    g = gb.build_graph_from_preprocessed(tmp_path, CFG, tmp_path / "out" / "graph.pt")
    gold, meta = load("edges_small.npz")
    check(g, gold, meta)
    h = gb.load_graph(tmp_path / "out" / "graph.pt")
    check(h, gold, meta)                                   # edge tensors, node counts, indexers incl. order
    assert h.node_types == g.node_types and h["lab"].metadata == g["lab"].metadata
    assert h.indexers["patient"]["index_to_id"][3] == g.indexers["patient"]["index_to_id"][3]
    st = gb.compute_graph_statistics(h)
    assert st["edge_counts"][("patient", "has_lab", "lab")] == g["patient", "has_lab", "lab"].edge_index.shape[1]
    assert 0 < st["density_patient_lab"] < 1 and st["patient_degree_has_lab"]["max"] <= 9
    torch.save({"x": 1}, tmp_path / "junk.pt")
    with pytest.raises(ValueError):
        gb.load_graph(tmp_path / "junk.pt")
