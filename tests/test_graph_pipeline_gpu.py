"""The path's input side at BASELINE scale: ehr-like frames of the eICU shape x100 (6.1 M lab rows, 0.54 M diagnosis
rows, 1.6 M medication rows) -> mmgnn.graph_build.build_heterogeneous_graph (the vectorised counterpart of the
reference's iterrows builder, graph_build.py:476-586: ~35 us per row there, i.e. ~5 minutes for these frames) -> the
device CSR (mmg_csr_build) -> one forward of the model.  The builder's output, not synth.make_graph's, feeds the kernels
here; edge tensors are checked bit for bit against the frames through the indexers."""
import time

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG_G = {"graph": {"edge_types": {k: {"enabled": True, "bidirectional": True}
                                  for k in ("patient_lab", "patient_diagnosis", "patient_medication")}}}
LAB = ("patient", "has_lab", "lab")


def test_vectorised_builder_at_x100_feeds_the_device_csr():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import mmgnn  # noqa: F401
    from mmgnn import graph_build as gb, ops
    from mmgnn.data import build_plan
    from mmgnn.model import build_model
    from mmgnn.synth import make_graph
    dev = torch.device("cuda:0")
    src = make_graph(100, seed=3, device=dev, with_reverse=False)
    P = int(src["patient"].num_nodes)
    pid = 10000 + 3 * np.arange(P, dtype=np.int64)                       # non-contiguous SUBJECT_IDs
    lab_e = src[LAB].edge_index.cpu().numpy()
    dx_e = src["patient", "has_diagnosis", "diagnosis"].edge_index.cpu().numpy()
    med_e = src["patient", "has_medication", "medication"].edge_index.cpu().numpy()
    vals = src[LAB].edge_attr.squeeze(-1).cpu().numpy().astype(np.float64)
    cohort = pd.DataFrame({"SUBJECT_ID": pid})
    labs = pd.DataFrame({"SUBJECT_ID": pid[lab_e[0]], "ITEMID": 50800 + 7 * lab_e[1], "VALUE_NORMALIZED": vals})
    dx_codes = np.array([f"{300 + 3 * j}" if j % 4 else f"V{10 + j}" for j in range(int(src["diagnosis"].num_nodes))], dtype=object)
    med_names = np.array([f"drug_{j:03d}" for j in range(int(src["medication"].num_nodes))], dtype=object)
    dx = pd.DataFrame({"SUBJECT_ID": pid[dx_e[0]], "ICD3_CODE": dx_codes[dx_e[1]]})
    med = pd.DataFrame({"SUBJECT_ID": pid[med_e[0]], "DRUG": med_names[med_e[1]]})
    labitems = pd.DataFrame({"ITEMID": 50800 + 7 * np.arange(int(src["lab"].num_nodes))})
    labitems["LABEL"] = [f"lab_{i}" for i in range(len(labitems))]

    t0 = time.perf_counter()
    g = gb.build_heterogeneous_graph(cohort, labs, dx, med, cohort.copy(), labitems, CFG_G)
    dt = time.perf_counter() - t0
    n_rows = len(labs) + len(dx) + len(med)
    print(f"build_heterogeneous_graph: {n_rows} frame rows in {dt:.2f} s ({n_rows / dt / 1e6:.2f} M rows/s; "
          f"the reference's iterrows loop: ~35 us/row = {35e-6 * n_rows:.0f} s)")
    assert dt < 60.0
    # ---- bit-exact against the frames, through the indexers (first-seen order of every id column)
    assert [int(g[t].num_nodes) for t in g.node_types] == [P, 50, 114, 100]
    for et, e_src, ids, nt in ((LAB, lab_e, (50800 + 7 * np.arange(50)).astype(str), "lab"),
                               (("patient", "has_diagnosis", "diagnosis"), dx_e, dx_codes, "diagnosis"),
                               (("patient", "has_medication", "medication"), med_e, med_names, "medication")):
        ei = g[et].edge_index
        assert ei.dtype == torch.int64 and ei.is_contiguous() and ei.shape == (2, e_src.shape[1])
        assert np.array_equal(ei[0].numpy(), e_src[0])                                  # cohort order = patient index
        to_ix = np.array([g.indexers[nt]["id_to_index"][str(i)] for i in ids])
        assert np.array_equal(ei[1].numpy(), to_ix[e_src[1]])
        assert sorted(to_ix.tolist()) == list(range(len(ids)))
        rev = (et[2], et[1] + "_rev", et[0])
        assert torch.equal(g[rev].edge_index, ei.flip(0))
    assert torch.equal(g[LAB].edge_attr.squeeze(-1), torch.from_numpy(vals.astype(np.float32)))
    # ---- the builder's tensors through the device CSR
    gd = g.to(dev)
    plan = build_plan(gd, dev, use_cache=False)
    for et in (LAB, ("diagnosis", "has_diagnosis_rev", "patient")):
        ei = gd[et].edge_index
        rel = plan.rels[et]
        prow = ei[1] if rel.patient_is_dst else ei[0]
        ocol = ei[0] if rel.patient_is_dst else ei[1]
        order = torch.sort(prow, stable=True).indices
        assert torch.equal(rel.perm.long(), order) and torch.equal(rel.col.long(), ocol[order])
        assert torch.equal(rel.rowptr.long(), torch.cat([torch.zeros(1, dtype=torch.long, device=dev),
                                                         torch.bincount(prow, minlength=P).cumsum(0)]))
        assert rel.simple and rel.mask_t is not None
    # ---- and through one forward of the model
    cfg = {"model": {"architecture": "RGCN", "hidden_dim": 128, "num_layers": 2, "dropout": 0.0,
                     "use_batch_norm": True, "activation": "relu"}}
    torch.manual_seed(1)
    model = build_model(cfg, (gd.node_types, gd.edge_types), None).to(dev)
    model.eval()
    with torch.no_grad():
        out = model(gd)
        pred = model.predict_lab_values(gd, gd[LAB].edge_index[0][:100000].contiguous(), gd[LAB].edge_index[1][:100000].contiguous())
    assert all(torch.isfinite(v).all() for v in out.values()) and torch.isfinite(pred).all()
    assert out["patient"].shape == (P, 128)
