"""The CPU oracle must reproduce what the REFERENCE's own host code produced
(tests/golden/*.npz, written by oracle/gen_golden.py from /root/reference).

Tolerances: integer/byte/index work bit-exact; fp32 <= 1e-5 relative (oracle vs reference
host code, both fp32 CPU).
"""
import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import graph as og
from oracle import model as om
from oracle import train as ot
from golden_io import checksum, load, rel_err, t, unpack_mask

FP_TOL = 1e-5


def _check_graph(g, gold, meta):
    for et in meta["edge_types"]:
        key = tuple(et.split("|"))
        assert torch.equal(g[key].edge_index, t(gold["edge_index/" + et])), et
        assert g[key].edge_index.dtype == torch.int64 and g[key].edge_index.is_contiguous()
        if "edge_attr/" + et in gold:
            assert torch.equal(g[key].edge_attr, t(gold["edge_attr/" + et])), et
    assert [g[n].num_nodes for n in g.node_types] == gold["num_nodes"].tolist()
    for nt, m in meta["indexers"].items():
        assert g.indexers[nt]["id_to_index"] == m


def test_edge_index_closed_form_frames():
    gold, meta = load("edges_small.npz")
    g = fx.graph_from_frames(fx.det_frames(60, 9, 11, 8))
    _check_graph(g, gold, meta)


def test_edge_index_quirks_float_string_unknown_empty():
    gold, meta = load("edges_quirks.npz")
    inp = meta["inputs"]
    g = og.build_graph(inp["cohort"], tuple(inp["labs"]), tuple(inp["dx"]), ([], []))
    _check_graph(g, gold, meta)
    # empty relation shape contract (graph_build.py:580-582)
    assert g["patient", "has_medication", "medication"].edge_index.shape == (2, 0)


@pytest.mark.parametrize("E", [10, 61484])
def test_edge_splits(E):
    gold, _ = load("splits.npz")
    masks = ot.edge_splits(E, 0.7, 0.15, 0.15, seed=42)
    for nm, m in zip(("train", "val", "test"), masks):
        assert torch.equal(m, unpack_mask(gold[f"E{E}/{nm}"], E))
        assert int(m.sum()) == int(gold[f"E{E}/{nm}_count"])
    if E == 61484:  # published sizes (outputs/evaluation_results.json:8)
        assert [int(m.sum()) for m in masks] == [43038, 9222, 9224]


def test_regression_metrics():
    gold, meta = load("metrics.npz")
    m = ot.regression_metrics(gold["pred"], gold["target"])
    for k, v in meta["metrics"].items():
        assert abs(m[k] - v) <= 1e-9 * max(1.0, abs(v)), k


def test_param_count_pins():
    # README.md:197 -- 483,970 trainable parameters before the lazy embeddings; 752,514 after
    nn_ = {"patient": 1834, "lab": 50, "diagnosis": 114, "medication": 100}
    sd = fx.det_state(nn_, 128)
    fl = {k: v for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    assert sum(v.numel() for k, v in fl.items() if not k.startswith("embeddings")) == 483970
    assert sum(v.numel() for v in fl.values()) == 752514


def _setup(tag, n, hidden):
    gold, meta = load(f"model_{tag}.npz")
    g = fx.graph_from_frames(fx.det_frames(*n))
    gv = om.GraphView(g)
    assert gv.num_nodes == meta["num_nodes"]
    sd = fx.det_state(gv.num_nodes, hidden)
    assert sorted(sd.keys()) == sorted(meta["state_keys"])
    ei = g["patient", "has_lab", "lab"].edge_index
    ea = g["patient", "has_lab", "lab"].edge_attr
    masks = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    return gold, meta, g, gv, sd, ei, ea, masks


@pytest.mark.parametrize("tag,n,hidden", [("small", (300, 12, 15, 10), 64), ("eicu", (1834, 50, 114, 100), 128)])
def test_model_eval_matches_reference(tag, n, hidden):
    gold, meta, g, gv, sd, ei, ea, (tr, va, te) = _setup(tag, n, hidden)
    with torch.no_grad():
        enc = om.encode_nodes(sd, gv)
        fwd = om.forward(sd, gv)
        pred, _ = om.predict_lab_values(sd, gv, ei[0][te], ei[1][te])
    assert pred.shape[0] == meta["n_test"]
    assert rel_err(pred, t(gold["eval/pred_test"])) <= FP_TOL
    for nt in gv.node_types:
        if f"eval/enc/{nt}" in gold:
            assert rel_err(enc[nt], t(gold[f"eval/enc/{nt}"])) <= FP_TOL
            assert rel_err(fwd[nt], t(gold[f"eval/fwd/{nt}"])) <= FP_TOL
        else:
            st = max(1, enc[nt].shape[0] // 16)
            assert rel_err(enc[nt][::st][:16], t(gold[f"eval/enc_rows/{nt}"])) <= FP_TOL
            assert rel_err(fwd[nt][::st][:16], t(gold[f"eval/fwd_rows/{nt}"])) <= FP_TOL
        assert rel_err(checksum(fwd[nt]), t(gold[f"eval/fwd_sum/{nt}"])) <= 1e-4
    # the gate must see both branches (model.py:312-315)
    deg = torch.bincount(ei[0], minlength=gv.num_nodes["patient"])
    low = deg[ei[0][te]] < 6
    assert low.any() and (~low).any()


@pytest.mark.parametrize("tag,n,hidden", [("small", (300, 12, 15, 10), 64), ("eicu", (1834, 50, 114, 100), 128)])
def test_train_step_matches_reference(tag, n, hidden):
    gold, meta, g, gv, sd, ei, ea, (tr, va, te) = _setup(tag, n, hidden)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    assert rel_err(w, t(gold["lab_weights"])) <= FP_TOL
    # train.py:156 with the wall clock pinned to 1234 by gen_golden.py
    torch.manual_seed(1234)
    sup = ot.supervision_mask(int(tr.sum()), 0.2)
    assert torch.equal(sup, unpack_mask(gold["train/sup_mask"], int(tr.sum())))
    assert int(sup.sum()) == meta["n_sup"]
    loss, pred, grads, bufs = ot.train_step_grads(sd, gv, pi, li, y, w, sup, p=0.0)
    assert abs(float(loss) - meta["train_loss"]) <= FP_TOL * abs(meta["train_loss"])
    assert rel_err(checksum(pred), t(gold["train/pred_train_sum"])) <= 1e-4
    for k, gr in grads.items():
        if f"train/grad/{k}" in gold:
            ref = t(gold[f"train/grad/{k}"])
            # biases in front of a BatchNorm have an analytically ZERO gradient (pure rounding
            # noise ~1e-8): absolute floor 1e-7 next to the relative bound
            assert float((gr - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-7, k
        cs, cr = checksum(gr), t(gold[f"train/grad_sum/{k}"])
        assert abs(float(cs[1] - cr[1])) <= 2e-4 * float(cr[1]) + 1e-7 * gr.numel(), k
    # BN running buffers after ONE step: patient_transform BNs were updated TWICE (F7)
    for k, b in bufs.items():
        ref = t(gold[f"train/buf/{k}"])
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(ref), k
        else:
            assert rel_err(b, ref) <= FP_TOL, k
    assert int(bufs["patient_transform.1.num_batches_tracked"]) == 3 + 2
    assert int(bufs["batch_norms.0.patient.num_batches_tracked"]) == 3 + 1


def test_oracle_fp64_agrees_with_fp32():
    g = fx.graph_from_frames(fx.det_frames(300, 12, 15, 10))
    gv = om.GraphView(g)
    sd = fx.det_state(gv.num_nodes, 64)
    ei = g["patient", "has_lab", "lab"].edge_index
    with torch.no_grad():
        p32, _ = om.predict_lab_values(sd, gv, ei[0], ei[1])
        p64, _ = om.predict_lab_values(om.cast_state(sd, torch.float64), gv, ei[0], ei[1])
    assert rel_err(p32, p64) <= 1e-5


def test_injected_dropout_masks_are_used():
    g = fx.graph_from_frames(fx.det_frames(120, 8, 9, 7))
    gv = om.GraphView(g)
    sd = fx.det_state(gv.num_nodes, 64)
    ei = g["patient", "has_lab", "lab"].edge_index
    n, P = ei.shape[1], gv.num_nodes["patient"]
    shapes = {"enc0.drop0": (P, 64), "enc0.drop1": (P, 64), "enc1.drop0": (P, 64), "enc1.drop1": (P, 64),
              "head.drop0": (n, 64), "head.drop1": (n, 32)}
    shapes.update({f"conv0.{nt}": (gv.num_nodes[nt], 64) for nt in gv.node_types})
    masks = fx.det_masks(shapes, 0.2)
    with torch.no_grad():
        a, _ = om.predict_lab_values(sd, gv, ei[0], ei[1], training=True, p=0.2, masks=masks)
        b, _ = om.predict_lab_values(sd, gv, ei[0], ei[1], training=True, p=0.2, masks=masks)
        c, _ = om.predict_lab_values(sd, gv, ei[0], ei[1], training=True, p=0.0)
    assert torch.equal(a, b) and not torch.allclose(a, c)


def test_csr_reference_properties():
    g = fx.graph_from_frames(fx.det_frames(60, 9, 11, 8))
    ei = g["patient", "has_lab", "lab"].edge_index
    rowptr, col, perm = og.csr_reference(ei, 60, 0)
    assert int(rowptr[-1]) == ei.shape[1]
    src = ei[0][perm.long()]
    assert torch.all(src[1:] >= src[:-1])
    assert torch.equal(col.long(), ei[1][perm.long()])
    # stability: inside a row the original edge ids increase
    for r in range(60):
        seg = perm[rowptr[r]:rowptr[r + 1]]
        assert torch.all(seg[1:] > seg[:-1])
