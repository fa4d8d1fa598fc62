"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/mmgnn.h
declares; the host mirror refuses to run without the HIP path; container + state_dict plumbing."""
import ctypes
import os
import re

import pytest
import torch

from oracle import fixtures as fx

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"model": {"architecture": "RGCN", "hidden_dim": 64, "num_layers": 2, "dropout": 0.0,
                 "use_batch_norm": True, "activation": "relu"}}


def header_symbols():
    txt = open(os.path.join(REPO, "include", "mmgnn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mmg_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    import mmgnn  # noqa: F401
    from mmgnn import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/mmgnn.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes prototype in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.mmg_version() >= 100


def test_argument_errors_are_reported_without_a_gpu():
    import mmgnn  # noqa: F401
    from mmgnn import _lib
    lib = _lib.load()
    # bad D is rejected on the host side before any launch
    rc = lib.mmg_l2norm_fwd(None, None, None, 4, 100, 1e-12, None)
    assert rc == -1 and b"unsupported" in lib.mmg_last_error()
    rc = lib.mmg_csr_build(None, 10, 4, 2, None, None, None, None, 0, None)
    assert rc == -1
    assert lib.mmg_csr_build_ws_bytes(1000, 100) > 3 * 4000


def test_cpu_model_fails_loudly():
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model
    g = fx.graph_from_frames(fx.det_frames(60, 9, 11, 8))
    model = build_model(CFG, (g.node_types, g.edge_types), None)
    model._init_embeddings(g)
    with pytest.raises(Exception, match="HIP device|no CPU fallback"):
        model(g)
    from mmgnn import ops
    with pytest.raises(Exception, match="HIP device"):
        ops.linear_fwd(torch.zeros(4, 64), torch.zeros(64, 64))


def test_param_count_and_state_dict_layout():
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model
    cfg = {"model": dict(CFG["model"], hidden_dim=128)}
    m = build_model(cfg, (fx.NODE_TYPES, fx.EDGE_TYPES), None)
    assert sum(p.numel() for p in m.parameters()) == 483970          # README.md:197 of the reference
    assert len(m.embeddings) == 0                                    # lazy (model.py:86-89)
    g = fx.graph_from_frames(fx.det_frames(1834, 50, 114, 100))
    m._init_embeddings(g)
    assert sum(p.numel() for p in m.parameters()) == 752514
    sd = fx.det_state({t: g[t].num_nodes for t in g.node_types}, 128)
    assert sorted(m.state_dict().keys()) == sorted(sd.keys())
    m.load_state_dict(sd)
    # PyG 2.3 key mangling 'src__rel__dst' is accepted too
    old = {}
    for k, v in sd.items():
        mt = re.match(r"(convs\.\d+\.convs\.)<(.+)>(\..+)", k)
        old[(mt.group(1) + mt.group(2).replace("___", "__") + mt.group(3)) if mt else k] = v
    assert any("patient__has_lab__lab" in k for k in old)
    m.load_state_dict(old)


def test_unknown_config_values_raise_like_the_reference():
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model, compute_regression_loss
    with pytest.raises(ValueError, match="Unknown activation"):
        build_model({"model": dict(CFG["model"], activation="gelu")}, (fx.NODE_TYPES, fx.EDGE_TYPES), None)
    with pytest.raises(ValueError, match="Unknown architecture"):
        build_model({"model": dict(CFG["model"], architecture="GAT")}, (fx.NODE_TYPES, fx.EDGE_TYPES), None)
    with pytest.raises(ValueError, match="Unknown loss type"):
        compute_regression_loss(torch.zeros(3), torch.zeros(3), "l3")
    a, b = torch.tensor([1.0, 2.0, 4.0]), torch.tensor([0.0, 2.0, 2.0])
    assert float(compute_regression_loss(a, b, "mae")) == pytest.approx(1.0)
    assert float(compute_regression_loss(a, b, "mse")) == pytest.approx(5.0 / 3)


def test_hetero_graph_container_protocol():
    import mmgnn  # noqa: F401
    from mmgnn.data import HeteroGraph
    g = HeteroGraph()
    g["patient"].num_nodes = 3
    g["lab"].num_nodes = 2
    ei = torch.tensor([[0, 1, 2], [1, 0, 1]])
    g["patient", "has_lab", "lab"].edge_index = ei
    g["patient", "has_lab", "lab"].edge_attr = torch.ones(3, 1)
    g["lab", "has_lab_rev", "patient"].edge_index = ei.flip(0)
    g.indexers = {"x": 1}
    assert g.node_types == ["patient", "lab"]
    assert g.edge_types == [("patient", "has_lab", "lab"), ("lab", "has_lab_rev", "patient")]
    assert g.metadata() == (g.node_types, g.edge_types)
    assert set(g.edge_index_dict) == set(g.edge_types)
    assert g[("patient", "has_lab", "lab")].edge_attr.shape == (3, 1) and g.indexers == {"x": 1}
    assert g.to("cpu") is g


@pytest.mark.parametrize("E", [10, 61484])
def test_edge_masker_splits_match_the_reference(E):
    """mmgnn.train.EdgeMasker (train.py:37-176) on a CPU graph: the 70/15/15 masks are the ones the REFERENCE's EdgeMasker
    produced for seed 42 (tests/golden/splits.npz), the supervision mask follows the injected generator, and the masker
    follows the graph when Trainer moves it (EdgeMasker.to)."""
    import mmgnn  # noqa: F401
    from mmgnn.data import HeteroGraph
    from mmgnn.train import EdgeMasker
    from golden_io import load, unpack_mask
    gold, _ = load("splits.npz")
    g = HeteroGraph()
    g["patient"].num_nodes = E
    g["lab"].num_nodes = 3
    ei = torch.stack([torch.arange(E), torch.arange(E) % 3])
    g["patient", "has_lab", "lab"].edge_index = ei
    g["patient", "has_lab", "lab"].edge_attr = torch.arange(E, dtype=torch.float32).unsqueeze(-1)
    m = EdgeMasker(g, 0.7, 0.15, 0.15, mask_fraction=0.2, seed=42, mask_generator=torch.Generator().manual_seed(9))
    for nm, mask in zip(("train", "val", "test"), (m.train_mask, m.val_mask, m.test_mask)):
        assert torch.equal(mask, unpack_mask(gold[f"E{E}/{nm}"], E))
    if E == 61484:
        assert [int(x.sum()) for x in (m.train_mask, m.val_mask, m.test_mask)] == [43038, 9222, 9224]
    idx, val, mask, sup = m.get_masked_data("train")
    n_tr = int(m.train_mask.sum())
    assert idx.shape == (2, n_tr) and val.shape == (n_tr,) and sup.shape == (n_tr,) and torch.equal(mask, m.train_mask)
    assert torch.equal(sup, torch.rand(n_tr, generator=torch.Generator().manual_seed(9)) < 0.2)
    assert m.get_masked_data("train")[0] is idx                       # one tensor object per split (pair-cache key)
    assert bool(m.get_masked_data("val")[3].all())
    with pytest.raises(ValueError, match="Unknown split"):
        m.get_masked_data("dev")
    with pytest.raises(AssertionError):
        EdgeMasker(g, 0.7, 0.2, 0.2)
    m.to(g, "cpu")
    assert m._cache == {} and m.edge_index is g["patient", "has_lab", "lab"].edge_index


def test_strided_job_views_of_vec_sums():
    """Host side of mmg_vec_sums' 2-D jobs: a contiguous tensor is a flat vector, a column slice of a wider matrix is
    (columns, row stride), anything else is refused (no silent copy)."""
    import mmgnn  # noqa: F401
    from mmgnn import ops
    w = torch.zeros(6, 10)
    assert ops._rows_view(w) == (0, 0) and ops._rows_view(w[2]) == (0, 0)
    assert ops._rows_view(w[:, :4]) == (4, 10) and ops._rows_view(w[:, 4:]) == (6, 10)
    with pytest.raises(ValueError):
        ops._rows_view(w.t())
    with pytest.raises(ValueError):
        ops._rows_view(w[:, ::2])


def test_deferred_weight_gradient_jobs_keep_their_order(monkeypatch):
    """wgrad_reduce_flush: jobs of different gradients share a launch; a second contribution to a gradient that an earlier
    job of the list writes waits for a later launch (and so does everything behind it for that gradient); jobs without
    slabs (small direct launches) are dropped; at most 16 jobs per launch."""
    import mmgnn  # noqa: F401
    from mmgnn import _lib, ops
    from mmgnn._lib import WgradReduceT
    calls = []

    class FakeLib:
        def mmg_wgrad_reduce_group(self, arr, n, stream):
            calls.append([(arr[i].dW, arr[i].accumulate) for i in range(n)])
            return 0

    monkeypatch.setattr(_lib, "load", lambda: FakeLib())
    monkeypatch.setattr(ops, "_stream", lambda: None)

    def job(dW, acc, slab=1):
        return (WgradReduceT(slab, 4, 2, dW, None, 4, acc), None, None, None)

    jobs = [job(100, 0), job(200, 0), job(100, 1), job(300, 0, slab=None), job(100, 1), job(400, 0)]
    ops.wgrad_reduce_flush(jobs)
    assert jobs == []
    assert calls == [[(100, 0), (200, 0), (400, 0)], [(100, 1)], [(100, 1)]]
    calls.clear()
    many = [job(1000 + i, 0) for i in range(20)]
    ops.wgrad_reduce_flush(many)
    assert [len(c) for c in calls] == [16, 4]


def test_vec_sums_refuses_two_jobs_with_one_destination():
    """The jobs of one mmg_vec_sums launch run concurrently: a shared destination would lose a contribution.  Checked on
    the host before any launch (as mmg_wgrad_reduce_group does for its gradients)."""
    import ctypes as C
    import mmgnn  # noqa: F401
    from mmgnn import _lib
    from mmgnn._lib import SumJobT
    lib = _lib.load()
    arr = (SumJobT * 2)()
    for j in range(2):
        src = (C.c_void_p * 4)(0x2000 + 0x100 * j, None, None, None)
        arr[j] = SumJobT(0x1000, src, 1, 8, 0, 0, (C.c_int * 4)(0, 0, 0, 0))
    assert lib.mmg_vec_sums(arr, 2, None) == -1 and b"share a destination" in lib.mmg_last_error()


def test_flush_grad_sums_chains_more_than_three_contributions(monkeypatch):
    """_Run.flush_grad_sums with 5 contributions to one parameter: sums go to NEW tensors (the first contribution may be
    shared by other names or belong to the caller), 3 further sources per job, and the second job -- which reads the
    first one's result -- is a later launch."""
    import mmgnn  # noqa: F401
    from mmgnn import model as mm, ops
    launches = []
    monkeypatch.setattr(ops, "wgrad_reduce_flush", lambda jobs: None)

    def fake_vec_sums(jobs):
        launches.append(len(jobs))
        for dst, srcs in jobs:
            acc = srcs[0].clone()
            for s_ in srcs[1:]:
                acc = acc + s_
            dst.copy_(acc)

    monkeypatch.setattr(ops, "vec_sums", fake_vec_sums)
    run = object.__new__(mm._Run)
    run.grads, run.pending, run.partial, run.wgrad_jobs = {}, {}, set(), []
    parts = [torch.full((4, 3), float(i + 1)) for i in range(5)]
    shared = torch.full((2,), 10.0)
    for p_ in parts:
        run.acc("w", p_)
    run.acc("a", shared); run.acc("b", shared); run.acc("b", torch.ones(2))
    run.flush_grad_sums()
    assert torch.equal(run.grads["w"], torch.full((4, 3), 15.0))
    assert torch.equal(parts[0], torch.full((4, 3), 1.0))                   # the first contribution is not overwritten
    assert torch.equal(run.grads["b"], torch.full((2,), 11.0)) and torch.equal(run.grads["a"], torch.full((2,), 10.0))
    assert run.grads["a"] is shared and launches == [2, 1] and run.pending == {}
