"""Device-side evaluation reducers (mmg_seg_moments / mmg_seg_metrics, include/mmgnn.h) against their checker, the host
reducers of mmgnn.evaluate (numpy, golden-pinned against the reference's evaluate.py in tests/test_evaluate_cpu.py):
per-lab +-3 sigma winsorisation, overall / per-lab / stratified metrics, and evaluate_model end to end on the GPU."""
import json
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as fx


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _close(a, b, tol):
    if isinstance(b, float) and math.isnan(b):
        return math.isnan(a)
    return abs(a - b) <= tol * max(1.0, abs(b))


@pytest.mark.parametrize("n,n_labs", [(9224, 50), (1_000_003, 50), (37, 5), (0, 4)])
def test_segment_reducers_match_the_host_reducers(dev, n, n_labs):
    import mmgnn  # noqa: F401
    from mmgnn import evaluate as ev, ops
    gen = torch.Generator().manual_seed(n + 1)
    lab = torch.randint(0, n_labs, (n,), generator=gen)
    if n > 30:
        lab[lab == 3] = 2                                            # an empty segment
        lab[7] = 3                                                   # ... and one with a single sample (left unclipped)
    target = torch.randn(n, generator=gen)
    target[::17] = 0.0                                               # MAPE skips zero targets
    pred = target + 0.5 * torch.randn(n, generator=gen) * (1 + (torch.rand(n, generator=gen) < 0.01) * 20)   # outliers
    sums, adj = ops.seg_sums(pred.to(dev), target.to(dev), lab.to(dev), n_labs, 3.0, want_adjusted=True)
    s = sums.cpu().numpy()
    p_np, t_np, l_np = pred.numpy(), target.numpy(), lab.numpy()
    adj_np, capped = ev.winsorize_residuals(p_np, t_np, l_np)
    # a residual within one fp32 ulp of its segment's bound may fall on either side (the bound is formed from fp64 sums
    # here and from fp32 pairwise sums in numpy): allow a handful of such ties, and 1e-6 on the adjusted values
    assert abs(int(round(s[:, 7].sum())) - capped) <= 2 + n // 200000
    if n:
        assert float(np.max(np.abs(adj.cpu().numpy() - adj_np))) <= 2e-6 * max(1.0, float(np.abs(adj_np).max()))
    want = ev.compute_regression_metrics(adj_np, t_np) if n else None
    got = ev.metrics_from_sums(s.sum(0))
    if n:
        for k in ("mae", "rmse", "r2", "mape"):
            assert _close(got[k], want[k], 2e-6), (k, got[k], want[k])
        df = ev.compute_per_lab_metrics(adj_np, t_np, l_np, {})
        for _, row in df.iterrows():
            j = int(row["lab_index"])
            m = ev.metrics_from_sums(s[j])
            assert int(s[j, 0]) == int(row["num_samples"])
            for k in ("mae", "rmse", "r2", "mape"):
                assert _close(m[k], float(row[k]), 5e-6), (j, k)
        assert int(s[3, 0]) == 1 if n > 30 else True
    else:
        assert float(s.sum()) == 0.0 and math.isnan(got["mae"])
    # no clipping requested: plain sums
    s0, _ = ops.seg_sums(pred.to(dev), target.to(dev), lab.to(dev), n_labs, 0.0)
    if n:
        w0 = ev.compute_regression_metrics(p_np, t_np)
        g0 = ev.metrics_from_sums(s0.cpu().numpy().sum(0))
        for k in ("mae", "rmse", "r2", "mape"):
            assert _close(g0[k], w0[k], 2e-6), k
        assert float(s0[:, 7].sum()) == 0.0


def test_evaluate_model_on_the_device_matches_the_host_path(dev, tmp_path):
    """evaluate_model (evaluate.py:349-570) with a HIP model: same result dict, json and csv as the host reducers give
    for the same predictions."""
    import mmgnn  # noqa: F401
    from mmgnn import evaluate as ev
    from mmgnn.model import build_model
    from oracle import model as om, train as ot
    cfg = {"model": {"architecture": "RGCN", "hidden_dim": 64, "num_layers": 2, "dropout": 0.0,
                     "use_batch_norm": True, "activation": "relu"},
           "evaluation": {"per_lab_metrics": True, "baselines": [], "stratify_by": ["num_labs", "lab_frequency"]}}
    g = fx.graph_from_frames(fx.det_frames(700, 20, 25, 18))
    gv = om.GraphView(g)
    sd = fx.det_state(gv.num_nodes, 64)
    model = build_model(cfg, (g.node_types, g.edge_types), None).to(dev)
    gd = g.clone().to(dev)
    model._init_embeddings(gd)
    model.load_state_dict(sd)
    ei, ea = g["patient", "has_lab", "lab"].edge_index, g["patient", "has_lab", "lab"].edge_attr
    _, _, te = ot.edge_splits(ei.shape[1])
    test_edges = (ei[:, te], ea[te].squeeze(-1))
    res = ev.evaluate_model(model, gd, test_edges, cfg, tmp_path)
    model.eval()
    with torch.no_grad():
        pred = model.predict_lab_values(gd, ei[0][te].to(dev), ei[1][te].to(dev)).cpu().numpy()
    t_np, l_np, p_np = ea[te].squeeze(-1).numpy(), ei[1][te].numpy(), ei[0][te].numpy()
    adj, capped = ev.winsorize_residuals(pred, t_np, l_np)
    want = ev.compute_regression_metrics(adj, t_np)
    assert res["num_test_samples"] == int(te.sum())
    for k in ("mae", "rmse", "r2", "mape"):
        assert _close(res["overall_metrics"][k], want[k], 5e-6), k
    w_deg = ev.stratify_by_patient_degree(adj, t_np, p_np, g)
    w_frq = ev.stratify_by_lab_frequency(adj, t_np, l_np, g)
    for got, wnt in ((res["stratified_results"]["by_patient_degree"], w_deg), (res["stratified_results"]["by_lab_frequency"], w_frq)):
        assert set(got) == set(wnt)
        for name in wnt:
            assert got[name]["num_samples"] == wnt[name]["num_samples"]
            for k in ("mae", "rmse", "r2", "mape"):
                assert _close(got[name][k], wnt[name][k], 1e-5), (name, k)
    assert json.load(open(tmp_path / "evaluation_results.json"))["overall_metrics"] == res["overall_metrics"]
    import pandas as pd
    df = pd.read_csv(tmp_path / "per_lab_metrics.csv")
    wdf = ev.compute_per_lab_metrics(adj, t_np, l_np, {})
    assert sorted(df["lab_index"].tolist()) == sorted(wdf["lab_index"].tolist())
    assert list(df.columns) == list(wdf.columns)
