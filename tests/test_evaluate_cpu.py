"""mmgnn.evaluate against the REFERENCE's evaluate.py outputs (tests/golden/eval_small.npz + metrics.npz, written by
oracle/gen_golden.py): winsorisation, overall / per-lab / stratified metrics, baselines, and the files written."""
import json
import math

import numpy as np
import pandas as pd
import pytest
import torch

import mmgnn  # noqa: F401
from mmgnn import evaluate as ev
from oracle import fixtures as fx
from golden_io import load, t
from test_graph_build_cpu import CFG as GRAPH_CFG, frames_to_pandas

CFG = {"evaluation": {"per_lab_metrics": True, "baselines": ["global_mean"],
                      "stratify_by": ["age_group", "num_labs", "lab_frequency"]}}
TOL = 1e-12


def close(a, b, tol=TOL):
    if isinstance(b, float) and math.isnan(b):
        return math.isnan(a)
    return abs(a - b) <= tol * max(1.0, abs(b))


class FixedModel(torch.nn.Module):
    def __init__(self, pred):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.pred = pred

    def predict_lab_values(self, graph, p, l):
        return self.pred.clone()


def test_regression_metrics_golden():
    gold, meta = load("metrics.npz")
    m = ev.compute_regression_metrics(gold["pred"], gold["target"])
    for k, v in meta["metrics"].items():
        assert close(m[k], v), k
    # degenerate cases: constant target, all-zero target
    assert ev.compute_regression_metrics(np.ones(4), np.ones(4))["r2"] == 1.0
    assert ev.compute_regression_metrics(np.zeros(4), np.ones(4))["r2"] == 0.0
    assert math.isnan(ev.compute_regression_metrics(np.ones(4), np.zeros(4))["mape"])


def test_regression_metrics_vs_sklearn_fp32():
    sk = pytest.importorskip("sklearn.metrics")
    p = fx.det_uniform((4001,), 31, -3, 3).numpy()
    y = fx.det_uniform((4001,), 32, -3, 3).numpy()
    m = ev.compute_regression_metrics(p, y)
    assert m["mae"] == float(sk.mean_absolute_error(y, p))
    assert m["rmse"] == float(np.sqrt(sk.mean_squared_error(y, p)))
    assert m["r2"] == float(sk.r2_score(y, p))


def test_evaluate_model_end_to_end_golden(tmp_path):
    gold, meta = load("eval_small.npz")
    from mmgnn import graph_build as gb
    g = gb.build_heterogeneous_graph(*frames_to_pandas(fx.det_frames(*meta["frames"])), GRAPH_CFG)   # carries lab names
    ei = g["patient", "has_lab", "lab"].edge_index
    sel = t(gold["sel"])
    res = ev.evaluate_model(FixedModel(t(gold["pred"])), g, (ei[:, sel], t(gold["target"])), CFG, tmp_path)
    want = meta["results"]
    assert res["num_test_samples"] == want["num_test_samples"]
    for k, v in want["overall_metrics"].items():
        assert close(res["overall_metrics"][k], v), k
    assert list(res["stratified_results"]) == list(want["stratified_results"])
    for strat, groups in want["stratified_results"].items():
        assert list(res["stratified_results"][strat]) == list(groups)
        for name, m in groups.items():
            for k, v in m.items():
                assert close(res["stratified_results"][strat][name][k], v), (strat, name, k)
    on_disk = json.load(open(tmp_path / "evaluation_results.json"))
    assert on_disk["overall_metrics"]["mae"] == res["overall_metrics"]["mae"]
    df = pd.read_csv(tmp_path / "per_lab_metrics.csv")
    assert df["lab_index"].tolist() == gold["per_lab/lab_index"].tolist()          # same MAE ordering
    assert df["num_samples"].tolist() == gold["per_lab/num_samples"].tolist()
    assert df["lab_name"].tolist() == meta["per_lab_names"]
    for c in ("mae", "rmse", "r2", "mape"):
        np.testing.assert_allclose(df[c].to_numpy(), gold["per_lab/" + c], rtol=1e-12, atol=0)


def test_winsorize_caps_and_counts():
    gold, meta = load("eval_small.npz")
    g = fx.graph_from_frames(fx.det_frames(*meta["frames"]))
    lab = g["patient", "has_lab", "lab"].edge_index[1, t(gold["sel"])].numpy()
    p, n = ev.winsorize_residuals(gold["pred"], gold["target"], lab)
    assert p.dtype == np.float32 and n > 0 and n == int((p != gold["pred"]).sum())
    for j in np.unique(lab):
        m = lab == j
        r = gold["pred"][m] - gold["target"][m]
        assert np.all(np.abs((p[m] - gold["target"][m]) - r.mean()) <= 3 * r.std() * (1 + 1e-5) + 1e-6)
    # a lab with a single sample is left alone
    p1, n1 = ev.winsorize_residuals(np.array([9.0, 1.0, 1.1], np.float32), np.zeros(3, np.float32), np.array([0, 1, 1]))
    assert p1[0] == 9.0 and n1 == 0


def test_baselines_golden():
    gold, meta = load("eval_small.npz")
    g = fx.graph_from_frames(fx.det_frames(*meta["frames"]))
    ei = g["patient", "has_lab", "lab"].edge_index
    ea = g["patient", "has_lab", "lab"].edge_attr.squeeze().numpy().astype(np.float64)
    sel = gold["sel"]
    res = ev.evaluate_baselines((ea, ei[1].numpy()), (gold["target"].astype(np.float64), ei[1].numpy()[sel], None))
    for name, m in meta["baselines"].items():
        for k, v in m.items():
            assert close(res[name][k], v), (name, k)


def test_segment_metrics_for_a_single_sample_group():
    """A stratification group / lab with ONE sample: sklearn's r2_score is undefined there (nan), MAE / RMSE are that
    sample's error; an empty group is all nan (evaluate.py:36-82 through sklearn)."""
    import math
    import mmgnn  # noqa: F401
    from mmgnn.evaluate import metrics_from_sums
    one = metrics_from_sums([1.0, 0.5, 0.25, 2.0, 4.0, 0.25, 1.0, 0.0])
    assert math.isnan(one["r2"]) and one["mae"] == 0.5 and one["rmse"] == 0.5 and one["mape"] == 25.0
    none = metrics_from_sums([0.0] * 8)
    assert all(math.isnan(v) for v in none.values())
    two = metrics_from_sums([2.0, 1.0, 0.5, 3.0, 5.0, 0.7, 2.0, 0.0])       # t = (1, 2): ss_tot = 0.5, ss_res = 0.5
    assert two["r2"] == 0.0
