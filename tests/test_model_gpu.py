"""End-to-end parity of the HIP path (through the C ABI) against
  (a) the golden vectors produced by the REFERENCE's own host code (tests/golden/model_*.npz) and
  (b) the CPU oracle on the same inputs (eval, train with p=0, train with the device's dropout
      masks injected into the oracle).
Bar (BASELINE.json): predictions within 1e-4 relative fp32; gradients within 2e-4 of max|grad|.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_io import grad_close, checksum, load, rel_close, rel_err, t, unpack_mask
from oracle import fixtures as fx
from oracle import model as om
from oracle import train as ot

TOL = 1e-4
CFG = {"model": {"architecture": "RGCN", "hidden_dim": 128, "num_layers": 2, "dropout": 0.0,
                 "use_batch_norm": True, "activation": "relu"}}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def make(dev, n, hidden, dropout=0.0, **model_kw):
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model
    g = fx.graph_from_frames(fx.det_frames(*n))
    gv = om.GraphView(g)
    cfg = {"model": dict(CFG["model"], hidden_dim=hidden, dropout=dropout, **model_kw)}
    sd = fx.det_state(gv.num_nodes, hidden, num_layers=cfg["model"]["num_layers"])
    if not cfg["model"]["use_batch_norm"]:
        sd = {k: v for k, v in sd.items() if not k.startswith("batch_norms.")}
    model = build_model(cfg, (g.node_types, g.edge_types), None).to(dev)
    gd = g.clone().to(dev)
    model._init_embeddings(gd)
    model.load_state_dict(sd, strict=True)
    ei, ea = g["patient", "has_lab", "lab"].edge_index, g["patient", "has_lab", "lab"].edge_attr
    return model, g, gd, gv, sd, ei, ea


CASES = [("small", (300, 12, 15, 10), 64), ("eicu", (1834, 50, 114, 100), 128)]


@pytest.mark.parametrize("tag,n,hidden", CASES)
def test_state_dict_layout_matches_reference(dev, tag, n, hidden):
    gold, meta = load(f"model_{tag}.npz")
    model, *_ = make(dev, n, hidden)
    assert list(model.state_dict().keys()) == meta["state_keys"]
    assert sum(p.numel() for p in model.parameters()) == meta["params_after"]


@pytest.mark.parametrize("tag,n,hidden", CASES)
def test_eval_matches_reference_golden(dev, tag, n, hidden):
    gold, meta = load(f"model_{tag}.npz")
    model, g, gd, gv, sd, ei, ea = make(dev, n, hidden)
    tr, va, te = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    model.eval()
    with torch.no_grad():
        enc = model.encode_nodes(gd)
        fwd = model(gd)
        pred = model.predict_lab_values(gd, ei[0][te].to(dev), ei[1][te].to(dev))
    assert rel_err(pred.cpu(), t(gold["eval/pred_test"])) <= TOL
    ok, worst = rel_close(pred.cpu(), t(gold["eval/pred_test"]))          # 1e-4 relative PER VALUE (atol 1e-6 max|ref|)
    assert ok, f"element-wise relative error {worst:.2f} x the 1e-4 bar"
    for nt in gv.node_types:
        if f"eval/enc/{nt}" in gold:
            assert rel_err(enc[nt].cpu(), t(gold[f"eval/enc/{nt}"])) <= TOL
            assert rel_err(fwd[nt].cpu(), t(gold[f"eval/fwd/{nt}"])) <= TOL
        else:
            st = max(1, enc[nt].shape[0] // 16)
            assert rel_err(enc[nt].cpu()[::st][:16], t(gold[f"eval/enc_rows/{nt}"])) <= TOL
            assert rel_err(fwd[nt].cpu()[::st][:16], t(gold[f"eval/fwd_rows/{nt}"])) <= TOL
        assert rel_err(checksum(fwd[nt].cpu()), t(gold[f"eval/fwd_sum/{nt}"])) <= TOL
    # eval must not touch the BN buffers
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert torch.equal(v.cpu(), sd[k]), k


@pytest.mark.parametrize("tag,n,hidden", CASES)
def test_train_step_matches_reference_golden(dev, tag, n, hidden):
    gold, meta = load(f"model_{tag}.npz")
    model, g, gd, gv, sd, ei, ea = make(dev, n, hidden)
    tr, va, te = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = t(gold["lab_weights"])
    sup = unpack_mask(gold["train/sup_mask"], int(tr.sum()))
    model.train()
    pred = model.predict_lab_values(gd, pi.to(dev), li.to(dev))
    assert rel_err(checksum(pred.detach().cpu()), t(gold["train/pred_train_sum"])) <= TOL
    loss = ((pred[sup.to(dev)] - y[sup].to(dev)).abs() * w[li[sup]].to(dev)).mean()       # train.py:366-386
    assert abs(float(loss) - meta["train_loss"]) <= TOL * abs(meta["train_loss"])
    loss.backward()
    gmax = max(float(t(gold[k]).abs().max()) for k in gold if k.startswith("train/grad/"))
    for k, p in model.named_parameters():
        gr = p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        if f"train/grad/{k}" in gold:
            ref = t(gold[f"train/grad/{k}"])
            assert float((gr - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6 * gmax, k
            ok, worst, _ = grad_close(gr, ref, floor=1e-6 * gmax)                       # ... and per element at the scale of its own row
            assert ok, f"{k}: element-wise gradient error {worst:.2f} x the bar"
        cs, cr = checksum(gr), t(gold[f"train/grad_sum/{k}"])
        assert abs(float(cs[1] - cr[1])) <= 2e-4 * float(cr[1]) + 1e-6 * gmax * gr.numel(), k
    for k, b in model.named_buffers():
        ref = t(gold[f"train/buf/{k}"])
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(ref), k
        else:
            assert rel_err(b.cpu(), ref) <= TOL, k


def _oracle_masks(ops, dev, seed, p, gv, n_pairs, D, num_layers=2):
    P = gv.num_nodes["patient"]
    m = {}
    for c in (0, 1):
        m[f"enc{c}.drop0"] = ops.dropout_mask(seed, 2 * c, P, D, p, dev).cpu().float()
        m[f"enc{c}.drop1"] = ops.dropout_mask(seed, 2 * c + 1, P, D, p, dev).cpu().float()
    for l in range(num_layers - 1):                 # dropout after every conv layer but the last (model.py:267-269)
        for ti, nt in enumerate(gv.node_types):
            m[f"conv{l}.{nt}"] = ops.dropout_mask(seed, 16 + 8 * l + ti, gv.num_nodes[nt], D, p, dev).cpu().float()
    m["head.drop0"] = ops.dropout_mask(seed, 64, n_pairs, 64, p, dev).cpu().float()
    m["head.drop1"] = ops.dropout_mask(seed, 65, n_pairs, 32, p, dev).cpu().float()
    return m


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_train_step_matches_oracle_with_injected_dropout(dev, p):
    from mmgnn import ops
    n, hidden = (500, 20, 25, 18), 128
    model, g, gd, gv, sd, ei, ea = make(dev, n, hidden, dropout=p)
    tr, va, te = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    sup = ot.supervision_mask(int(tr.sum()), 0.2, torch.Generator().manual_seed(7))
    seed = 424242
    model._dropout_seed = seed
    model.train()
    pred = model.predict_lab_values(gd, pi.to(dev), li.to(dev))
    loss = ((pred[sup.to(dev)] - y[sup].to(dev)).abs() * w[li[sup]].to(dev)).mean()
    loss.backward()
    masks = _oracle_masks(ops, dev, seed, p, gv, pi.numel(), hidden) if p > 0 else None
    oloss, opred, ograds, obufs = ot.train_step_grads(sd, gv, pi, li, y, w, sup, p=p, masks=masks)
    assert rel_err(pred.detach().cpu(), opred) <= TOL
    assert abs(float(loss) - float(oloss)) <= TOL * abs(float(oloss))
    gmax = max(float(v.abs().max()) for v in ograds.values())
    for k, pm in model.named_parameters():
        gr = pm.grad.cpu() if pm.grad is not None else torch.zeros_like(pm).cpu()
        ref = ograds[k]
        assert float((gr - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6 * gmax, k
    for k, b in model.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(obufs[k]), k
        else:
            assert rel_err(b.cpu(), obufs[k]) <= TOL, k


@pytest.mark.parametrize("thr", [0, 10 ** 6])
def test_degree_gate_extremes_in_training(dev, thr):
    """No low-degree patient at all (the first encoder pass then has an EMPTY row list behind its last BatchNorm) and
    only low-degree patients (the list is every row): the degree gate of model.py:312 at its two ends, with dropout."""
    from mmgnn import ops
    n, hidden, p = (500, 20, 25, 18), 128, 0.2
    model, g, gd, gv, sd, ei, ea = make(dev, n, hidden, dropout=p)
    model.degree_threshold = thr
    tr, va, te = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    sup = ot.supervision_mask(int(tr.sum()), 0.2, torch.Generator().manual_seed(7))
    seed = 777
    model._dropout_seed = seed
    model.train()
    pred = model.predict_lab_values(gd, pi.to(dev), li.to(dev))
    loss = ((pred[sup.to(dev)] - y[sup].to(dev)).abs() * w[li[sup]].to(dev)).mean()
    loss.backward()
    masks = _oracle_masks(ops, dev, seed, p, gv, pi.numel(), hidden)
    oloss, opred, ograds, obufs = ot.train_step_grads(sd, gv, pi, li, y, w, sup, p=p, masks=masks, degree_threshold=thr)
    assert rel_err(pred.detach().cpu(), opred) <= TOL
    gmax = max(float(v.abs().max()) for v in ograds.values())
    for k, pm in model.named_parameters():
        gr = pm.grad.cpu() if pm.grad is not None else torch.zeros_like(pm).cpu()
        ref = ograds[k]
        assert float((gr - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6 * gmax, k
    for k, b in model.named_buffers():
        if not k.endswith("num_batches_tracked"):
            assert rel_err(b.cpu(), obufs[k]) <= TOL, k


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_train_step_with_side_stream_overlap(dev, p, monkeypatch):
    """The vocab-side work of every layer on a side stream (on by default only above 16 k patients): same results."""
    import mmgnn.model as mm
    monkeypatch.setattr(mm, "OVERLAP_MODE", "on")
    test_train_step_matches_oracle_with_injected_dropout(dev, p)
    n, hidden = (500, 20, 25, 18), 128
    model, g, gd, *_ = make(dev, n, hidden)
    assert mm._Run(model, gd).overlap


@pytest.mark.parametrize("hidden", [64, 256])
def test_other_hidden_dims_match_oracle(dev, hidden):
    """BASELINE.json config 4 runs 256-d; the reference's own smoke test builds a 64-d model (model.py:642-648)."""
    # shape picked kink-free: at (600,30,40,25)/256-d one patient_transform ReLU input sits within fp32 rounding of 0
    # and its gradient flips with the summation order (fp32 and fp64 oracles disagree with each other there too)
    n = (640, 30, 40, 25)
    model, g, gd, gv, sd, ei, ea = make(dev, n, hidden)
    tr, va, te = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    sup = ot.supervision_mask(int(tr.sum()), 0.2, torch.Generator().manual_seed(11))
    model.train()
    pred = model.predict_lab_values(gd, pi.to(dev), li.to(dev))
    loss = ((pred[sup.to(dev)] - y[sup].to(dev)).abs() * w[li[sup]].to(dev)).mean()
    loss.backward()
    oloss, opred, ograds, obufs = ot.train_step_grads(sd, gv, pi, li, y, w, sup, p=0.0)
    assert rel_err(pred.detach().cpu(), opred) <= TOL
    gmax = max(float(v.abs().max()) for v in ograds.values())
    for k, pm in model.named_parameters():
        gr = pm.grad.cpu() if pm.grad is not None else torch.zeros_like(pm).cpu()
        assert float((gr - ograds[k]).abs().max()) <= 2e-4 * float(ograds[k].abs().max()) + 1e-6 * gmax, k


def _train_step_vs_oracle(dev, n, hidden, p=0.0, sup_seed=11, **model_kw):
    """One training step of the HIP model against the oracle on the same inputs (dropout 0 or injected masks):
    predictions 1e-4 (max-relative AND per value), loss 1e-4, every gradient 2e-4 of its max, BatchNorm buffers 1e-4."""
    from mmgnn import ops
    model, g, gd, gv, sd, ei, ea = make(dev, n, hidden, dropout=p, **model_kw)
    L = model_kw.get("num_layers", 2)
    tr, va, te = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    sup = ot.supervision_mask(int(tr.sum()), 0.2, torch.Generator().manual_seed(sup_seed))
    seed = 424242
    model._dropout_seed = seed
    model.train()
    pred = model.predict_lab_values(gd, pi.to(dev), li.to(dev))
    loss = ((pred[sup.to(dev)] - y[sup].to(dev)).abs() * w[li[sup]].to(dev)).mean()
    loss.backward()
    masks = _oracle_masks(ops, dev, seed, p, gv, pi.numel(), hidden, L) if p > 0 else None
    oloss, opred, ograds, obufs = ot.train_step_grads(
        sd, gv, pi, li, y, w, sup, p=p, masks=masks, num_layers=L,
        use_batch_norm=model_kw.get("use_batch_norm", True), activation=model_kw.get("activation", "relu"))
    assert rel_err(pred.detach().cpu(), opred) <= TOL
    ok, worst = rel_close(pred.detach().cpu(), opred)
    assert ok, f"element-wise relative error {worst:.2f} x the 1e-4 bar"
    assert abs(float(loss) - float(oloss)) <= TOL * abs(float(oloss))
    # ReLU ties: a pre-activation within fp32 rounding of 0 sends its gradient to one side or the other depending on the
    # summation order -- the oracle's own fp32 and fp64 runs disagree there (e.g. patient row 1179 of the eICU-shape
    # fixture at 256-d).  Where they do, the gap between the two oracle runs is added to the tolerance; everywhere else
    # the bar is 2e-4 of the gradient's max.
    sd64 = om.cast_state(sd, torch.float64)
    masks64 = {k: v.double() for k, v in masks.items()} if masks else None
    _, _, ograds64, _ = ot.train_step_grads(
        sd64, gv, pi, li, y.double(), w.double(), sup, p=p, masks=masks64, num_layers=L,
        use_batch_norm=model_kw.get("use_batch_norm", True), activation=model_kw.get("activation", "relu"))
    gmax = max(float(v.abs().max()) for v in ograds.values())
    n_tied, n_off64, n_el = 0, 0, 0
    for k, pm in model.named_parameters():
        gr = pm.grad.cpu() if pm.grad is not None else torch.zeros_like(pm).cpu()
        tie = (ograds[k].double() - ograds64[k]).abs()
        # a tie only counts where the two oracle runs disagree by more than rounding (>= 1e-5 of the tensor's max): the
        # slack is for flipped ReLU gradients, not a blanket allowance of the fp32 oracle's own rounding error
        tie = torch.where(tie > 1e-5 * float(ograds[k].abs().max()), tie, torch.zeros_like(tie))
        tol = 2e-4 * float(ograds[k].abs().max()) + 1e-6 * gmax + 1.5 * tie
        assert bool(((gr - ograds[k]).abs().double() <= tol).all()), k
        ok, worst, nt = grad_close(gr, ograds[k], tie=1.5 * tie, floor=1e-6 * gmax)     # per element, at the scale of the element's row
        assert ok, f"{k}: element-wise gradient error {worst:.2f} x the bar"
        n_tied += nt
        n_off64 += grad_close(gr, ograds64[k], tie=torch.zeros_like(tie), floor=1e-6 * gmax)[2]
        n_el += gr.numel()
    # The tie slack is an exception, not a tolerance.  Elements that pass ONLY because of it are counted; they must be a
    # handful (0 in every configuration but the 256-d eICU-vocabulary step) -- unless the device took the fp64 oracle's side
    # of the tie, in which case it has to match THAT run per element with no slack at all (one flipped unit moves a
    # rank-one slice of every weight gradient upstream and, through the BatchNorm mean, a little of every patient row).
    assert n_tied <= max(64, n_el // 2000) or n_off64 == 0, \
        f"{n_tied} of {n_el} gradient elements pass only through the ReLU-tie slack and {n_off64} miss the fp64 oracle's bar"
    for k, b in model.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(obufs[k]), k
        else:
            assert rel_err(b.cpu(), obufs[k]) <= TOL, k
    return model


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_config4_eicu_vocabulary_at_256d(dev, p):
    """BASELINE.json config 4's model: the eICU vocabulary (1,834 / 50 / 114 / 100) at 256-d -- the bit-plane aggregates
    with eight 32-column strips / two 128-column chunks and the K = 256 dense kernels, end to end against the oracle."""
    _train_step_vs_oracle(dev, (1834, 50, 114, 100), 256, p=p, sup_seed=3)


@pytest.mark.parametrize("sites,num_layers", [((), 2), (("heads", "conv", "enc2", "enc1"), 2), (("heads", "conv", "enc2", "enc1"), 3)])
def test_batchnorm_backward_statistics_from_producer_epilogues(dev, sites, num_layers, monkeypatch):
    """mmgnn.model.NEXT_BN_SITES: every BatchNorm-backward statistics pass taken from the epilogue of the kernel that
    produces its upstream gradient (all four sites, the default leaves one out) and none of them -- the same step against
    the oracle at the eICU vocabulary / 128-d (the fused kernels need more than 512 patients); three layers put the
    three-relation gather in front of a BatchNorm as well."""
    import mmgnn.model as mm
    monkeypatch.setattr(mm, "NEXT_BN_SITES", frozenset(sites))
    _train_step_vs_oracle(dev, (1834, 50, 114, 100), 128, p=0.2, sup_seed=3, num_layers=num_layers)


def test_two_models_of_one_process_run_with_their_own_execution_switches(dev):
    """HeteroRGCN.configure_execution: side stream, next-BatchNorm sites and saved pair state are per MODEL (the module-level
    switches are defaults) -- two models configured differently give the same predictions and gradients (to summation order:
    the producer-epilogue statistics are taken in another order than the separate pass) and each run reads ITS settings."""
    import mmgnn.model as mm
    n, hidden = (1834, 50, 114, 100), 128
    outs = []
    for kw in (dict(overlap="on", next_bn=("heads", "conv", "enc2", "enc1"), save_pair_state=True),
               dict(overlap="off", next_bn=(), save_pair_state=False)):
        model, g, gd, gv, sd, ei, ea = make(dev, n, hidden, dropout=0.2)
        model.configure_execution(**kw)
        run = mm._Run(model, gd)
        assert run.overlap == (kw["overlap"] == "on") and run.next_bn_sites == frozenset(kw["next_bn"])
        assert run.save_pair_state == kw["save_pair_state"]
        model.train()
        model._seed_dev = None
        torch.manual_seed(5)
        pi, li = ei[0].to(dev), ei[1].to(dev)
        pred = model.predict_lab_values(gd, pi, li)
        (pred * torch.linspace(-1, 1, pred.numel(), device=dev)).sum().backward()
        outs.append((pred.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    with pytest.raises(ValueError):
        model.configure_execution(overlap="sometimes")
    (p0, g0), (p1, g1) = outs
    assert torch.equal(p0, p1)
    gmax = max(float(v.abs().max()) for v in g1.values())      # (a bias in front of a BatchNorm has gradient 0 + rounding noise)
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * float(g1[k].abs().max()) + 2e-6 * gmax, k


@pytest.mark.parametrize("activation", ["elu", "leaky_relu"])
def test_activation_variants_match_oracle(dev, activation):
    """model.py:145-152 accepts relu / elu / leaky_relu for the conv layers."""
    _train_step_vs_oracle(dev, (500, 20, 25, 18), 128, p=0.2, activation=activation)


def test_without_batch_norm_matches_oracle(dev):
    """use_batch_norm = False (model.py:259-261): activation + dropout straight on the HeteroConv sums."""
    m = _train_step_vs_oracle(dev, (500, 20, 25, 18), 128, p=0.2, use_batch_norm=False)
    assert m.batch_norms is None


@pytest.mark.parametrize("num_layers", [1, 3])
def test_other_layer_counts_match_oracle(dev, num_layers):
    """num_layers != 2 (conf/config.yaml model.num_layers): dropout after every layer but the last (model.py:267-269)."""
    _train_step_vs_oracle(dev, (500, 20, 25, 18), 128, p=0.2, num_layers=num_layers)


def test_large_vocabulary_without_bit_planes_matches_oracle(dev):
    """A vocabulary beyond the matrix-core aggregates' 768 padded items (900 diagnosis codes): no adjacency bit planes
    are allocated, the CSR kernels (tables through L2 / global float atomics) carry the aggregates -- same results."""
    from mmgnn.data import build_plan
    m = _train_step_vs_oracle(dev, (400, 30, 900, 40), 128, p=0.0)
    g = fx.graph_from_frames(fx.det_frames(400, 30, 900, 40)).to(dev)
    plan = build_plan(g, dev, use_cache=False)
    assert all(r.mask_t is None and r.mask_r is None for r in plan.rels.values())


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_mimic_schema_vocabulary_at_128d_with_dropout(dev, p):
    """BASELINE.json config 5 at its own width: 50 / 200 / 100 vocabulary, 128-d, dropout 0.2 (injected masks) -- the
    unit-per-wave aggregates (k_gather_units / k_scatter_units) with their dropout and row-scale paths, per element."""
    _train_step_vs_oracle(dev, (900, 50, 200, 100), 128, p=p, sup_seed=5)


def test_mimic_schema_vocabulary_matches_oracle(dev):
    """BASELINE.json config 5: the MIMIC-III schema keeps the top 50 labs / 200 diagnoses / 100 medications
    (conf/config.yaml:70,101,112) -- a vocabulary layout other than eICU's (64 | 224 | 128 padded item rows)."""
    n = (900, 50, 200, 100)
    model, g, gd, gv, sd, ei, ea = make(dev, n, 128)
    tr, va, te = ot.edge_splits(ei.shape[1], 0.7, 0.15, 0.15, 42)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    sup = ot.supervision_mask(int(tr.sum()), 0.2, torch.Generator().manual_seed(5))
    model.train()
    pred = model.predict_lab_values(gd, pi.to(dev), li.to(dev))
    loss = ((pred[sup.to(dev)] - y[sup].to(dev)).abs() * w[li[sup]].to(dev)).mean()
    loss.backward()
    oloss, opred, ograds, obufs = ot.train_step_grads(sd, gv, pi, li, y, w, sup, p=0.0)
    assert rel_err(pred.detach().cpu(), opred) <= TOL
    gmax = max(float(v.abs().max()) for v in ograds.values())
    for k, pm in model.named_parameters():
        gr = pm.grad.cpu() if pm.grad is not None else torch.zeros_like(pm).cpu()
        assert float((gr - ograds[k]).abs().max()) <= 2e-4 * float(ograds[k].abs().max()) + 1e-6 * gmax, k


def test_forward_and_encode_are_differentiable(dev):
    model, g, gd, gv, sd, ei, ea = make(dev, (300, 12, 15, 10), 64)
    model.train()
    out = model(gd)
    (out["patient"].sum() + 2 * out["lab"].sum()).backward()
    leaf = {k: v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v
            for k, v in sd.items()}
    o = om.forward(leaf, gv, training=True, p=0.0)
    (o["patient"].sum() + 2 * o["lab"].sum()).backward()
    gmax = max(float(v.grad.abs().max()) for v in leaf.values() if torch.is_tensor(v) and v.requires_grad and v.grad is not None)
    for k, pm in model.named_parameters():
        ref = leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k])
        gr = pm.grad.cpu() if pm.grad is not None else torch.zeros_like(ref)
        assert float((gr - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6 * gmax, k
    model.zero_grad()
    enc = model.encode_nodes(gd)
    enc["patient"].pow(2).sum().backward()
    assert model.embeddings["patient"].weight.grad.abs().sum() > 0


def test_isolated_and_low_degree_patients_use_the_tabular_head(dev):
    model, g, gd, gv, sd, ei, ea = make(dev, (300, 12, 15, 10), 64)
    model.eval()
    deg = torch.bincount(ei[0], minlength=300)
    low_p = torch.nonzero((deg > 0) & (deg < 6)).flatten()
    assert len(low_p) > 0
    pi = low_p.repeat_interleave(2)
    li = torch.arange(len(pi)) % gv.num_nodes["lab"]
    with torch.no_grad():
        pred = model.predict_lab_values(gd, pi.to(dev), li.to(dev))
        ref, _ = om.predict_lab_values(sd, gv, pi, li)
    assert rel_err(pred.cpu(), ref) <= TOL


def test_empty_pair_list_and_missing_embeddings(dev):
    model, g, gd, gv, sd, ei, ea = make(dev, (120, 8, 9, 7), 64)
    model.eval()
    with torch.no_grad():
        out = model.predict_lab_values(gd, torch.empty(0, dtype=torch.long, device=dev),
                                       torch.empty(0, dtype=torch.long, device=dev))
    assert out.shape == (0,)
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model
    fresh = build_model({"model": dict(CFG["model"], hidden_dim=64)}, (g.node_types, g.edge_types), None).to(dev)
    assert len(fresh.embeddings) == 0
    fresh.eval()
    with torch.no_grad():
        fresh(gd)                       # lazy creation on first forward (model.py:247-248)
    assert len(fresh.embeddings) == 4
    with pytest.raises(ValueError):
        build_model({"model": dict(CFG["model"], activation="gelu")}, (g.node_types, g.edge_types), None)
    with pytest.raises(ValueError):
        build_model({"model": dict(CFG["model"], architecture="GAT")}, (g.node_types, g.edge_types), None)
