"""Caller side of the hot path on the GPU: the product's EdgeMasker / Trainer (counterparts of the reference's
src/train.py:37-176, 183-561) driven in the REFERENCE's call order (train.py:605-642), checkpoint layout (train.py:501-509)
and reload path (evaluate.py:620-632), frozen-embedding default and the explicit switch, and a multi-epoch
mask-and-recover run whose imputation metrics are compared with the CPU oracle trained on the same masks."""
import json
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_io import load, rel_err, t
from oracle import fixtures as fx
from oracle import model as om
from oracle import train as ot


def _config(hidden=64, dropout=0.0, epochs=3, opt="adam", lr=1e-3):
    return {"model": {"architecture": "RGCN", "hidden_dim": hidden, "num_layers": 2, "dropout": dropout,
                      "use_batch_norm": True, "activation": "relu"},
            "train": {"optimizer": {"type": opt, "lr": lr, "weight_decay": 1e-5, "momentum": 0.0},
                      "lr_scheduler": {"enabled": True, "type": "reduce_on_plateau", "factor": 0.5, "patience": 10},
                      "loss": "mae", "epochs": epochs, "early_stopping_patience": 20, "train_split": 0.7,
                      "val_split": 0.15, "test_split": 0.15, "mask_fraction": 0.2, "seed": 42, "device": "cuda"},
            "logging": {"save_checkpoints": True, "checkpoint_interval": 2}}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _reference_order(dev, cfg, n=(300, 12, 15, 10), sd=None, train_embeddings=False, mask_seed=123):
    """train.py:605-622 of the reference: the masker is built from the CPU graph, THEN Trainer moves model and graph."""
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model
    from mmgnn.train import EdgeMasker, Trainer
    g = fx.graph_from_frames(fx.det_frames(*n))                      # on the CPU, as torch.load(graph.pt) yields it
    tc = cfg["train"]
    masker = EdgeMasker(g, tc["train_split"], tc["val_split"], tc["test_split"], tc["mask_fraction"], tc["seed"],
                        mask_generator=torch.Generator().manual_seed(mask_seed))
    model = build_model(cfg, (g.node_types, g.edge_types), None)
    if sd is not None:
        model._init_embeddings(g)
        model.load_state_dict(sd)
    trainer = Trainer(model, g, masker, cfg, dev, train_embeddings=train_embeddings)
    return g, masker, model, trainer


def test_reference_main_order_trains_and_roundtrips_checkpoint(dev, tmp_path):
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model
    cfg = _config(epochs=3)
    g, masker, model, trainer = _reference_order(dev, cfg)
    assert masker.edge_index.device.type == "cuda" and masker.train_mask.device.type == "cuda"      # followed the graph
    assert trainer.lab_weights.device.type == "cuda"
    hist = trainer.train(tmp_path)
    assert len(hist["train_loss"]) == 3 and all(np.isfinite(hist["train_loss"])) and all(np.isfinite(hist["val_loss"]))
    assert json.load(open(tmp_path / "training_history.json"))["val_loss"] == hist["val_loss"]
    assert (tmp_path / "best_model.pt").exists() and (tmp_path / "checkpoint_epoch_2.pt").exists()
    ck = torch.load(tmp_path / "best_model.pt", map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "config"}      # train.py:503-509
    gold, meta = load("model_small.npz")
    assert list(ck["model_state_dict"].keys()) == meta["state_keys"]                                   # the reference's layout
    assert ck["val_loss"] == min(hist["val_loss"]) and ck["config"] == cfg
    n_opt = sum(len(gr["params"]) for gr in ck["optimizer_state_dict"]["param_groups"])
    assert n_opt == len([p for n, p in model.named_parameters() if not n.startswith("embeddings.")])   # F5: the lazily
    # created tables are not in the optimizer (train.py:219 runs before model.py:247) -> never updated, while their
    # .grad is computed
    ck2 = torch.load(tmp_path / "checkpoint_epoch_2.pt", map_location="cpu", weights_only=False)
    for k, v in model.state_dict().items():
        if k.startswith("embeddings."):
            assert torch.equal(v.cpu(), ck2["model_state_dict"][k]), k
            assert float(dict(model.named_parameters())[k].grad.abs().sum()) > 0
    trainer.load_best_model(tmp_path)
    test_loss = trainer.validate("test")
    # evaluate.py:620-632: fresh model, _init_embeddings BEFORE load_state_dict; both PyG key manglings load
    gd = trainer.data
    pi, li = masker.edge_index[0][masker.test_mask].contiguous(), masker.edge_index[1][masker.test_mask].contiguous()
    y = masker.edge_attr[masker.test_mask].squeeze(-1)
    for old_keys in (False, True):
        sd = ck["model_state_dict"]
        if old_keys:
            sd = {(re.sub(r"<(.+)>", lambda m: m.group(1).replace("___", "__"), k)): v for k, v in sd.items()}
            assert any("patient__has_lab__lab" in k for k in sd)
        fresh = build_model(cfg, (gd.node_types, gd.edge_types), None).to(dev)
        fresh._init_embeddings(gd)
        fresh.load_state_dict(sd)
        fresh.eval()
        with torch.no_grad():
            pred = fresh.predict_lab_values(gd, pi, li)
        assert abs(float((pred - y).abs().mean()) - test_loss) <= 1e-6 * max(1.0, test_loss)


def test_train_epoch_is_the_oracle_step_plus_sgd(dev):
    """One Trainer.train_epoch = the oracle's fwd + weighted MAE + bwd on the same supervision mask, followed by a plain SGD
    update of the non-embedding parameters (SGD: the update is linear in the gradient, so the comparison is exact)."""
    cfg = _config(hidden=64, dropout=0.0, epochs=1, opt="sgd", lr=0.05)
    n = (300, 12, 15, 10)
    g0 = fx.graph_from_frames(fx.det_frames(*n))
    gv = om.GraphView(g0)
    sd = fx.det_state(gv.num_nodes, 64)
    g, masker, model, trainer = _reference_order(dev, cfg, n, sd=sd, mask_seed=77)
    gold, meta = load("model_small.npz")
    assert rel_err(trainer.lab_weights.cpu(), t(gold["lab_weights"])) <= 1e-5            # train.py:295-330
    loss = trainer.train_epoch()
    # the same mask the trainer drew: its generator restarted from the same seed
    ei, ea = g0["patient", "has_lab", "lab"].edge_index, g0["patient", "has_lab", "lab"].edge_attr
    tr, _, _ = ot.edge_splits(ei.shape[1])
    assert torch.equal(masker.train_mask.cpu(), tr)
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    gen = torch.Generator().manual_seed(77)
    torch.rand(int(tr.sum()), generator=gen)                 # Trainer.__init__ -> _compute_lab_weights draws one mask first
    sup = torch.rand(int(tr.sum()), generator=gen) < 0.2
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    oloss, opred, ograds, obufs = ot.train_step_grads(sd, gv, pi, li, y, w, sup, p=0.0)
    assert abs(loss - float(oloss)) <= 1e-4 * abs(float(oloss))
    lr, wd = 0.05, 1e-5
    for k, p in model.named_parameters():
        # (the tables exist before the optimizer is built here -- a loaded state -- so they are updated as well)
        # a parameter no output depends on (the last layer's convs into the vocab types) has .grad None: SGD skips it
        want = sd[k] if p.grad is None else sd[k] - lr * (ograds[k] + wd * sd[k])
        assert p.grad is not None or float(ograds[k].abs().max()) == 0.0, k
        tol = lr * (2e-4 * float(ograds[k].abs().max()) + 1e-7) + 2.4e-7 * float(sd[k].abs().max())   # + 2 ulp of the value
        assert float((p.detach().cpu() - want).abs().max()) <= tol, k


def test_train_embeddings_switch(dev):
    cfg = _config(epochs=2, lr=1e-2)
    g, masker, model, trainer = _reference_order(dev, cfg, train_embeddings=True)
    before = {k: v.detach().clone() for k, v in model.embeddings.state_dict().items()}
    trainer.train_epoch(); trainer.train_epoch()
    after = model.embeddings.state_dict()
    assert all(float((after[k] - before[k]).abs().max()) > 0 for k in before)   # the switch makes them trainable


def test_mask_and_recover_metrics_follow_the_oracle(dev):
    """BASELINE config 5's parity criterion at test size: several epochs of mask-and-recover training (a new 20 % mask
    every epoch, injected so that both sides see the same masks), then R^2 / MAE of the imputed TEST labs.  The oracle is
    trained by the same plain-SGD rule on the CPU; dropout 0 (the device RNG cannot be replayed on the host)."""
    import mmgnn  # noqa: F401
    from mmgnn.evaluate import compute_regression_metrics
    cfg = _config(hidden=64, dropout=0.0, epochs=6, opt="sgd", lr=0.2)
    n = (900, 50, 200, 100)                                   # the MIMIC-III schema's vocabulary (config.yaml:70,101,112)
    g0 = fx.graph_from_frames(fx.det_frames(*n))
    gv = om.GraphView(g0)
    sd = fx.det_state(gv.num_nodes, 64)
    g, masker, model, trainer = _reference_order(dev, cfg, n, sd=sd, mask_seed=5)
    for _ in range(6):
        trainer.train_epoch()
    ei, ea = g0["patient", "has_lab", "lab"].edge_index, g0["patient", "has_lab", "lab"].edge_attr
    tr, va, te = ot.edge_splits(ei.shape[1])
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    w = ot.lab_weights(li, y, gv.num_nodes["lab"])
    gen = torch.Generator().manual_seed(5)
    torch.rand(int(tr.sum()), generator=gen)
    cur = {k: v.clone() for k, v in sd.items()}
    for _ in range(6):
        sup = torch.rand(int(tr.sum()), generator=gen) < 0.2
        _, _, grads, bufs = ot.train_step_grads(cur, gv, pi, li, y, w, sup, p=0.0)
        for k in grads:                                       # (loaded state: the tables are in the optimizer too)
            if float(grads[k].abs().max()) > 0.0:             # .grad None in autograd: the optimizer skips the tensor
                cur[k] = cur[k] - 0.2 * (grads[k] + 1e-5 * cur[k])
        cur.update(bufs)
    tp, tl, ty = ei[0][te], ei[1][te], ea[te].squeeze(-1)
    model.eval()
    with torch.no_grad():
        pred = model.predict_lab_values(trainer.data, tp.to(dev), tl.to(dev)).cpu()
    opred, _ = om.predict_lab_values(cur, gv, tp, tl)
    got = compute_regression_metrics(pred.numpy(), ty.numpy())
    want = ot.regression_metrics(opred.numpy(), ty.numpy())
    assert rel_err(pred, opred) <= 2e-3                      # six chained steps: rounding differences compound
    assert abs(got["mae"] - want["mae"]) <= 1e-3 * want["mae"]
    assert abs(got["r2"] - want["r2"]) <= 2e-3
    assert abs(got["rmse"] - want["rmse"]) <= 1e-3 * want["rmse"]


def _history_pair(dev, tmp_path, cfg, n=(300, 12, 15, 10), mask_seed=11):
    """The same training run twice from one state: eager autograd steps (device_step=False) and the captured device step."""
    g0 = fx.graph_from_frames(fx.det_frames(*n))
    gv = om.GraphView(g0)
    sd = fx.det_state(gv.num_nodes, cfg["model"]["hidden_dim"])
    out = []
    for device_step in (False, True):
        import mmgnn  # noqa: F401
        from mmgnn.model import build_model
        from mmgnn.train import EdgeMasker, Trainer
        g = fx.graph_from_frames(fx.det_frames(*n))
        tc = cfg["train"]
        masker = EdgeMasker(g, tc["train_split"], tc["val_split"], tc["test_split"], tc["mask_fraction"], tc["seed"],
                            mask_generator=torch.Generator().manual_seed(mask_seed))
        model = build_model(cfg, (g.node_types, g.edge_types), None)
        model._init_embeddings(g)
        model.load_state_dict(sd)
        trainer = Trainer(model, g, masker, cfg, dev, device_step=device_step)
        d = tmp_path / ("graph" if device_step else "eager")
        hist = trainer.train(d)
        out.append((hist, trainer, model))
    return out


@pytest.mark.parametrize("sched", ["step", "reduce_on_plateau"])
@pytest.mark.parametrize("loss_fn", ["mae", "huber"])
def test_trainer_train_runs_the_captured_step_and_reproduces_the_eager_history(dev, tmp_path, sched, loss_fn):
    """Trainer.train() -- the reference's entry point (train.py:433-544) -- replays the captured device step and a captured
    validation pass with one host read per epoch; its history (losses, learning rates under a scheduler that really
    changes lr, early-stopping bookkeeping, checkpoints) is the eager autograd path's on the same injected masks."""
    cfg = _config(hidden=64, dropout=0.0, epochs=5, opt="adam", lr=5e-3)
    cfg["train"]["loss"] = loss_fn
    if sched == "step":
        cfg["train"]["lr_scheduler"] = {"enabled": True, "type": "step", "step_size": 2, "gamma": 0.5}
    else:
        cfg["train"]["lr_scheduler"] = {"enabled": True, "type": "reduce_on_plateau", "factor": 0.5, "patience": 0}
    (he, te, me), (hg, tg, mg) = _history_pair(dev, tmp_path, cfg)
    assert tg._dstep is not None and te._dstep is None               # the graph path ran / the eager path did not build one
    assert "val" in tg._deval
    assert len(hg["train_loss"]) == len(he["train_loss"]) == 5
    assert hg["learning_rates"] == he["learning_rates"]
    if sched == "step":
        assert hg["learning_rates"] == pytest.approx([5e-3, 5e-3, 2.5e-3, 2.5e-3, 1.25e-3])
    for k in ("train_loss", "val_loss"):
        for a, b in zip(hg[k], he[k]):
            assert abs(a - b) <= 2e-4 * abs(b), (k, hg[k], he[k])
    assert tg.best_val_loss == pytest.approx(te.best_val_loss, rel=2e-4) and tg.patience_counter == te.patience_counter
    for (k, a), (_, b) in zip(mg.state_dict().items(), me.state_dict().items()):
        if a.is_floating_point():
            assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-6, k      # five chained Adam steps
        else:
            assert torch.equal(a, b), k                                                     # BatchNorm step counters
    assert (tmp_path / "graph" / "best_model.pt").exists() and (tmp_path / "graph" / "checkpoint_epoch_2.pt").exists()
    ck = torch.load(tmp_path / "graph" / "best_model.pt", map_location="cpu", weights_only=False)
    assert ck["val_loss"] == min(hg["val_loss"])
    # the public single-epoch calls are the same captured steps (one host read each)
    assert isinstance(tg.train_epoch(), float) and isinstance(tg.validate("val"), float)
    assert isinstance(tg.validate("test"), float) and "test" in tg._deval


def test_trainer_without_a_mask_generator_draws_its_masks_inside_the_step(dev, tmp_path):
    """No mask_generator = the reference's wall-clock seeded redraw every epoch (train.py:156): the captured step draws the
    subset itself (mmg_sup_mask_draw), a different one every epoch, and normalises by its size."""
    import mmgnn  # noqa: F401
    from mmgnn.model import build_model
    from mmgnn.train import EdgeMasker, Trainer
    cfg = _config(hidden=64, dropout=0.2, epochs=4, opt="adam", lr=1e-3)
    g = fx.graph_from_frames(fx.det_frames(300, 12, 15, 10))
    masker = EdgeMasker(g, 0.7, 0.15, 0.15, 0.2, 42)
    model = build_model(cfg, (g.node_types, g.edge_types), None)
    trainer = Trainer(model, g, masker, cfg, dev)
    masks, losses = [], []
    for _ in range(3):
        losses.append(trainer.train_epoch())
        st = trainer._dstep
        masks.append(st.sup.clone())
        n_sup = float(st.sup.sum())
        assert float(st._sv.count) == n_sup and float(st._sv.inv_den) == pytest.approx(1.0 / n_sup)
        assert 0.1 < n_sup / st.sup.numel() < 0.3
    assert st.mask_fraction == pytest.approx(0.2) and all(np.isfinite(losses))
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    with pytest.raises(ValueError, match="draws its supervision subset"):
        st.set_mask(masks[0])
    hist = trainer.train(tmp_path)
    assert len(hist["train_loss"]) == 4 and all(np.isfinite(hist["train_loss"] + hist["val_loss"]))


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_heads_on_the_supervised_pairs_only_change_nothing_a_step_returns(dev, p):
    """The captured training step evaluates the edge heads on the supervised pairs alone (the loss of train.py:366-386
    reads predictions[supervision_mask] and nothing else).  Against the full sweep over all train pairs: the predictions of
    the supervised pairs, the loss, and every parameter / buffer after two Adam steps are BITWISE equal (per-pair
    arithmetic, counter RNG keyed on the pair id); unsupervised pairs read 0 instead of a prediction nobody uses."""
    import mmgnn  # noqa: F401
    from mmgnn.data import build_plan
    from mmgnn.model import build_model
    from mmgnn.optim import Adam
    from mmgnn.train import PiecewiseGraphedTrainStep
    n = (600, 20, 25, 18)
    cfg = _config(hidden=128, dropout=p)
    g0 = fx.graph_from_frames(fx.det_frames(*n))
    gv = om.GraphView(g0)
    sd = fx.det_state(gv.num_nodes, 128)
    ei, ea = g0["patient", "has_lab", "lab"].edge_index, g0["patient", "has_lab", "lab"].edge_attr
    tr, _, _ = ot.edge_splits(ei.shape[1])
    pi, li, y = ei[0][tr].to(dev), ei[1][tr].to(dev), ea[tr].squeeze(-1).to(dev)
    w = ot.lab_weights(ei[1][tr], ea[tr].squeeze(-1), gv.num_nodes["lab"]).to(dev)
    sup = (torch.rand(int(tr.sum()), generator=torch.Generator().manual_seed(3)) < 0.2).to(dev)
    res = []
    for flag in (False, True):
        torch.manual_seed(99)                                     # the same dropout seed stream for both
        g = fx.graph_from_frames(fx.det_frames(*n)).to(dev)
        model = build_model(cfg, (g.node_types, g.edge_types), None).to(dev)
        model._init_embeddings(g)
        model.load_state_dict(sd)
        opt = Adam([q for k, q in model.named_parameters() if not k.startswith("embeddings.")], lr=1e-2)
        step = PiecewiseGraphedTrainStep(model, build_plan(g, dev, use_cache=False), pi, li, y, w, opt, sup, None,
                                         supervised_heads_only=flag)
        losses = [float(step.step()) for _ in range(2)]
        res.append((losses, step.pred.clone(), {k: v.clone() for k, v in model.state_dict().items()}))
    (l0, p0, s0), (l1, p1, s1) = res
    assert l0 == l1
    assert torch.equal(p0[sup], p1[sup]) and float(p1[~sup].abs().max()) == 0.0 and float(p0[~sup].abs().max()) > 0.0
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_pair_backward_from_saved_forward_state_in_the_training_step(dev, monkeypatch):
    """mmgnn.model.SAVE_PAIR_STATE: the supervised-only heads' forward leaves the first layer's sign bits and the second
    layer's activations, the backward reads them instead of recomputing masks and the 64 x 32 product.  The recomputing
    backward takes that product in the forward's own order, so the two steps are the SAME step: losses, predictions and
    every parameter / buffer after two Adam steps bit for bit."""
    import mmgnn  # noqa: F401
    import mmgnn.model as mm
    from mmgnn.data import build_plan
    from mmgnn.model import build_model
    from mmgnn.optim import Adam
    from mmgnn.train import PiecewiseGraphedTrainStep
    n = (600, 20, 25, 18)
    cfg = _config(hidden=128, dropout=0.2)
    g0 = fx.graph_from_frames(fx.det_frames(*n))
    gv = om.GraphView(g0)
    sd = fx.det_state(gv.num_nodes, 128)
    ei, ea = g0["patient", "has_lab", "lab"].edge_index, g0["patient", "has_lab", "lab"].edge_attr
    tr, _, _ = ot.edge_splits(ei.shape[1])
    pi, li, y = ei[0][tr].to(dev), ei[1][tr].to(dev), ea[tr].squeeze(-1).to(dev)
    w = ot.lab_weights(ei[1][tr], ea[tr].squeeze(-1), gv.num_nodes["lab"]).to(dev)
    sup = (torch.rand(int(tr.sum()), generator=torch.Generator().manual_seed(3)) < 0.2).to(dev)
    res = []
    for flag in (False, True):
        monkeypatch.setattr(mm, "SAVE_PAIR_STATE", flag)
        torch.manual_seed(99)
        g = fx.graph_from_frames(fx.det_frames(*n)).to(dev)
        model = build_model(cfg, (g.node_types, g.edge_types), None).to(dev)
        model._init_embeddings(g)
        model.load_state_dict(sd)
        opt = Adam([q for k, q in model.named_parameters() if not k.startswith("embeddings.")], lr=1e-2)
        step = PiecewiseGraphedTrainStep(model, build_plan(g, dev, use_cache=False), pi, li, y, w, opt, sup, None,
                                         supervised_heads_only=True)
        losses = [float(step.step()) for _ in range(2)]
        res.append((losses, step.pred.clone(), {k: v.clone() for k, v in model.state_dict().items()}))
    (l0, p0, s0), (l1, p1, s1) = res
    assert l0 == l1 and torch.equal(p0, p1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
