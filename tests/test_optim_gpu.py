"""mmgnn.optim.Adam (one launch of mmg_adam_step over flat buckets) against torch.optim.Adam -- the reference's optimizer
(src/train.py:216-229) -- on the same gradients: parameters, moments, step count, skipped (None) gradients, weight decay,
state_dict round trip, and hipGraph replays advancing the device-resident step counter.  Plus mmg_vec_sums."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _params(dev, gen):
    shapes = [(128, 128), (128,), (64, 256), (1, 32), (1,), (300, 128), (7,)]
    return [torch.nn.Parameter(torch.randn(*s, generator=gen).to(dev)) for s in shapes]


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_matches_torch(dev, wd):
    import mmgnn  # noqa: F401
    from mmgnn.optim import Adam
    gen = torch.Generator().manual_seed(0)
    ps = _params(dev, gen)
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = Adam(ps, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    ropt = torch.optim.Adam(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    assert all(p.data_ptr() != 0 and p.is_contiguous() for p in ps)
    for it in range(6):
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i == 4 and it % 2 == 0:                 # a parameter without a gradient this step: skipped by both
                p.grad, r.grad = None, None
                continue
            g = torch.randn(*p.shape, generator=gen) * (10.0 ** (i - 3))
            p.grad, r.grad = g.to(dev), g.clone()
        opt.step()
        ropt.step()
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i == 4:
                continue                                # torch keeps a per-tensor step; ours is per bucket (see below)
            assert float((p.detach().cpu() - r.detach()).abs().max()) <= 2e-6 * float(r.detach().abs().max()) + 1e-7, (it, i)
    st = opt.state_dict()
    assert set(st) == {"state", "param_groups"} and set(st["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(st["state"][0]["step"]) == 6.0
    rst = ropt.state_dict()
    for i in (0, 1, 2, 3, 5, 6):
        assert float((st["state"][i]["exp_avg"].cpu() - rst["state"][i]["exp_avg"]).abs().max()) <= \
            1e-6 * float(rst["state"][i]["exp_avg"].abs().max()) + 1e-12
    # round trip: a fresh optimizer over fresh parameters continues from the saved state
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = Adam(ps2, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    opt2.load_state_dict(copy.deepcopy(st))
    for p, q in zip(ps, ps2):
        g = torch.randn(*p.shape, generator=gen).to(dev)
        p.grad, q.grad = g, g.clone()
    opt.step(); opt2.step()
    for p, q in zip(ps, ps2):
        assert torch.equal(p.detach(), q.detach())


def test_adam_step_counter_advances_inside_a_hipgraph(dev):
    import mmgnn  # noqa: F401
    from mmgnn.optim import Adam
    gen = torch.Generator().manual_seed(1)
    ps = _params(dev, gen)
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    grads = [torch.randn(*p.shape, generator=gen).to(dev) for p in ps]
    for p, g in zip(ps, grads):
        p.grad = g
    opt = Adam(ps, lr=1e-2)
    ropt = torch.optim.Adam(ref, lr=1e-2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph):
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(4):
        graph.replay()
    torch.cuda.synchronize()
    for _ in range(4):
        for r, g in zip(ref, grads):
            r.grad = g.cpu()
        ropt.step()
    assert float(opt.state[ps[0]]["step"]) == 4.0
    for p, r in zip(ps, ref):
        assert float((p.detach().cpu() - r.detach()).abs().max()) <= 2e-6 * float(r.detach().abs().max()) + 1e-7


def test_vec_sums(dev):
    import mmgnn  # noqa: F401
    from mmgnn import ops
    gen = torch.Generator().manual_seed(2)
    a = [torch.randn(128, 128, generator=gen).to(dev) for _ in range(3)]
    b = [torch.randn(128, generator=gen).to(dev) for _ in range(3)]
    c = [torch.randn(5, generator=gen).to(dev) for _ in range(4)]
    A, B, Cc, Dd = torch.empty(128, 128, device=dev), torch.empty(128, device=dev), torch.empty(5, device=dev), torch.empty(5, device=dev)
    ops.vec_sums([(A, a), (B, b), (Cc, c), (Dd, c[:1])])
    assert torch.equal(A, (a[0] + a[1]) + a[2]) and torch.equal(B, (b[0] + b[1]) + b[2])
    assert torch.equal(Cc, ((c[0] + c[1]) + c[2]) + c[3]) and torch.equal(Dd, c[0])


def test_vec_sums_takes_column_slices(dev):
    """2-D jobs with their own row strides: the halves W1[:, :D] | W1[:, D:] of an edge head's Linear(2D, H) are split into
    contiguous matrices and their gradients joined again in the launch that sums vectors; in-place accumulation."""
    import mmgnn  # noqa: F401
    from mmgnn import ops
    gen = torch.Generator().manual_seed(3)
    H, D = 64, 128
    w1 = torch.randn(H, 2 * D, generator=gen).to(dev)
    halves = torch.empty(2, H, D, device=dev)
    ops.vec_sums([(halves[0], [w1[:, :D]]), (halves[1], [w1[:, D:]])])
    assert torch.equal(halves[0], w1[:, :D]) and torch.equal(halves[1], w1[:, D:])
    ga, gb = torch.randn(H, D, generator=gen).to(dev), torch.randn(H, D, generator=gen).to(dev)
    g = torch.full((H, 2 * D), 7.0, device=dev)
    ops.vec_sums([(g[:, :D], [ga]), (g[:, D:], [gb, gb])])
    assert torch.equal(g, torch.cat([ga, gb + gb], dim=1))
    acc = ga.clone()
    ops.vec_sums([(acc, [acc, gb, w1[:, D:]])])                        # dst is also the first source
    assert torch.equal(acc, (ga + gb) + w1[:, D:])
    with pytest.raises(ValueError):
        ops.vec_sums([(g[:, :D], [ga.reshape(-1)])])                   # strided jobs are 2-D on every side
    with pytest.raises(ValueError):
        ops.vec_sums([(g.t(), [g.t()])])                               # unit column stride only


def test_counters_seed_and_zero_fill(dev):
    import mmgnn  # noqa: F401
    from mmgnn import ops
    cs = [torch.tensor(v, dtype=torch.int64, device=dev) for v in (0, 5, 2 ** 40)]
    ops.counters_add(cs, [1, 2, 3])
    ops.counters_add(cs[:1], [4])
    assert [int(c) for c in cs] == [5, 7, 2 ** 40 + 3]
    many = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(40)]        # more than one table
    ops.counters_add(many, list(range(40)))
    assert [int(c) for c in many] == list(range(40))
    # SplitMix64 stream: state[1] is the position, state[0] the seed the kernels read (< 2^62)
    st = torch.tensor([0, 12345], dtype=torch.int64, device=dev)
    seen = []
    for _ in range(3):
        ops.seed_advance(st)
        seen.append(int(st[0]))
    M = (1 << 64) - 1
    pos, ref = 12345, []
    for _ in range(3):
        pos = (pos + 0x9E3779B97F4A7C15) & M
        z = pos
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        ref.append((z ^ (z >> 31)) >> 2)
    assert seen == ref and len(set(seen)) == 3 and all(0 <= v < 2 ** 62 for v in seen)
    z = ops.zeros(3, 1000, device=dev)
    assert z.shape == (3, 1000) and float(z.abs().max()) == 0.0
    assert ops.zeros(0, 4, device=dev).numel() == 0


def test_graphed_step_draws_new_masks_on_the_device(dev):
    """A captured step advances its own dropout seed (mmg_seed_advance inside the graph): consecutive replays see
    different masks, and nothing runs on the stream between two replays."""
    import mmgnn  # noqa: F401
    from mmgnn.data import build_plan
    from mmgnn.model import build_model
    from mmgnn.synth import make_graph
    from mmgnn.train import PiecewiseGraphedTrainStep
    g = make_graph(1, seed=0, device=dev)
    cfg = {"model": {"architecture": "RGCN", "hidden_dim": 128, "num_layers": 2, "dropout": 0.2, "use_batch_norm": True,
                     "activation": "relu"}}
    torch.manual_seed(0)
    m = build_model(cfg, (g.node_types, g.edge_types), None).to(dev)
    m._init_embeddings(g)
    plan = build_plan(g, dev, use_cache=False)
    et = ("patient", "has_lab", "lab")
    ei = g[et].edge_index
    pi, li = ei[0].contiguous(), ei[1].contiguous()
    y = g[et].edge_attr.squeeze(-1).contiguous()
    sup = torch.arange(pi.numel(), device=dev) % 5 == 0
    opt = torch.optim.SGD([p for n, p in m.named_parameters() if not n.startswith("embeddings.")], lr=0.0)
    step = PiecewiseGraphedTrainStep(m, plan, pi, li, y, torch.ones(int(g["lab"].num_nodes), device=dev), opt, sup, None,
                                     n_sup_global=float(sup.sum()), warmup=1)
    seeds, losses = [], []
    for _ in range(3):
        seeds.append(int(m._seed_dev[0]))
        losses.append(float(step.step()))
    assert len(set(seeds)) == 3                        # lr = 0: the parameters stand still, only the masks change
    assert len(set(losses)) == 3 and step.loss.dtype == torch.float64
    nbt = int(m.patient_transform[1].num_batches_tracked)
    step.step()
    assert int(m.patient_transform[1].num_batches_tracked) > nbt      # BatchNorm counters advance inside the graph


def test_scheduler_survives_hipgraph_replay(dev, monkeypatch):
    """The hyper-parameters live on the device (mmg_adam_step_dev): a StepLR / ReduceLROnPlateau that changes
    param_groups['lr'] between replays (src/train.py:271-291 of the reference) changes the captured step."""
    import mmgnn  # noqa: F401
    from mmgnn.optim import Adam
    gen = torch.Generator().manual_seed(4)
    ps = _params(dev, gen)
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    grads = [torch.randn(*p.shape, generator=gen).to(dev) for p in ps]
    for p, g in zip(ps, grads):
        p.grad = g
    opt = Adam(ps, lr=1e-2, weight_decay=1e-3)
    ropt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.1)
    rsched = torch.optim.lr_scheduler.StepLR(ropt, step_size=2, gamma=0.1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph):
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    lrs = []
    for _ in range(5):
        opt.sync_hyper()
        graph.replay()
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
        for r, g in zip(ref, grads):
            r.grad = g.cpu()
        ropt.step()
        rsched.step()
    torch.cuda.synchronize()
    assert lrs[0] == pytest.approx(1e-2) and lrs[-1] == pytest.approx(1e-4)          # the schedule did change lr
    for p, r in zip(ps, ref):
        assert float((p.detach().cpu() - r.detach()).abs().max()) <= 2e-6 * float(r.detach().abs().max()) + 1e-7
    # a change made INSIDE a capture cannot reach the device: refused, not ignored
    opt.param_groups[0]["lr"] = 0.5
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    with pytest.raises(Exception, match="sync_hyper"):
        opt.step()


def test_add_param_group_gets_its_own_bucket(dev):
    import mmgnn  # noqa: F401
    from mmgnn.optim import Adam
    gen = torch.Generator().manual_seed(5)
    ps = _params(dev, gen)
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = Adam(ps[:4], lr=1e-2)
    ropt = torch.optim.Adam(ref[:4], lr=1e-2)
    opt.add_param_group({"params": ps[4:], "lr": 3e-2})
    ropt.add_param_group({"params": ref[4:], "lr": 3e-2})
    for _ in range(3):
        for p, r in zip(ps, ref):
            g = torch.randn(*p.shape, generator=gen)
            p.grad, r.grad = g.to(dev), g.clone()
        opt.step()
        ropt.step()
    for p, r in zip(ps, ref):
        assert float((p.detach().cpu() - r.detach()).abs().max()) <= 2e-6 * float(r.detach().abs().max()) + 1e-7


def test_supervision_mask_drawn_on_the_device(dev):
    """mmg_sup_mask_draw: the per-epoch 20 % subset (train.py:150-176) from the counter RNG -- fraction, subset size and
    normaliser, determinism per seed, a new subset per seed, id-keyed (partition-invariant) draws."""
    import mmgnn  # noqa: F401
    from mmgnn import ops
    n = 300_001
    st = torch.tensor([1234567, 0], dtype=torch.int64, device=dev)
    sup, cnt, inv = ops.sup_mask_draw(n, 0.2, dev, seed_dev=st)
    assert set(sup.unique().tolist()) == {0.0, 1.0}
    k = float(sup.sum(dtype=torch.float64))
    assert float(cnt) == k and float(inv) == 1.0 / k
    assert abs(k / n - 0.2) < 4 * (0.2 * 0.8 / n) ** 0.5 + 1e-4
    sup2, _, _ = ops.sup_mask_draw(n, 0.2, dev, seed=1234567)
    assert torch.equal(sup, sup2)                                   # the device seed and the host seed are the same key
    st[0] = 7654321
    sup3, cnt3, _ = ops.sup_mask_draw(n, 0.2, dev, seed_dev=st)
    assert not torch.equal(sup, sup3) and abs(float(cnt3) / n - 0.2) < 0.01
    agree = float((sup == sup3).float().mean())
    assert abs(agree - (0.04 + 0.64)) < 0.01                        # independent draws
    # keyed on the pair id: a shard that holds pairs [a, b) of the global list draws their part of the global mask
    ids = torch.arange(1000, 5000, dtype=torch.int64, device=dev)
    part, _, _ = ops.sup_mask_draw(ids.numel(), 0.2, dev, seed=1234567, ids=ids)
    assert torch.equal(part, sup[1000:5000])
    # ...and counts the GLOBAL subset by itself (count_only: no per-pair output), so no collective is needed for 1 / n_sup
    rest = torch.cat([torch.arange(0, 1000, dtype=torch.int64), torch.arange(5000, n, dtype=torch.int64)]).to(dev)
    other, c_other, _ = ops.sup_mask_draw(rest.numel(), 0.2, dev, seed=1234567, ids=rest)
    _, c_part, _ = ops.sup_mask_draw(ids.numel(), 0.2, dev, seed=1234567, ids=ids)
    none_, c_glob, i_glob = ops.sup_mask_draw(n, 0.2, dev, seed=1234567, count_only=True)
    assert none_ is None and float(c_glob) == float(c_part) + float(c_other) == k and float(i_glob) == 1.0 / k
    full, c1, i1 = ops.sup_mask_draw(1000, 1.0, dev, seed=3)
    none, c0, i0 = ops.sup_mask_draw(1000, 0.0, dev, seed=3)
    assert float(full.sum()) == 1000 and float(none.sum()) == 0 and float(c0) == 0 and float(i0) == 1.0
    empty, ce, _ = ops.sup_mask_draw(0, 0.2, dev, seed=3)
    assert empty.numel() == 0 and float(ce) == 0.0


@pytest.mark.parametrize("loss_type", ["mae", "mse", "huber"])
def test_pair_loss_types_match_torch(dev, loss_type):
    """mmg_pair_loss against compute_regression_loss's torch functions (model.py:579-612), value and gradient; huber
    crosses its delta = 1 knee."""
    import mmgnn  # noqa: F401
    from mmgnn import ops
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(6)
    n = 50_000
    pred = (torch.randn(n, generator=gen) * 2).to(dev)
    y = torch.randn(n, generator=gen).to(dev)
    pred[:3] = y[:3] + torch.tensor([1.0, -1.0, 0.0], device=dev)    # on the knee and at zero
    loss, dpred = ops.pair_loss(pred, y, None, None, 1.0 / n, loss_type)
    p64 = pred.double().cpu().requires_grad_(True)
    fn = {"mae": F.l1_loss, "mse": F.mse_loss, "huber": F.huber_loss}[loss_type]
    want = fn(p64, y.double().cpu())
    want.backward()
    assert abs(float(loss) - float(want)) <= 1e-6 * abs(float(want))
    assert float((dpred.double().cpu() - p64.grad).abs().max()) <= 1e-6 * float(p64.grad.abs().max())
    slot = torch.zeros(2, dtype=torch.float64, device=dev)
    l2, d2 = ops.pair_loss(pred, y, None, None, 1.0 / n, loss_type, loss_out=slot[1], want_dpred=False)
    assert d2 is None and float(slot[1]) == float(loss) and float(slot[0]) == 0.0
    with pytest.raises(ValueError, match="Unknown loss type"):
        ops.pair_loss(pred, y, None, None, 1.0, "logcosh")
