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
