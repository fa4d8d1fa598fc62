"""Per-kernel parity: every C-ABI entry point vs a CPU restatement on the same seeded inputs.

Bars: integer/index work bit-exact; fp32 within 1e-5 relative of an fp64 CPU computation unless a
test states otherwise (the step-level bar of BASELINE.json is 1e-4).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import graph as og
from oracle.pyg_min import scatter_mean


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    import mmgnn  # noqa: F401
    from mmgnn import ops as o
    return o


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rand_edges(gen, n_rows, n_cols, E):
    r = torch.randint(0, n_rows, (E,), generator=gen)
    c = torch.randint(0, n_cols, (E,), generator=gen)
    return torch.stack([r, c]).contiguous()


# ------------------------------------------------------------------------------------------ CSR
@pytest.mark.parametrize("n_rows,n_cols,E,sort_row", [
    (1, 1, 0, 0), (7, 5, 1, 0), (300, 12, 5000, 0), (300, 12, 5000, 1), (1834, 50, 61484, 0),
    (70000, 50, 300000, 0), (50, 70000, 300000, 1), (3, 3, 4097, 0),
])
def test_csr_build_bit_exact(ops, dev, n_rows, n_cols, E, sort_row):
    gen = torch.Generator().manual_seed(E + n_rows)
    ei = rand_edges(gen, n_rows, n_cols, E)
    if sort_row == 1:
        ei = ei.flip(0).contiguous()
        n_sort = n_rows
    else:
        n_sort = n_rows
    rp, col, perm = ops.csr_build(ei.to(dev), n_sort, sort_row)
    rrp, rcol, rperm = og.csr_reference(ei, n_sort, sort_row)
    assert torch.equal(rp.cpu(), rrp)
    assert torch.equal(perm.cpu(), rperm)
    assert torch.equal(col.cpu(), rcol)
    deg, inv = ops.row_degree(rp)
    d = (rrp[1:] - rrp[:-1])
    assert torch.equal(deg.cpu(), d)
    assert torch.equal(inv.cpu(), 1.0 / d.clamp(min=1).float())
    if E:
        nc = int(rcol.max()) + 1
        cnt, cinv = ops.col_degree(col, nc)
        assert torch.equal(cnt.cpu().long(), torch.bincount(rcol.long(), minlength=nc))


def test_csr_build_large_keys_three_passes(ops, dev):
    # > 2^16 rows forces three 8-bit radix passes (odd pass count lands in perm directly)
    gen = torch.Generator().manual_seed(1)
    ei = rand_edges(gen, 200000, 50, 1 << 20)
    rp, col, perm = ops.csr_build(ei.to(dev), 200000, 0)
    rrp, rcol, rperm = og.csr_reference(ei, 200000, 0)
    assert torch.equal(rp.cpu(), rrp) and torch.equal(perm.cpu(), rperm) and torch.equal(col.cpu(), rcol)


# ------------------------------------------------------------------------------------- aggregates
def _csr(ops, dev, ei, n_rows):
    rp, col, perm = ops.csr_build(ei.to(dev), n_rows, 0)
    return rp, col


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("n_rows", [1, 257, 1834])
def test_gather_rows_is_scatter_mean(ops, dev, D, n_rows):
    gen = torch.Generator().manual_seed(D + n_rows)
    sizes = [50, 114, 100]
    rels, ref = [], torch.zeros(n_rows, D, dtype=torch.float64)
    for k, nc in enumerate(sizes):
        E = n_rows * (3 + 10 * k)
        ei = rand_edges(gen, n_rows, nc, E)
        ei[0, : E // 10] = 0 if n_rows == 1 else ei[0, : E // 10] % max(n_rows // 2, 1)   # leave empty rows
        tab = torch.randn(nc, D, generator=gen)
        rp, col = _csr(ops, dev, ei, n_rows)
        _, inv = ops.row_degree(rp)
        rels.append(ops.Rel(rp, col, nc, rowscale=inv, table=tab.to(dev)))
        ref += scatter_mean(tab.double(), ei.flip(0), n_rows)   # src = vocab (row 1), dst = patient (row 0)
    out = torch.full((n_rows, D), 7.0, device=dev)
    ops.gather_rows(rels, n_rows, D, out, accumulate=False)
    assert rel(out, ref) <= 1e-5
    base = torch.randn(n_rows, D, generator=gen)
    out2 = base.to(dev).clone()
    ops.gather_rows(rels, n_rows, D, out2, accumulate=True)
    assert rel(out2, ref + base.double()) <= 1e-5


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("n_rows", [1, 300, 5000])
def test_scatter_rows_is_scatter_mean(ops, dev, D, n_rows):
    gen = torch.Generator().manual_seed(3 * D + n_rows)
    sizes = [50, 114, 100]
    x = torch.randn(n_rows, D, generator=gen)
    rels, refs = [], []
    for k, nc in enumerate(sizes):
        E = n_rows * (5 + 12 * k)
        ei = rand_edges(gen, n_rows, nc, E)
        ei[1] = ei[1] % max(nc - 3, 1)          # some vocab rows get no edge -> 0
        rp, col = _csr(ops, dev, ei, n_rows)
        cnt, cinv = ops.col_degree(col, nc)
        out = torch.full((nc, D), -3.0, device=dev)
        rels.append(ops.Rel(rp, col, nc, colscale=cinv, out=out))
        refs.append(scatter_mean(x.double(), ei, nc))
    ops.scatter_rows(rels, n_rows, D, x.to(dev))
    for r, ref in zip(rels, refs):
        assert rel(r.out, ref) <= 1e-5


def test_scatter_rows_rowscale_and_global_atomic_fallback(ops, dev):
    # n_cols too large for the LDS accumulators -> global-atomic path
    gen = torch.Generator().manual_seed(5)
    n_rows, nc, D = 2000, 3000, 128
    ei = rand_edges(gen, n_rows, nc, 20000)
    x = torch.randn(n_rows, D, generator=gen)
    rs = torch.rand(n_rows, generator=gen) + 0.5
    rp, col = _csr(ops, dev, ei, n_rows)
    out = torch.empty(nc, D, device=dev)
    ops.scatter_rows([ops.Rel(rp, col, nc, rowscale=rs.to(dev), out=out)], n_rows, D, x.to(dev))
    ref = torch.zeros(nc, D, dtype=torch.float64).index_add_(0, ei[1], (x * rs[:, None]).double()[ei[0]])
    assert rel(out, ref) <= 1e-5


def simple_edges(gen, n_rows, n_cols, max_deg):
    """Edges without duplicate (row, col) pairs and with ragged degrees (some rows empty)."""
    order = torch.rand(n_rows, n_cols, generator=gen).argsort(1)
    deg = torch.randint(0, min(max_deg, n_cols) + 1, (n_rows,), generator=gen)
    deg[::7] = 0
    keep = torch.arange(n_cols)[None, :] < deg[:, None]
    r, k = torch.nonzero(keep, as_tuple=True)
    ei = torch.stack([r, order[r, k]])
    return ei[:, torch.randperm(ei.shape[1], generator=gen)].contiguous()


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("n_rows", [257, 1834, 5000])
def test_aggregates_simple_graph_paths(ops, dev, D, n_rows):
    """MMG_REL_SIMPLE relations (no repeated (row, col) pair).  With MMG_AGG_BF16=1 the gather takes the
    0/1-indicator x 3-way bf16 split matrix-core kernel: products are exact, so either way the result must be
    fp32-accurate (1e-6 of an fp64 reference)."""
    gen = torch.Generator().manual_seed(7 * D + n_rows)
    sizes, degs = [50, 114, 100], [50, 9, 25]
    x = torch.randn(n_rows, D, generator=gen) * 3 + 0.5
    grels, srels, gref, srefs = [], [], torch.zeros(n_rows, D, dtype=torch.float64), []
    for nc, md in zip(sizes, degs):
        ei = simple_edges(gen, n_rows, nc, md)
        tab = torch.randn(nc, D, generator=gen) * 2 - 0.3
        rp, col = _csr(ops, dev, ei, n_rows)
        _, inv = ops.row_degree(rp)
        cnt, cinv = ops.col_degree(col, nc)
        grels.append(ops.Rel(rp, col, nc, rowscale=inv, colscale=cinv, table=tab.to(dev), simple=True))
        gref += scatter_mean((tab.double() * cinv.cpu().double()[:, None]), ei.flip(0), n_rows)
        if len(grels) == 1:
            gref0 = gref.clone()
        out = torch.full((nc, D), -3.0, device=dev)
        mask, mask_r = ops.rel_mask_build(rp, col, nc)
        padc = (nc + 31) // 32 * 32
        nf = padc // 16
        want_r = torch.zeros(n_rows * 2 * nf, dtype=torch.int64)                 # row-major fields, bit-exact
        want_r.index_add_(0, (ei[0] * 2 + ((ei[1] % 16) // 8)) * nf + ei[1] // 16,
                          torch.ones(ei.shape[1], dtype=torch.int64) << (4 + ei[1] % 8))
        got_r = mask_r.cpu().view(torch.int16).to(torch.int64)[:n_rows * 2 * nf] & 0xFFFF
        assert torch.equal(got_r, want_r)
        grels[-1].mask_r = mask_r
        want = torch.zeros((n_rows + 63) // 64, padc, 2, dtype=torch.int64)      # bit planes, bit-exact
        p64 = ei[0] % 64
        want.view(-1).index_add_(0, ((ei[0] // 64) * padc + ei[1]) * 2 + ((p64 // 8) & 1),
                                 torch.ones(ei.shape[1], dtype=torch.int64) << (16 * (p64 // 16) + 4 + p64 % 8))
        assert torch.equal(mask.cpu().view(-1, padc, 2), want)
        srels.append(ops.Rel(rp, col, nc, colscale=cinv, out=out, simple=True, mask_t=mask))
        srefs.append(scatter_mean(x.double(), ei, nc))
    base = torch.randn(n_rows, D, generator=gen)
    o = base.to(dev).clone()
    ops.gather_rows(grels, n_rows, D, o, accumulate=True)
    assert rel(o, gref + base.double()) <= 1e-6
    o2 = torch.empty(n_rows, D, device=dev)
    _, gs = ops.gather_rows(grels, n_rows, D, o2, accumulate=False, with_stats=True)
    assert rel(o2, gref) <= 1e-6
    assert rel(gs[0], gref.sum(0)) <= 2e-6 and rel(gs[1], (gref * gref).sum(0)) <= 2e-6      # epilogue statistics
    o3 = torch.empty(n_rows, D, device=dev)                 # first relation alone (last layer's backward)
    ops.gather_rows(grels[:1], n_rows, D, o3, accumulate=False)
    assert rel(o3, gref0) <= 1e-6
    ops.scatter_rows(srels, n_rows, D, x.to(dev))
    for r, ref in zip(srels, srefs):
        assert rel(r.out, ref) <= 1e-6
    # backward use of scatter: per-relation rowscale
    for r in srels:
        r.colscale = None
    rs = [torch.rand(n_rows, generator=gen) + 0.25 for _ in srels]
    for r, w in zip(srels, rs):
        r.rowscale = w.to(dev)
    ops.scatter_rows(srels, n_rows, D, x.to(dev))
    for r, w in zip(srels, rs):
        ei_r = torch.stack([torch.repeat_interleave(torch.arange(n_rows), (r.rowptr[1:] - r.rowptr[:-1]).cpu().long()),
                            r.col.cpu().long()])
        ref = torch.zeros(r.n_cols, D, dtype=torch.float64).index_add_(0, ei_r[1], (x.double() * w.double()[:, None])[ei_r[0]])
        assert rel(r.out, ref) <= 1e-6


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("n_rows", [64, 640, 1000, 5000])
@pytest.mark.parametrize("sizes", [[50, 200, 100], [30, 40, 25], [60, 100, 90], [64, 150, 100], [20], [300, 90]])
def test_scatter_rows_strip_instances(ops, dev, D, n_rows, sizes):
    """Every instance of the two-wave strip kernel (forward scatter, no rowscale) against scatter_mean in fp64: 11 packed item
    tiles <6, 5> (the 50 / 200 / 100 vocabulary of config 5), 3 -> <2, 2>, 8 -> <4, 4>, 10 -> <5, 5>, one tile, and 13 tiles
    (beyond the strip instances: the unit-per-wave kernel)."""
    gen = torch.Generator().manual_seed(11 * D + n_rows + sum(sizes))
    degs = [min(50, sizes[0])] + [9, 25][:len(sizes) - 1]
    x = torch.randn(n_rows, D, generator=gen) * 2 - 0.4
    rels, refs = [], []
    for nc, md in zip(sizes, degs):
        ei = simple_edges(gen, n_rows, nc, md)
        rp, col = _csr(ops, dev, ei, n_rows)
        cnt, cinv = ops.col_degree(col, nc)
        mask, _ = ops.rel_mask_build(rp, col, nc)
        rels.append(ops.Rel(rp, col, nc, colscale=cinv, out=torch.full((nc, D), 5.0, device=dev), simple=True, mask_t=mask))
        refs.append(scatter_mean(x.double(), ei, nc))
    ops.scatter_rows(rels, n_rows, D, x.to(dev))
    first = [r.out.clone() for r in rels]
    for r, ref in zip(rels, refs):
        assert rel(r.out, ref) <= 1e-6
    for _ in range(12):                                   # fixed-order sums: bit-reproducible, launch after launch
        ops.scatter_rows(rels, n_rows, D, x.to(dev))      # (a <2, 2> instance of this kernel was dropped for failing exactly this)
        for r, f in zip(rels, first):
            assert torch.equal(r.out, f)


def _simple_scatter_problem(ops, dev, gen, n_rows, sizes, degs, D):
    rels, eis = [], []
    for nc, md in zip(sizes, degs):
        ei = simple_edges(gen, n_rows, nc, md)
        rp, col = _csr(ops, dev, ei, n_rows)
        mask, _ = ops.rel_mask_build(rp, col, nc)
        rels.append(ops.Rel(rp, col, nc, out=torch.full((nc, D), 5.0, device=dev), simple=True, mask_t=mask))
        # edges in CSR order (what the kernel's sums are compared with)
        eis.append(torch.stack([torch.repeat_interleave(torch.arange(n_rows), (rp[1:] - rp[:-1]).cpu().long()),
                                col.cpu().long()]))
    return rels, eis


@pytest.mark.parametrize("rowscale", [False, True])
@pytest.mark.parametrize("sizes", [[50, 114, 100], [50, 200, 100]])
def test_scatter_f16_pieces_follow_the_data_range(ops, dev, sizes, rowscale):
    """The matrix-core scatter multiplies TWO f16 pieces of x * 2^e, e chosen per wave from the rows it streams and lowered
    (accumulators rescaled, block split again) when a later block does not fit.  Whatever the magnitudes -- 1e-30 .. 1e+30,
    rising, falling, one huge row in the middle, zero blocks at the start -- a sum must be as good as an fp32 running sum
    of a handful of terms: the error is measured against the sum of |terms| of the SAME output element, so a small column
    cannot hide behind a large one, and every output is finite.  Bar 6e-7: two f16 pieces carry 22 bits (2^-22 = 2.4e-7 per
    term at worst), the matrix cores' fp32 accumulation adds a few 6e-8 (the exact three-bf16-piece kernels of the row-scaled
    launches measure 3.7e-7 here), and in the steep `falling` / `rising` cases ONE term dominates a sum, so the per-term
    bound is what is measured."""
    D, n_rows = 128, 6000
    gen = torch.Generator().manual_seed(len(sizes) * 97 + sizes[1] + int(rowscale))
    rels, eis = _simple_scatter_problem(ops, dev, gen, n_rows, sizes, [50, 9, 25], D)
    rs = [torch.rand(n_rows, generator=gen) + 0.25 for _ in rels]
    if rowscale:
        for r, w in zip(rels, rs):
            r.rowscale = w.to(dev)
    base = torch.randn(n_rows, D, generator=gen)
    ramp = torch.linspace(-30, 30, n_rows)[:, None]
    cases = {
        "unit": base,
        "tiny": base * 1e-30,
        "huge": base * 1e30,
        "rising": base * 10.0 ** ramp,                     # every row range starts small and ends large: e keeps falling
        "falling": base * 10.0 ** (-ramp),
        "spike": base.clone().index_put_((torch.tensor([3001]),), base[3001] * 1e25),
        "zero_head": torch.cat([torch.zeros(1500, D), base[1500:] * 1e-12]),      # all-zero blocks before the first data
        "columns": base * 10.0 ** torch.linspace(-6, 6, D)[None, :],              # columns of one strip 1e3 apart, strips 1e12
    }
    for name, x in cases.items():
        x = x.float()
        for r in rels:
            r.out.fill_(5.0)
        ops.scatter_rows(rels, n_rows, D, x.to(dev))
        for r, ei, w in zip(rels, eis, rs):
            xs = x.double() * (w.double()[:, None] if rowscale else 1.0)
            xs = xs.float().double() if rowscale else xs                         # (the reference's own fp32 product)
            ref = torch.zeros(r.n_cols, D, dtype=torch.float64).index_add_(0, ei[1], xs[ei[0]])
            mag = torch.zeros(r.n_cols, D, dtype=torch.float64).index_add_(0, ei[1], xs[ei[0]].abs())
            got = r.out.double().cpu()
            assert torch.isfinite(got).all(), name
            err = ((got - ref).abs() / mag.clamp(min=1e-300)).max().item() if mag.max() > 0 else 0.0
            assert err <= 6e-7, (name, r.n_cols, err)
            assert torch.equal(got[mag == 0], torch.zeros_like(got[mag == 0])), name


@pytest.mark.parametrize("rowscale", [False, True])
def test_scatter_on_heavy_tailed_rows_is_reproducible_and_accurate(ops, dev, rowscale):
    """Gradient-like inputs -- log-normal magnitudes, rows whose scale varies by e^4, isolated spikes of 1e6 -- drive the f16
    scatter through every rare path (scale lowered with the accumulators rescaled, outlier blocks multiplied exactly) many
    times per launch: repeats must agree bit for bit and every sum must be fp32-class against the sum of |terms| of its own
    element.  (A row-scaled strip kernel tried in round 4 failed exactly this; the row-scaled launches keep the exact
    three-bf16-piece kernel.)"""
    D, n_rows = 128, 20000
    gen = torch.Generator().manual_seed(31 + int(rowscale))
    rels, eis = _simple_scatter_problem(ops, dev, gen, n_rows, [50, 114, 100], [50, 9, 25], D)
    rs = [torch.rand(n_rows, generator=gen) + 0.25 for _ in rels]
    if rowscale:
        for r, w in zip(rels, rs):
            r.rowscale = w.to(dev)
    base = torch.randn(n_rows, D, generator=gen)
    spikes = base.clone()
    spikes[torch.randint(0, n_rows, (60,), generator=gen), torch.randint(0, D, (60,), generator=gen)] *= 1e6
    cases = {"lognormal": base * torch.exp(3 * torch.randn(n_rows, D, generator=gen)),
             "rowscaled": base * torch.exp(4 * torch.randn(n_rows, 1, generator=gen)), "spikes": spikes}
    for name, x in cases.items():
        xd = x.float().to(dev)
        ops.scatter_rows(rels, n_rows, D, xd)
        first = [r.out.clone() for r in rels]
        for _ in range(5):
            ops.scatter_rows(rels, n_rows, D, xd)
            for r, f in zip(rels, first):
                assert torch.equal(r.out, f), name
        for r, ei, w in zip(rels, eis, rs):
            xs = x.float().double() * (w.double()[:, None] if rowscale else 1.0)
            xs = xs.float().double() if rowscale else xs
            ref = torch.zeros(r.n_cols, D, dtype=torch.float64).index_add_(0, ei[1], xs[ei[0]])
            mag = torch.zeros(r.n_cols, D, dtype=torch.float64).index_add_(0, ei[1], xs[ei[0]].abs())
            err = ((r.out.double().cpu() - ref).abs() / mag.clamp(min=1e-300)).max().item()
            assert err <= 1.5e-6, (name, r.n_cols, err)       # (a dozen fp32 additions of terms 1e6 apart)


def test_scatter_f16_pieces_nonfinite_inputs_stay_in_their_column(ops, dev):
    """An infinity or a NaN in x[row, c] makes the sums of that row's items in column c non-finite (the matrix product also
    multiplies it by the 0 of every other item of the block: NaN there too -- the stated deviation from index_add_, as in the
    dense layers); every OTHER column is bitwise what it is without it: a non-finite value takes part in no scale decision."""
    D, n_rows = 128, 3000
    gen = torch.Generator().manual_seed(77)
    rels, eis = _simple_scatter_problem(ops, dev, gen, n_rows, [50, 114, 100], [50, 9, 25], D)
    x = torch.randn(n_rows, D, generator=gen)
    ops.scatter_rows(rels, n_rows, D, x.to(dev))
    clean = [r.out.clone() for r in rels]
    for bad in (float("inf"), float("nan")):
        xb = x.clone()
        xb[1234, 17] = bad
        ops.scatter_rows(rels, n_rows, D, xb.to(dev))
        for r, ei, c in zip(rels, eis, clean):
            hit = torch.zeros(r.n_cols, dtype=torch.bool)
            hit[ei[1][ei[0] == 1234]] = True
            got = r.out.cpu()
            assert not torch.isfinite(got[hit, 17]).any()
            keep = torch.ones(D, dtype=torch.bool)
            keep[17] = False
            assert torch.equal(got[:, keep], c.cpu()[:, keep])


# ------------------------------------------------------------------------------------------ dense
@pytest.mark.parametrize("M,N,K", [(1, 64, 64), (50, 128, 128), (1834, 128, 128), (1834, 64, 128), (1000, 128, 64),
                                   (777, 256, 256), (333, 64, 256), (114, 128, 128), (5000, 256, 128)])
def test_linear_fwd(ops, dev, M, N, K):
    gen = torch.Generator().manual_seed(M + N + K)
    x, W, b = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) / K ** 0.5, torch.randn(N, generator=gen)
    y = ops.linear_fwd(x.to(dev), W.to(dev), b.to(dev))
    ref = x.double() @ W.double().t() + b.double()
    assert rel(y, ref) <= 1e-5
    y2 = ops.linear_fwd(x.to(dev), W.to(dev), None, out=y.clone(), accumulate=True)
    assert rel(y2, 2 * ref - b.double()) <= 1e-5
    # the weight stored [K, N] (data-gradient GEMMs read the forward weight in place): same arithmetic, same bits
    y3 = ops.linear_fwd(x.to(dev), W.t().contiguous().to(dev), b.to(dev), w_kn=True)
    assert torch.equal(y3, y)
    y4, sums = ops.linear_fwd(x.to(dev), W.to(dev), b.to(dev), with_stats=True)   # BatchNorm statistics from the epilogue
    assert torch.equal(y4, y)
    yd = y.double().cpu()
    assert rel(sums[0], yd.sum(0)) <= 1e-6 and rel(sums[1], (yd * yd).sum(0)) <= 1e-6
    assert rel(y, ref) <= 2e-6                     # the 6-term bf16 split keeps fp32 accuracy


@pytest.mark.parametrize("M", [2000, 300])          # the x6 matrix-core kernels / the small-M kernel
@pytest.mark.parametrize("sx,sw", [(1e15, 1e15), (1e-15, 1e-15), (1e30, 1e-30), (3e18, 3e18)])
def test_dense_kernels_over_the_fp32_range(ops, dev, M, sx, sw):
    """The 6-term bf16 split over the whole fp32 exponent range (the reference computes these layers with plain fp32
    addmm, model.py:93-105): operands at 1e+-15 / 1e+-30, products up to ~1e38.  Forward, weight gradient and the fused
    BatchNorm-backward GEMM keep fp32 accuracy relative to the tensor's scale -- nothing in the split depends on the
    magnitude as long as every intermediate stays finite."""
    gen = torch.Generator().manual_seed(int(M))
    N = K = 128
    x, W = torch.randn(M, K, generator=gen) * sx, torch.randn(N, K, generator=gen) * sw / K ** 0.5
    y = ops.linear_fwd(x.to(dev), W.to(dev))
    ref = x.double() @ W.double().t()
    assert bool(torch.isfinite(y).all()) and rel(y, ref) <= 2e-6
    dy = torch.randn(M, N, generator=gen) * min(sw, 1e15)            # (the sum over M rows has to stay below 3.4e38)
    dW = ops.linear_wgrad(dy.to(dev), x.to(dev))
    refw = dy.double().t() @ x.double()
    assert bool(torch.isfinite(dW).all()) and rel(dW, refw) <= 2e-6
    if ops.linear_bnbwd_supported(M, N, K):
        g = torch.randn(M, K, generator=gen) * sx
        yy = torch.randn(M, K, generator=gen)
        Wkn = torch.randn(K, N, generator=gen) * sw / K ** 0.5
        dz, dx = ops.linear_bnbwd(g.to(dev), yy.to(dev), ops.Pro(None, None, True, 0.0), None, Wkn.to(dev))
        dzr = g.double() * (yy.double() > 0)
        assert torch.equal(dz.cpu().double(), dzr)                           # relu mask only: exact
        assert rel(dx, dzr @ Wkn.double()) <= 2e-6


@pytest.mark.parametrize("case", ["unit", "1e15", "1e-15", "1e30x1e-30", "rows", "cols", "sparse_rows"])
def test_k256_layer_on_three_f16_products(ops, dev, case):
    """The [M, 256] x [256, 256] layers of the 256-d model (BASELINE config 4) multiply three f16 products per tile with a
    power-of-two scale per row of X and per row of W (k_linear_fwd_h3_k256).  Same bar as the six-term bf16 kernels -- 2e-6
    of an fp64 reference, per ROW of the output here so that a small row cannot hide behind a large one -- at O(1), at
    1e+-15 / 1e+-30, with rows (and weight rows) whose magnitudes are spread over 40 decades, and with mostly-zero rows;
    with and without the BatchNorm / ReLU / dropout prologue, accumulating, and with the weight stored [K, N]."""
    gen = torch.Generator().manual_seed(len(case))
    M, N, K = 3000, 256, 256
    x, W = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen)
    if case == "1e15":
        x, W, b = x * 1e15, W * 1e15, b * 1e30
    elif case == "1e-15":
        x, W, b = x * 1e-15, W * 1e-15, b * 1e-30
    elif case == "1e30x1e-30":
        x, W = x * 1e30, W * 1e-30
    elif case == "rows":
        x = x * 10.0 ** torch.linspace(-20, 20, M)[:, None]
        b = b * 0
    elif case == "cols":
        W = W * 10.0 ** torch.linspace(-18, 18, N)[:, None]
        b = b * 0
    elif case == "sparse_rows":                       # ReLU / dropout style inputs: most entries exactly zero
        x = x * (torch.rand(M, K, generator=gen) < 0.05)
        x[5] = 0
    ref = x.double() @ W.double().t() + b.double()
    rowmax = ref.abs().amax(1, keepdim=True).clamp(min=1e-300)
    colmax = ref.abs().amax(0, keepdim=True).clamp(min=1e-300)

    def bar(y, r, tol=2e-6):
        e = (y.double().cpu() - r).abs()
        return bool(torch.isfinite(y).all()) and float((e / rowmax).max()) <= tol and (case != "cols" or float((e / colmax).max()) <= tol)

    y = ops.linear_fwd(x.to(dev), W.to(dev), b.to(dev))
    assert bar(y, ref)
    y3 = ops.linear_fwd(x.to(dev), W.t().contiguous().to(dev), b.to(dev), w_kn=True)
    assert torch.equal(y3, y)
    y2 = ops.linear_fwd(x.to(dev), W.to(dev), None, out=y.clone(), accumulate=True)
    assert bar(y2, 2 * ref - b.double(), 4e-6)
    if case in ("unit", "sparse_rows"):
        scale, shift = torch.rand(K, generator=gen) + 0.5, torch.randn(K, generator=gen) * 0.3
        for p_drop in (0.0, 0.2):
            pro = ops.Pro(scale.to(dev), shift.to(dev), True, p_drop, 77, 5)
            yp = ops.linear_fwd(x.to(dev), W.to(dev), b.to(dev), pro=pro)
            xp = _host_pro(ops, dev, x, scale, shift, True, p_drop, 77, 5)
            refp = xp @ W.double().t() + b.double()
            assert rel(yp, refp) <= 2e-6


def test_dense_kernels_denormals_and_non_finite_inputs(ops, dev):
    """Edges of the range.  (1) Operands within ~2^8 of the smallest normal fp32 (1.18e-38): the low pieces of the split
    are bf16 denormals, which the matrix cores may flush, so the result degrades towards the first piece alone -- the
    error stays below the smallest normal number in absolute terms, but it is NOT fp32-accurate relative to such values
    (fp32 addmm underflows gradually).  Stated, not hidden: activations of this model are O(1).  (2) Non-finite inputs propagate: every output fp32 addmm makes non-finite is
    non-finite here, every other output is untouched; NaN in -> NaN out.  An inf input row comes out as NaN rather than
    +-inf (hi = inf makes the remainder inf - inf): non-finite either way, and stated here so nobody reads it as parity."""
    gen = torch.Generator().manual_seed(3)
    M, N, K = 1500, 128, 128
    # (1) tiny magnitudes
    x = torch.randn(M, K, generator=gen) * 1e-37
    W = torch.randn(N, K, generator=gen) / K ** 0.5
    y = ops.linear_fwd(x.to(dev), W.to(dev))
    ref = x.double() @ W.double().t()
    assert float((y.double().cpu() - ref).abs().max()) <= 1.2e-38                  # below the smallest normal
    xd = torch.full((M, K), 1e-41)                                                  # fp32 denormals in
    yd = ops.linear_fwd(xd.to(dev), W.to(dev))
    assert bool(torch.isfinite(yd).all()) and float(yd.abs().max()) <= 1e-38        # (flushed or kept: both below normal)
    # (2) non-finite inputs
    x = torch.randn(M, K, generator=gen)
    x[3, 5], x[700, 9], x[1499, 127] = float("inf"), float("nan"), float("-inf")
    cpu = x @ W.t()                                                                 # fp32 addmm on the host
    y = ops.linear_fwd(x.to(dev), W.to(dev)).cpu()
    bad_rows = torch.tensor([3, 700, 1499])
    assert not bool(torch.isfinite(cpu[bad_rows]).any()) and not bool(torch.isfinite(y[bad_rows]).any())
    assert bool(torch.isnan(y[700]).all())
    ok_rows = torch.ones(M, dtype=torch.bool)
    ok_rows[bad_rows] = False
    assert bool(torch.isfinite(y[ok_rows]).all()) and rel(y[ok_rows], cpu[ok_rows].double()) <= 2e-6
    # the weight gradient sums over rows: a non-finite row poisons exactly the columns fp32 would poison (all of them here)
    dy = torch.randn(M, N, generator=gen)
    dW = ops.linear_wgrad(dy.to(dev), x.to(dev)).cpu()
    cw = dy.t() @ x
    assert torch.equal(torch.isfinite(dW), torch.isfinite(cw))
    fin = torch.isfinite(cw)
    assert rel(dW[fin], cw[fin].double()) <= 1e-5


def _host_pro(ops, dev, x, scale, shift, relu, p, seed, site, row_offset=0):
    v = x.double() * scale.double() + shift.double()
    if relu:
        v = v.clamp(min=0)
    if p > 0:
        m = ops.dropout_mask(seed, site, x.shape[0], x.shape[1], p, dev, row_offset).cpu().double()
        v = v * m / (1 - p)
    return v


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_linear_fwd_prologue_and_wgrad(ops, dev, p):
    gen = torch.Generator().manual_seed(11)
    M, N, K = 1500, 128, 128
    x, W = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) / K ** 0.5
    sc, sh = torch.rand(K, generator=gen) + 0.5, torch.randn(K, generator=gen) * 0.3
    pro = ops.Pro(sc.to(dev), sh.to(dev), True, p, seed=99, site=3, row_offset=40)
    xp = _host_pro(ops, dev, x, sc, sh, True, p, 99, 3, 40)
    if p > 0:
        keep = float((xp != 0).double().mean() / (x.double() * sc.double() + sh.double() > 0).double().mean())
        assert abs(keep - (1 - p)) < 0.02
    y = ops.linear_fwd(x.to(dev), W.to(dev), None, pro=pro)
    assert rel(y, xp @ W.double().t()) <= 1e-5
    mat = ops.affine_act_drop(x.to(dev), pro)
    assert rel(mat, xp) <= 1e-6
    dy = torch.randn(M, N, generator=gen)
    dW = ops.linear_wgrad(dy.to(dev), x.to(dev), pro)
    assert rel(dW, dy.double().t() @ xp) <= 1e-5
    dW2 = ops.linear_wgrad(dy.to(dev), x.to(dev), pro, out=dW.clone(), accumulate=True)
    assert rel(dW2, 2 * (dy.double().t() @ xp)) <= 1e-5


@pytest.mark.parametrize("M,N,K", [(50, 64, 128), (1834, 128, 128), (9000, 256, 256), (3, 128, 64), (264, 128, 128)])
def test_linear_wgrad_shapes(ops, dev, M, N, K):
    gen = torch.Generator().manual_seed(M)
    dy, x = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
    dW = ops.linear_wgrad(dy.to(dev), x.to(dev))
    assert rel(dW, dy.double().t() @ x.double()) <= 1e-5
    dW2, db = ops.linear_wgrad(dy.to(dev), x.to(dev), with_bias=True)       # bias gradient from the same pass
    assert torch.equal(dW2, dW)
    assert rel(db, dy.double().sum(0)) <= 2e-6


@pytest.mark.parametrize("M,N", [(1, 128), (50, 128), (1834, 64), (100000, 128), (777, 256)])
def test_col_reduce_and_bn_finalize(ops, dev, M, N):
    gen = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, N, generator=gen) * 2 + 0.7
    b = torch.randn(M, N, generator=gen)
    s = ops.col_reduce2(a.to(dev), b.to(dev)).cpu()
    assert rel(s[0], a.double().sum(0)) <= 1e-12 and rel(s[1], (a.double() * b.double()).sum(0)) <= 1e-12
    s2 = ops.col_reduce2(a.to(dev)).cpu()
    assert rel(s2[1], (a.double() ** 2).sum(0)) <= 1e-12
    if M > 1:
        bn = torch.nn.BatchNorm1d(N)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5, generator=gen); bn.bias.uniform_(-0.3, 0.3, generator=gen)
            bn.running_mean.uniform_(-0.2, 0.2, generator=gen); bn.running_var.uniform_(0.5, 1.5, generator=gen)
        rm, rv = bn.running_mean.clone().to(dev), bn.running_var.clone().to(dev)
        fold = ops.bn_finalize(ops.col_reduce2(a.to(dev)), M, bn.weight.detach().to(dev), bn.bias.detach().to(dev),
                               rm, rv, True, 2)
        y = ops.affine_act_drop(a.to(dev), ops.Pro(fold.scale, fold.shift, False))
        bn.train()
        ref = bn(a); bn(a)           # two updates (F7)
        assert rel(y, ref.detach()) <= 2e-5
        assert rel(rm, bn.running_mean) <= 1e-5 and rel(rv, bn.running_var) <= 1e-5
        bn.eval()
        fold_e = ops.bn_finalize(None, M, bn.weight.detach().to(dev), bn.bias.detach().to(dev), rm, rv, False)
        ye = ops.affine_act_drop(a.to(dev), ops.Pro(fold_e.scale, fold_e.shift, False))
        assert rel(ye, bn(a).detach()) <= 2e-5


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_bn_relu_dropout_backward(ops, dev, p):
    gen = torch.Generator().manual_seed(21)
    M, N = 3000, 128
    y = (torch.randn(M, N, generator=gen) * 1.5 + 0.2)
    g = torch.randn(M, N, generator=gen)
    gamma, beta = torch.rand(N, generator=gen) + 0.5, torch.randn(N, generator=gen) * 0.2
    fold = ops.bn_finalize(ops.col_reduce2(y.to(dev)), M, gamma.to(dev), beta.to(dev), None, None, True)
    pro = ops.Pro(fold.scale, fold.shift, True, p, seed=5, site=17)
    # reference in fp64 autograd with the device's own dropout mask injected
    yd = y.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    xh = (yd - yd.mean(0)) / torch.sqrt(yd.var(0, unbiased=False) + 1e-5)
    out = torch.relu(xh * gd + bd)
    if p > 0:
        out = out * ops.dropout_mask(5, 17, M, N, p, dev).cpu().double() / (1 - p)
    out.backward(g.double())
    sums = ops.bn_bwd_stats(g.to(dev), y.to(dev), pro, fold)
    assert rel(sums[0], bd.grad) <= 1e-6 and rel(sums[1], gd.grad) <= 1e-5
    dbg = torch.empty(2, N, device=dev)
    dy = ops.bn_bwd_apply(g.to(dev), y.to(dev), pro, fold, sums, M, dbg[0], dbg[1])
    assert rel(dy, yd.grad) <= 2e-5
    assert torch.equal(dbg.double(), sums.float().double())        # d beta / d gamma ride along
    acc = torch.full((M, N), 0.5, device=dev)                      # accumulate: the result is ADDED to the target
    ops.bn_bwd_apply(g.to(dev), y.to(dev), pro, fold, sums, M, out=acc, accumulate=True)
    assert torch.equal(acc, dy + 0.5)


@pytest.mark.parametrize("M,K,N", [(5003, 128, 128), (4100, 128, 64), (3333, 64, 128), (600, 64, 64), (300, 128, 128),
                                   (2000, 256, 256)])
@pytest.mark.parametrize("with_pro", [False, True])
def test_l2_norm_inside_the_dense_forward(ops, dev, M, K, N, with_pro):
    """mmg_linear_fwd_l2norm (row norm in the GEMM epilogue) == mmg_linear_fwd followed by mmg_l2norm_fwd up to the order
    of the 128 squares of a row, and == F.normalize of an fp64 product; unsupported shapes take the two launches."""
    gen = torch.Generator().manual_seed(5 + M)
    x = torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    pro = None
    if with_pro:
        sc, sh = (torch.rand(K, generator=gen) + 0.5).to(dev), (torch.randn(K, generator=gen) * 0.3).to(dev)
        pro = ops.Pro(sc, sh, True, 0.25, seed=11, site=4, row_offset=77)
    out, rn = ops.linear_l2norm_fwd(x, W, b, pro)
    z = ops.linear_fwd(x, W, b, pro=pro)
    out_ref, rn_ref = ops.l2norm_fwd(z)
    assert rel(out, out_ref) <= 5e-7 and rel(rn, rn_ref) <= 5e-7
    assert rel(out, torch.nn.functional.normalize(z.double(), p=2, dim=1)) <= 5e-7
    assert rel(out.norm(dim=1), torch.ones(M, device=dev)) <= 1e-6
    zero = torch.zeros(M, K, device=dev)                           # a zero row: y = 0 -> out = 0, rn = 1 / eps (F.normalize's clamp)
    o0, r0 = ops.linear_l2norm_fwd(zero, W, None)
    assert float(o0.abs().max()) == 0.0 and torch.equal(r0, torch.full((M,), 1.0 / ops.L2_EPS, device=dev))


@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("n_sel", [0, 1, 157])
def test_row_list_bn_backward_inside_the_data_gradient_gemm(ops, dev, p, n_sel):
    """mmg_linear_bnbwd_rows == mmg_bn_bwd_apply(G = NULL) + mmg_bn_bwd_apply_rows + mmg_linear_fwd(W_KN): the upstream
    gradient is zero outside a row list (rows of the last, partial tile included)."""
    gen = torch.Generator().manual_seed(321 + n_sel)
    M, K, N = 4999, 128, 128
    y = (torch.randn(M, K, generator=gen) * 1.5 + 0.2).to(dev)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    rows = torch.randperm(M, generator=gen)[:n_sel].sort().values
    if n_sel > 1:
        rows[-1] = M - 1                                                # a row of the tail tile
        rows = rows.unique()
    rows = rows.to(dev)
    g_rows = torch.randn(rows.numel(), K, generator=gen).to(dev)
    row_pos = torch.full((M,), -1, dtype=torch.int32, device=dev)
    row_pos[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=dev)
    gamma, beta = (torch.rand(K, generator=gen) + 0.5).to(dev), (torch.randn(K, generator=gen) * 0.2).to(dev)
    fold = ops.bn_finalize(ops.col_reduce2(y), M, gamma, beta, None, None, True)
    pro = ops.Pro(fold.scale, fold.shift, True, p, seed=9, site=1, row_offset=10)
    sums = ops.bn_bwd_stats_rows(g_rows, y, rows, pro, fold) if rows.numel() else torch.zeros(2, K, dtype=torch.float64, device=dev)
    d0, d1 = torch.zeros(2, K, device=dev), torch.zeros(2, K, device=dev)
    dz_ref = ops.bn_bwd_apply(None, y, pro, fold, sums, M, d0[0], d0[1])
    if rows.numel():
        ops.bn_bwd_apply_rows(g_rows, y, rows, pro, dz_ref)
    dz, dx = ops.linear_bnbwd_rows(g_rows, row_pos, y, pro, fold, W, sums, M, d1[0], d1[1])
    assert torch.equal(d0, d1)
    mask = torch.ones(M, dtype=torch.bool, device=dev)
    mask[rows] = False
    assert torch.equal(dz[mask], dz_ref[mask])                          # rows without a gradient: the same arithmetic
    assert rel(dz, dz_ref) <= 1e-6                                      # listed rows: scale * (g' - ...) in one expression
    assert rel(dx, ops.linear_fwd(dz_ref, W, w_kn=True)) <= 2e-6


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_joint_bn_backward_inside_the_data_gradient_gemm(ops, dev, p):
    """mmg_linear_bnbwd2 == mmg_bn_bwd_apply2 followed by mmg_linear_fwd(W_KN), bit for bit."""
    gen = torch.Generator().manual_seed(123)
    M, K, N = 4999, 128, 128
    y = (torch.randn(M, K, generator=gen) * 1.5 + 0.2).to(dev)
    g, g2 = torch.randn(M, K, generator=gen).to(dev), torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    gamma, beta = (torch.rand(K, generator=gen) + 0.5).to(dev), (torch.randn(K, generator=gen) * 0.2).to(dev)
    fold = ops.bn_finalize(ops.col_reduce2(y), M, gamma, beta, None, None, True)
    pro = ops.Pro(fold.scale, fold.shift, True, p, seed=9, site=0, row_offset=10)
    pro2 = ops.Pro(fold.scale, fold.shift, True, p, seed=9, site=2, row_offset=10)
    sums = ops.bn_bwd_stats2(g, g2, y, pro, pro2, fold)
    d0, d1 = torch.zeros(2, K, device=dev), torch.zeros(2, K, device=dev)
    dz_ref = ops.bn_bwd_apply2(g, g2, y, pro, pro2, fold, sums, M, d0[0], d0[1])
    assert ops.linear_bnbwd2_supported(M, N, K) and not ops.linear_bnbwd2_supported(M, 64, K)
    dz, dx = ops.linear_bnbwd2(g, g2, y, pro, pro2, fold, W, sums, M, d1[0], d1[1])
    assert torch.equal(dz, dz_ref) and torch.equal(d0, d1)
    assert torch.equal(dx, ops.linear_fwd(dz_ref, W, w_kn=True))


@pytest.mark.parametrize("M,K,N", [(5003, 128, 128), (4100, 128, 64), (3333, 64, 128), (600, 64, 64), (300, 128, 128)])
def test_l2_norm_backward_inside_the_data_gradient_gemm(ops, dev, M, K, N):
    """mmg_linear_l2bwd == mmg_l2norm_bwd followed by mmg_linear_fwd(W_KN) (the row dot product is summed in another
    order), against fp64 autograd of F.normalize; a clamped (all-zero) row keeps dz = g / eps."""
    gen = torch.Generator().manual_seed(9 + M)
    z = torch.randn(M, K, generator=gen)
    z[7] = 0.0                                                           # norm <= eps: F.normalize divides by the constant eps
    g = torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    out, rn = ops.l2norm_fwd(z.to(dev))
    dz, dx = ops.linear_l2bwd(g, out, rn, W)
    dz_ref = ops.l2norm_bwd(g, out, rn)
    assert rel(dz, dz_ref) <= 1e-6 and torch.equal(dz[7], dz_ref[7])
    assert rel(dx, ops.linear_fwd(dz_ref, W, w_kn=True)) <= 2e-6
    zd = z.double().requires_grad_(True)
    torch.nn.functional.normalize(zd, p=2, dim=1, eps=ops.L2_EPS).backward(g.cpu().double())
    keep = torch.ones(M, dtype=torch.bool); keep[7] = False              # (row 7: huge values, compared above)
    assert rel(dz.cpu()[keep], zd.grad[keep]) <= 2e-6
    assert rel(dx.cpu()[keep], zd.grad[keep] @ W.cpu().double()) <= 2e-6


@pytest.mark.parametrize("M,K,N", [(5003, 128, 128), (4100, 128, 64), (3333, 64, 128), (600, 64, 64)])
@pytest.mark.parametrize("p,mode", [(0.0, "train"), (0.3, "train"), (0.3, "eval"), (0.3, "nobn")])
def test_bn_backward_inside_the_data_gradient_gemm(ops, dev, M, K, N, p, mode):
    """mmg_linear_bnbwd == mmg_bn_bwd_apply followed by mmg_linear_fwd(W_KN) on its output, bit for bit (the same
    arithmetic in the same order: the BatchNorm backward is computed while the tile is staged), incl. d beta / d gamma,
    the tail tile, a row offset of the dropout stream, eval mode and no BatchNorm at all."""
    gen = torch.Generator().manual_seed(77 + M)
    y = (torch.randn(M, K, generator=gen) * 1.5 + 0.2).to(dev)
    g = torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    gamma, beta = (torch.rand(K, generator=gen) + 0.5).to(dev), (torch.randn(K, generator=gen) * 0.2).to(dev)
    assert ops.linear_bnbwd_supported(M, N, K) and not ops.linear_bnbwd_supported(200, N, K)
    assert not ops.linear_bnbwd_supported(M, 256, K) and not ops.linear_bnbwd_supported(M, N, 256)
    if mode == "nobn":
        fold, pro, sums = None, ops.Pro(None, None, True, p, seed=9, site=3, row_offset=1000), None
    else:
        training = mode == "train"
        rm, rv = torch.zeros(K, device=dev), torch.ones(K, device=dev)
        fold = ops.bn_finalize(ops.col_reduce2(y) if training else None, M, gamma, beta, rm, rv, training)
        pro = ops.Pro(fold.scale, fold.shift, True, p, seed=9, site=3, row_offset=1000)
        sums = ops.bn_bwd_stats(g, y, pro, fold) if training else None
    d0, d1 = torch.zeros(2, K, device=dev), torch.zeros(2, K, device=dev)
    if sums is not None:
        dz_ref = ops.bn_bwd_apply(g, y, pro, fold, sums, M, d0[0], d0[1])
        dz, dx = ops.linear_bnbwd(g, y, pro, fold, W, sums, M, d1[0], d1[1])
    else:
        dz_ref = ops.bn_bwd_apply(g, y, pro, fold)
        dz, dx = ops.linear_bnbwd(g, y, pro, fold, W)
    dx_ref = ops.linear_fwd(dz_ref, W, w_kn=True)
    assert torch.equal(dz, dz_ref) and torch.equal(d0, d1)
    assert torch.equal(dx, dx_ref)
    assert rel(dx, dz_ref.double() @ W.double()) <= 2e-6


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_row_subset_variants_match_the_dense_kernels(ops, dev, p):
    """affine_act_drop_rows / bn_bwd_stats_rows / bn_bwd_apply(None) + bn_bwd_apply_rows == the dense kernels fed
    with a gradient that is zero outside the listed rows (dropout masks of the ORIGINAL rows)."""
    gen = torch.Generator().manual_seed(33)
    M, N = 2500, 128
    y = (torch.randn(M, N, generator=gen) * 1.5 + 0.2).to(dev)
    rows = torch.randperm(M, generator=gen)[:173].sort().values.to(dev)
    g_rows = torch.randn(rows.numel(), N, generator=gen).to(dev)
    gamma, beta = (torch.rand(N, generator=gen) + 0.5).to(dev), (torch.randn(N, generator=gen) * 0.2).to(dev)
    fold = ops.bn_finalize(ops.col_reduce2(y), M, gamma, beta, None, None, True)
    pro = ops.Pro(fold.scale, fold.shift, True, p, seed=9, site=4, row_offset=11)
    assert torch.equal(ops.affine_act_drop_rows(y, pro, rows), ops.affine_act_drop(y, pro)[rows])
    g = torch.zeros(M, N, device=dev)
    g[rows] = g_rows
    sums_d = ops.bn_bwd_stats(g, y, pro, fold)
    sums_r = ops.bn_bwd_stats_rows(g_rows, y, rows, pro, fold)
    assert rel(sums_r, sums_d.cpu()) <= 1e-12
    dy_d = ops.bn_bwd_apply(g, y, pro, fold, sums_d, M)
    dy_r = ops.bn_bwd_apply(None, y, pro, fold, sums_d, M)
    ops.bn_bwd_apply_rows(g_rows, y, rows, pro, dy_r)
    assert rel(dy_r, dy_d.cpu()) <= 1e-6
    empty = rows[:0]
    assert ops.affine_act_drop_rows(y, pro, empty).shape == (0, N)
    assert float(ops.bn_bwd_stats_rows(g_rows[:0], y, empty, pro, fold).abs().sum()) == 0.0


def test_two_upstream_gradients_through_one_batchnorm(ops, dev):
    """bn_bwd_stats2 / bn_bwd_apply2 == the sum of two single backward passes (same BN + ReLU, own dropout masks)."""
    gen = torch.Generator().manual_seed(44)
    M, N = 40000, 128
    y = (torch.randn(M, N, generator=gen) * 1.5 + 0.2).to(dev)
    ga, gb = torch.randn(M, N, generator=gen).to(dev), torch.randn(M, N, generator=gen).to(dev)
    gamma, beta = (torch.rand(N, generator=gen) + 0.5).to(dev), (torch.randn(N, generator=gen) * 0.2).to(dev)
    fold = ops.bn_finalize(ops.col_reduce2(y), M, gamma, beta, None, None, True)
    pa = ops.Pro(fold.scale, fold.shift, True, 0.2, seed=9, site=0, row_offset=5)
    pb = ops.Pro(fold.scale, fold.shift, True, 0.2, seed=9, site=2, row_offset=5)
    sa, sb = ops.bn_bwd_stats(ga, y, pa, fold), ops.bn_bwd_stats(gb, y, pb, fold)
    s2 = ops.bn_bwd_stats2(ga, gb, y, pa, pb, fold)
    assert rel(s2, (sa + sb).cpu()) <= 1e-6        # (the two masked gradients are added in fp32 before the fp64 sums)
    ref = ops.bn_bwd_apply(ga, y, pa, fold, sa, M) + ops.bn_bwd_apply(gb, y, pb, fold, sb, M)
    dbg = torch.empty(2, N, device=dev)
    dy = ops.bn_bwd_apply2(ga, gb, y, pa, pb, fold, s2, M, dbg[0], dbg[1])
    assert rel(dy, ref.cpu()) <= 1e-5
    assert torch.equal(dbg.double(), s2.float().double())


@pytest.mark.parametrize("N", [64, 128, 256])
def test_l2norm(ops, dev, N):
    gen = torch.Generator().manual_seed(N)
    z = torch.randn(500, N, generator=gen)
    z[3] = 0.0                      # clamp branch: ||z|| <= eps
    z[4] = 1e-20
    out, rn = ops.l2norm_fwd(z.to(dev))
    zd = z.double().requires_grad_(True)
    ref = torch.nn.functional.normalize(zd, p=2, dim=1, eps=1e-12)
    assert rel(out, ref.detach()) <= 1e-6
    g = torch.randn(500, N, generator=gen)
    ref.backward(g.double())
    dz = ops.l2norm_bwd(g.to(dev), out, rn)
    good = torch.ones(500, dtype=torch.bool); good[3] = False
    assert rel(dz[good.to(dev)], zd.grad[good]) <= 1e-5
    assert torch.isfinite(dz).all()


# ------------------------------------------------------------------------------------------ heads
@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("sorted_pairs,L", [(False, 50), (True, 50), (True, 100), (True, 200)])
def test_pair_head_fwd_bwd(ops, dev, p, sorted_pairs, L):
    """L = labs: up to 64 the backward is k_pair_bwd_duo (two waves per tile), up to 128 k_pair_bwd_mfma<4>, beyond that
    the fp32 kernel with the lab table in LDS -- three kernels, one contract."""
    gen = torch.Generator().manual_seed(31)
    P, n = 400, 5000
    A, B = torch.randn(P, 64, generator=gen), torch.randn(L, 64, generator=gen)
    W2, b2 = torch.randn(32, 64, generator=gen) / 8, torch.randn(32, generator=gen) * 0.1
    W3, b3 = torch.randn(32, generator=gen) / 5, torch.randn(1, generator=gen)
    pi = torch.randint(0, P, (n,), generator=gen)
    if sorted_pairs:
        pi = pi.sort().values
    li = torch.randint(0, L, (n,), generator=gen)
    deg = torch.randint(0, 12, (P,), generator=gen)
    pid = torch.randperm(n, generator=gen)
    dpred = torch.randn(n, generator=gen)
    head = ops.Head(*[t.to(dev) for t in (A, B, W2, b2, W3, b3)])
    i32 = lambda t: t.to(torch.int32).to(dev)
    for want_low in (False, True):
        pred = torch.full((n,), 123.0, device=dev)
        ops.pair_head_fwd(head, i32(pi), i32(li), i32(deg), 6, want_low, p, 77, pid.to(dev), pred)
        sel = (deg[pi] < 6) == want_low
        leaf = [t.double().requires_grad_(True) for t in (A, B, W2, b2, W3, b3)]
        h1 = torch.relu(leaf[0][pi] + leaf[1][li])
        if p > 0:
            m1 = ops.dropout_mask(77, 64, n, 64, p, dev).cpu().double()[pid]
            h1 = h1 * m1 / (1 - p)
        h2 = torch.relu(h1 @ leaf[2].t() + leaf[3])
        if p > 0:
            m2 = ops.dropout_mask(77, 65, n, 32, p, dev).cpu().double()[pid]
            h2 = h2 * m2 / (1 - p)
        ref = h2 @ leaf[4] + leaf[5]
        assert rel(pred[sel.to(dev)], ref.detach()[sel]) <= 1e-5
        assert bool((pred[(~sel).to(dev)] == 123.0).all())
        (ref * dpred.double() * sel.double()).sum().backward()
        g = ops.Head(*[torch.zeros_like(t, device=dev) for t in (A, B, W2, b2, W3, b3)])
        ops.pair_head_bwd(head, g, i32(pi), i32(li), i32(deg), 6, want_low, L, p, 77, pid.to(dev), dpred.to(dev))
        for name, got, want in zip("A B W2 b2 W3 b3".split(), (g.A, g.B, g.W2, g.b2, g.W3, g.b3), leaf):
            assert rel(got, want.grad) <= 2e-5, (name, want_low)

@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("listed", [False, True, "same"])
def test_pair_head_backward_from_the_state_the_forward_saved(ops, dev, p, listed):
    """mmg_pair_head_fwd_save / mmg_pair_head_bwd_saved: the forward leaves the first layer's sign bits and the second
    layer's activations per visited pair, the backward reads them instead of recomputing masks and the 64 x 32 product --
    same gradients as the fp64 reference (and as the recomputing backward), full sweep and compacted lists (the backward
    visits a SUBSET of the forward's pairs: those with a non-zero upstream gradient)."""
    gen = torch.Generator().manual_seed(57)
    P, L, n = 700, 50, 9000
    A, B = torch.randn(P, 64, generator=gen), torch.randn(L, 64, generator=gen)
    W2, b2 = torch.randn(32, 64, generator=gen) / 8, torch.randn(32, generator=gen) * 0.1
    W3, b3 = torch.randn(32, generator=gen) / 5, torch.randn(1, generator=gen)
    pi = torch.randint(0, P, (n,), generator=gen).sort().values
    li = torch.randint(0, L, (n,), generator=gen)
    deg = torch.randint(0, 12, (P,), generator=gen)
    pid = torch.randperm(n, generator=gen)
    dpred = torch.randn(n, generator=gen) * (torch.rand(n, generator=gen) < 0.3)       # ~70 % of the pairs: exactly 0
    head = ops.Head(*[t.to(dev) for t in (A, B, W2, b2, W3, b3)])
    i32 = lambda t: t.to(torch.int32).to(dev)
    for want_low in (False, True):
        save = ops.pair_saved_alloc(n, dev)
        save[0].fill_(-1); save[1].fill_(float("nan"))                                 # an entry that was not written shows
        pred = torch.zeros(n, device=dev)
        fsel = bsel = None
        if listed:
            lo, hi, cnt = ops.pair_select(i32(pi), i32(deg), 6, None)
            blo, bhi, bcnt = ops.pair_select(i32(pi), i32(deg), 6, dpred.to(dev))
            fsel = dict(sel=lo if want_low else hi, n_sel=cnt[0:1] if want_low else cnt[1:2], n_bound=n)
            bsel = dict(sel=blo if want_low else bhi, n_sel=bcnt[0:1] if want_low else bcnt[1:2], n_bound=n)
            if listed == "same":            # forward and backward over the SAME list: the state may be indexed by list position
                fsel = bsel
                save = save + (True,)
        ops.pair_head_fwd(head, i32(pi), i32(li), i32(deg), 6, want_low, p, 77, pid.to(dev), pred, save=save, **(fsel or {}))
        ref_pred = torch.zeros(n, device=dev)
        ops.pair_head_fwd(head, i32(pi), i32(li), i32(deg), 6, want_low, p, 77, pid.to(dev), ref_pred, **(fsel or {}))
        assert torch.equal(pred, ref_pred)                                             # saving changes nothing in the forward
        g1 = ops.Head(*[torch.zeros_like(t, device=dev) for t in (A, B, W2, b2, W3, b3)])
        g0 = ops.Head(*[torch.zeros_like(t, device=dev) for t in (A, B, W2, b2, W3, b3)])
        ops.pair_head_bwd(head, g1, i32(pi), i32(li), i32(deg), 6, want_low, L, p, 77, pid.to(dev), dpred.to(dev),
                          saved=save, **(bsel or {}))
        ops.pair_head_bwd(head, g0, i32(pi), i32(li), i32(deg), 6, want_low, L, p, 77, pid.to(dev), dpred.to(dev),
                          **(bsel or {}))
        sel = (deg[pi] < 6) == want_low
        leaf = [t.double().requires_grad_(True) for t in (A, B, W2, b2, W3, b3)]
        h1 = torch.relu(leaf[0][pi] + leaf[1][li])
        if p > 0:
            h1 = h1 * ops.dropout_mask(77, 64, n, 64, p, dev).cpu().double()[pid] / (1 - p)
        h2 = torch.relu(h1 @ leaf[2].t() + leaf[3])
        if p > 0:
            h2 = h2 * ops.dropout_mask(77, 65, n, 32, p, dev).cpu().double()[pid] / (1 - p)
        ((h2 @ leaf[4] + leaf[5]) * dpred.double() * sel.double()).sum().backward()
        for name, got, rec, want in zip("A B W2 b2 W3 b3".split(), (g1.A, g1.B, g1.W2, g1.b2, g1.W3, g1.b3),
                                        (g0.A, g0.B, g0.W2, g0.b2, g0.W3, g0.b3), leaf):
            assert rel(got, want.grad) <= 2e-5, (name, want_low)
            # the recomputing kernel takes the layer-2 product in the forward's own order: what it recomputes IS what the
            # forward saved, bit for bit (dA: the same run sums, added by atomics whose order is free only where a patient's
            # pairs straddle two tiles -- two partial sums, a + b either way)
            assert torch.equal(got, rec), (name, want_low)


@pytest.mark.parametrize("n", [1, 2047, 2048, 70001])
def test_pair_select_stable_two_way(ops, dev, n):
    gen = torch.Generator().manual_seed(n)
    P = 300
    pi = torch.randint(0, P, (n,), generator=gen).sort().values
    deg = torch.randint(0, 12, (P,), generator=gen)
    dpred = torch.randn(n, generator=gen) * (torch.rand(n, generator=gen) < 0.2)
    i32 = lambda t: t.to(torch.int32).to(dev)
    for dp in (None, dpred):
        lo, hi, cnt = ops.pair_select(i32(pi), i32(deg), 6, None if dp is None else dp.to(dev))
        live = torch.ones(n, dtype=torch.bool) if dp is None else dp != 0
        want_lo = torch.nonzero(live & (deg[pi] < 6)).squeeze(1)
        want_hi = torch.nonzero(live & (deg[pi] >= 6)).squeeze(1)
        assert cnt.tolist() == [want_lo.numel(), want_hi.numel()]
        assert torch.equal(lo[:want_lo.numel()].cpu().long(), want_lo)          # bit-exact, order kept
        assert torch.equal(hi[:want_hi.numel()].cpu().long(), want_hi)


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_pair_head_selected_lists_match_full_sweep(ops, dev, p):
    """Heads driven by compacted lists (forward: per head; backward: per head and non-zero upstream gradient) give
    what the predicated sweep over all pairs gives."""
    gen = torch.Generator().manual_seed(5)
    P, L, n = 500, 50, 9000
    A, B = torch.randn(P, 64, generator=gen), torch.randn(L, 64, generator=gen)
    W2, b2 = torch.randn(32, 64, generator=gen) / 8, torch.randn(32, generator=gen) * 0.1
    W3, b3 = torch.randn(32, generator=gen) / 5, torch.randn(1, generator=gen)
    pi = torch.randint(0, P, (n,), generator=gen).sort().values
    li = torch.randint(0, L, (n,), generator=gen)
    deg = torch.randint(0, 12, (P,), generator=gen)
    pid = torch.randperm(n, generator=gen).to(dev)
    dpred = (torch.randn(n, generator=gen) * (torch.rand(n, generator=gen) < 0.2)).to(dev)
    head = ops.Head(*[t.to(dev) for t in (A, B, W2, b2, W3, b3)])
    i32 = lambda t: t.to(torch.int32).to(dev)
    pi_d, li_d, deg_d = i32(pi), i32(li), i32(deg)
    flo, fhi, fcnt = ops.pair_select(pi_d, deg_d, 6)
    blo, bhi, bcnt = ops.pair_select(pi_d, deg_d, 6, dpred)
    nf = fcnt.tolist()
    full, listed = torch.zeros(n, device=dev), torch.full((n,), -7.0, device=dev)
    for want_low in (False, True):
        ops.pair_head_fwd(head, pi_d, li_d, deg_d, 6, want_low, p, 9, pid, full)
        sel, k = (flo, 0) if want_low else (fhi, 1)
        ops.pair_head_fwd(head, pi_d, li_d, deg_d, 6, want_low, p, 9, pid, listed, sel=sel, n_sel=fcnt[k:k + 1],
                          n_bound=nf[k])
    assert torch.equal(full, listed)                      # same arithmetic per pair: bitwise equal
    for want_low in (False, True):
        g0 = ops.Head(*[torch.zeros_like(t, device=dev) for t in (A, B, W2, b2, W3, b3)])
        g1 = ops.Head(*[torch.zeros_like(t, device=dev) for t in (A, B, W2, b2, W3, b3)])
        ops.pair_head_bwd(head, g0, pi_d, li_d, deg_d, 6, want_low, L, p, 9, pid, dpred)
        sel, k = (blo, 0) if want_low else (bhi, 1)
        ops.pair_head_bwd(head, g1, pi_d, li_d, deg_d, 6, want_low, L, p, 9, pid, dpred, sel=sel, n_sel=bcnt[k:k + 1],
                          n_bound=nf[k])
        for name in "A B W2 b2 W3 b3".split():
            assert rel(getattr(g1, name), getattr(g0, name).double().cpu()) <= 2e-5, (name, want_low)


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_pair_heads_never_turn_a_bad_list_entry_into_an_address(ops, dev, p):
    """Every indexed access of the pair kernels is range-checked (buffer descriptors sized on the host from n_total,
    n_patients, n_labs).  A list with entries outside the pair arrays (negative, past the end), a stale count that is
    larger than the list, patient / lab ids outside their tables and io_perm slots outside pred: nothing of it is read as
    an address -- the bad entries contribute nothing and the good ones give what the clean call gives, bit for bit."""
    gen = torch.Generator().manual_seed(15)
    P, L, n = 300, 40, 4000
    A, B = torch.randn(P, 64, generator=gen), torch.randn(L, 64, generator=gen)
    W2, b2 = torch.randn(32, 64, generator=gen) / 8, torch.randn(32, generator=gen) * 0.1
    W3, b3 = torch.randn(32, generator=gen) / 5, torch.randn(1, generator=gen)
    pi = torch.randint(0, P, (n,), generator=gen).sort().values
    li = torch.randint(0, L, (n,), generator=gen)
    deg = torch.full((P,), 9, dtype=torch.int64)                     # one head serves every pair
    perm = torch.randperm(n, generator=gen)
    dpred = torch.randn(n, generator=gen).to(dev)
    head = ops.Head(*[t.to(dev) for t in (A, B, W2, b2, W3, b3)])
    i32 = lambda t: t.to(torch.int32).to(dev)
    pi_d, li_d, deg_d, io = i32(pi), i32(li), i32(deg), perm.to(dev)
    good = torch.arange(0, n, 3, dtype=torch.int32)                  # the clean list
    n_good = good.numel()
    zeros = lambda: ops.Head(*[torch.zeros_like(t, device=dev) for t in (A, B, W2, b2, W3, b3)])

    def run(sel, count, pi_=pi_d, li_=li_d, io_=io):
        pred = torch.full((n,), -3.0, device=dev)
        cnt = torch.tensor([count], dtype=torch.int32, device=dev)
        ops.pair_head_fwd(head, pi_, li_, deg_d, 6, False, p, 21, io_, pred, sel=sel.to(dev), n_sel=cnt, n_bound=sel.numel(),
                          io_perm=io_)
        g = zeros()
        ops.pair_head_bwd(head, g, pi_, li_, deg_d, 6, False, L, p, 21, io_, dpred, sel=sel.to(dev), n_sel=cnt,
                          n_bound=sel.numel(), io_perm=io_)
        torch.cuda.synchronize()
        return pred, g

    ref_pred, ref_g = run(good, n_good)
    assert float((ref_pred != -3.0).sum()) == n_good
    # (1) garbage entries inside the list and a stale count that runs past it into more garbage
    bad = torch.cat([good, torch.tensor([-1, -2 ** 31, n, n + 5, 2 ** 30, 2 ** 31 - 1], dtype=torch.int32),
                     torch.full((58,), 2 ** 29, dtype=torch.int32)])
    pred, g = run(bad, bad.numel())
    assert torch.equal(pred, ref_pred)
    for name in "A B W2 b2 W3 b3".split():
        assert torch.equal(getattr(g, name), getattr(ref_g, name)), name
    # (2) a count larger than the launch bound cannot reach beyond the list either
    pred, g = run(good, n_good + 10 ** 6)
    assert torch.equal(pred, ref_pred) and torch.equal(g.A, ref_g.A) and torch.equal(g.B, ref_g.B)
    # (3) patient ids outside A / deg are not pairs; lab ids outside B never index past the table and add nothing to dB
    pi_bad = pi_d.clone()
    hit = good[::7].long().to(dev)
    pi_bad[hit[::2]] = P + 17
    pi_bad[hit[1::2]] = -5
    pred, g = run(good, n_good, pi_=pi_bad)
    keep = torch.ones(n, dtype=torch.bool, device=dev)
    keep[perm.to(dev)[hit]] = False                                  # their output slots stay untouched
    assert torch.equal(pred[keep], ref_pred[keep]) and bool((pred[~keep] == -3.0).all())
    assert bool(torch.isfinite(g.A).all()) and bool(torch.isfinite(g.B).all())
    li_bad = li_d.clone()
    li_bad[hit] = L + 3
    pred, g = run(good, n_good, li_=li_bad)
    assert bool(torch.isfinite(pred).all()) and torch.equal(pred[keep], ref_pred[keep])
    assert bool(torch.isfinite(g.B).all())
    # (4) an output slot outside pred is dropped, not written
    io_bad = io.clone()
    io_bad[hit] = n + 1000
    pred, g = run(good, n_good, io_=io_bad)
    assert torch.equal(pred[keep], ref_pred[keep]) and bool((pred[~keep] == -3.0).all())
    # the wrapper refuses per-pair arrays of different lengths (the kernels range-check against pi's length)
    with pytest.raises(ValueError, match="entries"):
        ops.pair_head_fwd(head, pi_d, li_d[:-1], deg_d, 6, False, p, 21, io, ref_pred)


@pytest.mark.parametrize("loss_type", ["mae", "mse", "huber"])
def test_weighted_pair_loss(ops, dev, loss_type):
    gen = torch.Generator().manual_seed(41)
    n = 100003
    pred, y = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    y[:10] = pred[:10]                      # |0| has subgradient 0 (torch.abs convention)
    w, sup = torch.rand(n, generator=gen) + 0.5, (torch.rand(n, generator=gen) < 0.2).float()
    inv = 1.0 / float(sup.sum())
    p = pred.to(dev).requires_grad_(True)
    loss = ops.weighted_pair_loss(p, y.to(dev), w.to(dev), sup.to(dev), inv, loss_type)
    loss.backward()
    pd = pred.double().requires_grad_(True)
    d = pd - y.double()
    per = {"mae": d.abs(), "mse": d * d, "huber": torch.where(d.abs() <= 1, 0.5 * d * d, d.abs() - 0.5)}[loss_type]
    ref = (per * w.double() * sup.double()).sum() * inv
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
    assert rel(p.grad, pd.grad) <= 1e-6
    l2 = ops.weighted_pair_loss(p, y.to(dev), None, None, 1.0 / n, loss_type)
    r2 = per.mean()
    assert abs(float(l2) - float(r2)) <= 1e-6 * abs(float(r2))


def test_c_abi_error_behaviour(ops, dev):
    """Bad arguments come back as an error code + message (MmgError), never as a crash or a silent fallback."""
    import ctypes as C
    from mmgnn import _lib
    lib = _lib.load()
    x = torch.zeros(8, 96, device=dev)
    with pytest.raises(_lib.MmgError, match="K=96"):                       # unsupported inner dimension
        ops.linear_fwd(x, torch.zeros(64, 96, device=dev))
    with pytest.raises(_lib.MmgError, match="multiple of 64"):             # unsupported output width
        ops.linear_fwd(torch.zeros(8, 64, device=dev), torch.zeros(48, 64, device=dev))
    with pytest.raises(ValueError):                                         # shape mismatch caught by the wrapper
        ops.linear_fwd(torch.zeros(8, 64, device=dev), torch.zeros(64, 128, device=dev))
    with pytest.raises(ValueError, match="contiguous"):
        ops.linear_fwd(torch.zeros(64, 8, device=dev).t(), torch.zeros(64, 64, device=dev))
    with pytest.raises(TypeError):
        ops.linear_fwd(torch.zeros(8, 64, device=dev, dtype=torch.float64), torch.zeros(64, 64, device=dev))
    # a workspace that is too small is refused with MMG_E_WS, not overrun
    dy, xx, dW = torch.zeros(4096, 128, device=dev), torch.zeros(4096, 128, device=dev), torch.zeros(128, 128, device=dev)
    ws = torch.zeros(256, dtype=torch.uint8, device=dev)
    rc = lib.mmg_linear_wgrad(C.c_void_p(dy.data_ptr()), C.c_void_p(xx.data_ptr()), None, C.c_void_p(dW.data_ptr()), None,
                              4096, 128, 128, 0, C.c_void_p(ws.data_ptr()), ws.numel(), None)
    assert rc != 0 and b"workspace" in lib.mmg_last_error()
    # null buffers
    rc = lib.mmg_linear_fwd(None, None, C.c_void_p(dW.data_ptr()), None, C.c_void_p(dy.data_ptr()), 16, 128, 128, 0, None)
    assert rc != 0 and b"null" in lib.mmg_last_error()
    # empty problems are fine
    assert ops.linear_fwd(torch.zeros(0, 64, device=dev), torch.zeros(64, 64, device=dev)).shape == (0, 64)
    torch.cuda.synchronize()


def test_ops_reject_cpu_tensors(ops):
    with pytest.raises(Exception):
        ops.linear_fwd(torch.zeros(4, 64), torch.zeros(64, 64))


def test_small_fwd_and_wgrad_groups(dev):
    """Grouped launches of the vocab-side dense ops (mmg_small_fwd_group / mmg_small_wgrad_group): every problem of a
    group against an fp64 reference -- one and two terms, bias, accumulate, W stored [K,N], ragged row counts (1 .. 300
    and an empty table), more problems than one launch holds."""
    import mmgnn  # noqa: F401
    from mmgnn import ops
    gen = torch.Generator().manual_seed(11)
    for K, N in ((128, 128), (64, 64), (256, 64), (128, 64)):
        Ms = [50, 114, 100, 1, 33, 300, 200, 64, 7, 0]
        probs, refs = [], []
        for i, M in enumerate(Ms):
            x = torch.randn(M, K, generator=gen).to(dev)
            wkn = i % 3 == 1
            W = (torch.randn(K, N, generator=gen) if wkn else torch.randn(N, K, generator=gen)).to(dev) / K ** 0.5
            two = i % 2 == 0
            x2 = torch.randn(M, K, generator=gen).to(dev) if two else None
            W2 = (torch.randn(*W.shape, generator=gen).to(dev) / K ** 0.5) if two else None
            bias = torch.randn(N, generator=gen).to(dev) if i % 4 != 3 else None
            acc = i % 5 == 2
            out = torch.randn(M, N, generator=gen).to(dev) if acc else None
            ref = x.double() @ (W.double() if wkn else W.double().t())
            if two:
                ref = ref + x2.double() @ (W2.double() if wkn else W2.double().t())
            if bias is not None:
                ref = ref + bias.double()
            if acc:
                ref = ref + out.double()
            probs.append(ops.SmallFwd(x, W, out=out, bias=bias, x2=x2, W2=W2, accumulate=acc, w_kn=wkn))
            refs.append(ref)
        outs = ops.small_fwd_group(probs)
        for o, r, M in zip(outs, refs, Ms):
            assert o.shape == (M, N)
            if M:
                assert float((o.double() - r).abs().max()) <= 2e-6 * float(r.abs().max()), (K, N, M)
        wp, wrefs = [], []
        for i, M in enumerate(Ms):
            dy = torch.randn(M, N, generator=gen).to(dev)
            x = torch.randn(M, K, generator=gen).to(dev)
            acc = i % 3 == 0
            dW0 = torch.randn(N, K, generator=gen).to(dev) if acc else None
            db0 = torch.randn(N, generator=gen).to(dev) if acc else None
            wb = i % 2 == 0
            rW = dy.double().t() @ x.double() + (dW0.double() if acc else 0)
            rb = dy.double().sum(0) + (db0.double() if acc else 0)
            wp.append(ops.SmallWgrad(dy, x, dW=dW0.clone() if acc else None, dbias=db0.clone() if (acc and wb) else None,
                                     with_bias=wb, accumulate=acc))
            wrefs.append((rW, rb, wb))
        res = ops.small_wgrad_group(wp)
        for (dW, db), (rW, rb, wb), M in zip(res, wrefs, Ms):
            assert float((dW.double() - rW).abs().max()) <= 2e-6 * max(float(rW.abs().max()), 1e-30) + (0 if M else 0), (K, N, M)
            if wb:
                assert float((db.double() - rb).abs().max()) <= 2e-6 * max(float(rb.abs().max()), 1e-30), (K, N, M)
            else:
                assert db is None


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act,p", [(1, 0.0), (1, 0.3), (3, 0.2), (2, 0.0)])
def test_small_bn_groups_match_the_separate_kernels(dev, training, act, p):
    """mmg_small_bn_act_group / mmg_small_bn_bwd_group against the kernels they fuse (column statistics, mmg_bn_finalize,
    mmg_affine_act_drop; mmg_bn_bwd_stats, mmg_bn_bwd_apply): same outputs, same dropout masks, same running statistics,
    same folded vectors -- for several small node types in one launch, with and without BatchNorm."""
    import mmgnn  # noqa: F401
    from mmgnn import ops
    gen = torch.Generator().manual_seed(17)
    N, seed = 128, 9001
    Ms = [50, 114, 100, 2, 333]
    items, refs = [], []
    for i, M in enumerate(Ms):
        y = (torch.randn(M, N, generator=gen) * 2 + 0.3).to(dev)
        use_bn = i != 3
        mod = torch.nn.BatchNorm1d(N).to(dev) if use_bn else None
        if use_bn:
            with torch.no_grad():
                mod.weight.copy_(torch.rand(N, generator=gen) + 0.5); mod.bias.copy_(torch.randn(N, generator=gen) * 0.1)
                mod.running_mean.copy_(torch.randn(N, generator=gen) * 0.1); mod.running_var.copy_(torch.rand(N, generator=gen) + 0.5)
        site = 16 + i
        # reference: the separate kernels on a copy of the module
        if use_bn:
            ref_mod = torch.nn.BatchNorm1d(N).to(dev)
            ref_mod.load_state_dict(mod.state_dict())
            sums = ops.col_reduce2(y) if training else None
            fold = ops.bn_finalize(sums, M, ref_mod.weight.detach(), ref_mod.bias.detach(), ref_mod.running_mean,
                                   ref_mod.running_var, training, 1)
            pro_r = ops.Pro(fold.scale, fold.shift, act, p, seed, site, 0)
        else:
            ref_mod, fold = None, None
            pro_r = ops.Pro(None, None, act, p, seed, site, 0)
        out_r = ops.affine_act_drop(y, pro_r)
        g = torch.randn(M, N, generator=gen).to(dev)
        if fold is not None and training:
            bs = ops.bn_bwd_stats(g, y, pro_r, fold)
            dbg = torch.empty(2, N, device=dev)
            dy_r = ops.bn_bwd_apply(g, y, pro_r, fold, bs, M, dbg[0], dbg[1])
        elif fold is not None:
            bs = ops.bn_bwd_stats(g, y, pro_r, fold)
            dbg = bs.float()
            dy_r = ops.bn_bwd_apply(g, y, pro_r, fold)
        else:
            dbg = None
            dy_r = ops.bn_bwd_apply(g, y, pro_r, None)
        items.append((y, mod, ops.Pro(None, None, act, p, seed, site, 0)))
        refs.append((out_r, fold, ref_mod, g, dy_r, dbg))
    res = ops.small_bn_act_group(items, training)
    bitems = []
    for (y, mod, pro), (out, fold), (out_r, fold_r, ref_mod, g, dy_r, dbg) in zip(items, res, refs):
        assert torch.equal(out == 0, out_r == 0)                       # the same dropout / ReLU pattern
        assert rel(out, out_r) <= 2e-6
        if mod is not None:
            assert rel(fold.scale, fold_r.scale) <= 1e-6 and rel(fold.mean, fold_r.mean) <= 1e-6 + 1e-6
            assert rel(mod.running_mean, ref_mod.running_mean) <= 1e-6 and rel(mod.running_var, ref_mod.running_var) <= 1e-6
        bitems.append((g, y, pro, fold))
    for (dy, dbeta, dgamma), (out_r, fold_r, ref_mod, g, dy_r, dbg) in zip(ops.small_bn_bwd_group(bitems), refs):
        assert rel(dy, dy_r) <= 5e-6
        if dbg is not None:
            assert rel(dbeta, dbg[0]) <= 2e-6 and rel(dgamma, dbg[1]) <= 2e-6
        else:
            assert dbeta is None and dgamma is None


def test_deferred_weight_gradient_sums(ops, dev):
    """linear_wgrad(defer=list) + wgrad_reduce_flush == the immediate form, bit for bit: several layers in one launch, an
    accumulating second contribution to the same gradient (a later launch), a small direct launch behind a deferred one,
    bias gradients riding along."""
    gen = torch.Generator().manual_seed(4242)

    def mk(M, N, K):
        return torch.randn(M, N, generator=gen).to(dev), torch.randn(M, K, generator=gen).to(dev)

    a, b, c, d = mk(5000, 128, 128), mk(3000, 64, 128), mk(4200, 128, 128), mk(100, 128, 128)
    ref1, refb1 = ops.linear_wgrad(a[0], a[1], with_bias=True)
    ref2 = ops.linear_wgrad(b[0], b[1])
    ops.linear_wgrad(c[0], c[1], out=ref1, accumulate=True, with_bias=True, bias_out=refb1)
    ops.linear_wgrad(d[0], d[1], out=ref1, accumulate=True, with_bias=True, bias_out=refb1)
    jobs = []
    g1, gb1 = ops.linear_wgrad(a[0], a[1], with_bias=True, defer=jobs)
    g2 = ops.linear_wgrad(b[0], b[1], defer=jobs)
    ops.linear_wgrad(c[0], c[1], out=g1, accumulate=True, with_bias=True, bias_out=gb1, defer=jobs)
    assert len(jobs) == 3
    ops.linear_wgrad(d[0], d[1], out=g1, accumulate=True, with_bias=True, bias_out=gb1, defer=jobs)   # direct: flushes first
    ops.wgrad_reduce_flush(jobs)
    assert not jobs
    assert torch.equal(g1, ref1) and torch.equal(gb1, refb1) and torch.equal(g2, ref2)


@pytest.mark.parametrize("M", [5003, 300])
def test_batchnorm_fold_in_the_statistics_launch(ops, dev, M):
    """linear_fwd(bn=...) == linear_fwd(with_stats) + bn_finalize, bit for bit (fold, running statistics, sums), on the
    epilogue-statistics path (M > 512) and on the separate-reduction path (small M)."""
    gen = torch.Generator().manual_seed(31 + M)
    K = N = 128
    x = torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    gamma, beta = (torch.rand(N, generator=gen) + 0.5).to(dev), torch.randn(N, generator=gen).to(dev)
    rm0, rv0 = torch.randn(N, generator=gen).to(dev), (torch.rand(N, generator=gen) + 0.5).to(dev)
    rm1, rv1 = rm0.clone(), rv0.clone()
    y0, s0 = ops.linear_fwd(x, W, b, with_stats=True)
    f0 = ops.bn_finalize(s0, M, gamma, beta, rm0, rv0, True, 2)
    y1, s1, f1 = ops.linear_fwd(x, W, b, bn=(gamma, beta, rm1, rv1, 2))
    assert torch.equal(y0, y1) and torch.equal(s0, s1)
    for a, c in ((f0.scale, f1.scale), (f0.shift, f1.shift), (f0.mean, f1.mean), (f0.rstd, f1.rstd), (rm0, rm1), (rv0, rv1)):
        assert torch.equal(a, c)
    assert f1.count == M and f1.training


# ------------------------------------------------------------------------------------------ next-BatchNorm statistics
def _bn_below(ops, dev, gen, M, N, p, relu=True, row_offset=10, site=23):
    """A BatchNorm + activation + dropout 'below' a producer: (y, pro, fold)."""
    y = (torch.randn(M, N, generator=gen) * 1.5 + 0.2).to(dev)
    gamma, beta = (torch.rand(N, generator=gen) + 0.5).to(dev), (torch.randn(N, generator=gen) * 0.2).to(dev)
    fold = ops.bn_finalize(ops.col_reduce2(y), M, gamma, beta, None, None, True)
    return y, ops.Pro(fold.scale, fold.shift, relu, p, seed=11, site=site, row_offset=row_offset), fold


def _stats_close(ops, got, out, y, pro, fold, bar=1e-6):
    want = ops.bn_bwd_stats(out, y, pro, fold)          # the separate pass (fp64 per element) over the finished output
    assert rel(got[0], want[0]) <= bar and rel(got[1], want[1]) <= bar, (rel(got[0], want[0]), rel(got[1], want[1]))


@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("M,N,K,relu", [(5003, 128, 64, 1), (5003, 128, 128, 1), (700, 256, 64, 1), (2049, 128, 64, 0),
                                         (3000, 64, 64, 1), (300, 128, 64, 1), (3000, 128, 64, 2)])
def test_next_bn_statistics_from_the_linear_epilogue(ops, dev, M, N, K, relu, p):
    """mmg_linear_fwd_next_bn: the data-gradient GEMM also returns mmg_bn_bwd_stats of its own output (tail tile, two column
    slices, no activation; N = 64, M <= 512 and leaky_relu take the separate pass inside the call)."""
    gen = torch.Generator().manual_seed(M + N + K)
    dy = torch.randn(M, K, generator=gen).to(dev)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    y, pro, fold = _bn_below(ops, dev, gen, M, N, p, relu)
    ref = ops.linear_fwd(dy, W, w_kn=True)
    out, sums = ops.linear_fwd(dy, W, w_kn=True, next_bn=ops.NextBN(y, pro, fold))
    assert torch.equal(out, ref)
    _stats_close(ops, sums, out, y, pro, fold)
    again = ops.linear_fwd(dy, W, w_kn=True, next_bn=ops.NextBN(y, pro, fold))[1]
    assert torch.equal(again, sums)                                      # fixed summation order
    acc = sums.clone()                                                   # sums given: the statistics are ADDED
    ops.linear_fwd(dy, W, w_kn=True, next_bn=ops.NextBN(y, pro, fold, sums=acc))
    assert rel(acc, 2 * sums) <= 1e-12


@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("M,K,N", [(5003, 128, 128), (3000, 64, 128)])
def test_next_bn_statistics_from_the_l2_and_bn_backward_gemms(ops, dev, M, K, N, p):
    """mmg_linear_l2bwd_next_bn / mmg_linear_bnbwd_next_bn / mmg_linear_bnbwd_rows_next_bn: dz and dx unchanged, the statistics
    of the BatchNorm that consumes dx from the epilogue; two producers ADD into one set of sums (== mmg_bn_bwd_stats2)."""
    gen = torch.Generator().manual_seed(5 * M + K)
    W = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    yb, pro_b, fold_b = _bn_below(ops, dev, gen, M, N, p)
    # L2 backward
    z = torch.randn(M, K, generator=gen).to(dev)
    outn, rn = ops.l2norm_fwd(z)
    g = torch.randn(M, K, generator=gen).to(dev)
    dz0, dx0 = ops.linear_l2bwd(g, outn, rn, W)
    dz1, dx1, s1 = ops.linear_l2bwd(g, outn, rn, W, next_bn=ops.NextBN(yb, pro_b, fold_b))
    assert torch.equal(dz0, dz1) and torch.equal(dx0, dx1)
    _stats_close(ops, s1, dx1, yb, pro_b, fold_b)
    # BatchNorm backward GEMM, dense, then the row-list form adding its share with its own dropout mask
    y, pro, fold = _bn_below(ops, dev, gen, M, K, p, site=31)
    sums = ops.bn_bwd_stats(g, y, pro, fold)
    d0, d1 = torch.zeros(2, K, device=dev), torch.zeros(2, K, device=dev)
    dz0, dx0 = ops.linear_bnbwd(g, y, pro, fold, W, sums, M, d0[0], d0[1])
    dz1, dx1, sa = ops.linear_bnbwd(g, y, pro, fold, W, sums, M, d1[0], d1[1], next_bn=ops.NextBN(yb, pro_b, fold_b))
    assert torch.equal(dz0, dz1) and torch.equal(dx0, dx1) and torch.equal(d0, d1)
    _stats_close(ops, sa, dx1, yb, pro_b, fold_b)
    if K == 128:
        rows = torch.randperm(M, generator=gen)[:157].sort().values.to(dev)
        g_rows = torch.randn(rows.numel(), K, generator=gen).to(dev)
        row_pos = torch.full((M,), -1, dtype=torch.int32, device=dev)
        row_pos[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=dev)
        sr = ops.bn_bwd_stats_rows(g_rows, y, rows, pro, fold)
        dzr0, dxr0 = ops.linear_bnbwd_rows(g_rows, row_pos, y, pro, fold, W, sr, M, d0[0], d0[1])
        pro_b2 = ops.Pro(pro_b.scale, pro_b.shift, True, p, seed=11, site=pro_b.site + 1, row_offset=10)   # the other pass's mask
        both = sa.clone()
        dzr1, dxr1, got = ops.linear_bnbwd_rows(g_rows, row_pos, y, pro, fold, W, sr, M, d1[0], d1[1],
                                                next_bn=ops.NextBN(yb, pro_b2, fold_b, sums=both))
        assert got is both and torch.equal(dzr0, dzr1) and torch.equal(dxr0, dxr1)
        want = ops.bn_bwd_stats2(dx1, dxr1, yb, pro_b, pro_b2, fold_b)
        assert rel(both[0], want[0]) <= 1e-6 and rel(both[1], want[1]) <= 1e-6


@pytest.mark.parametrize("p", [0.0, 0.3])
@pytest.mark.parametrize("D,n_rows", [(128, 5000), (256, 1834), (64, 1834)])
def test_next_bn_statistics_from_the_gather_epilogue(ops, dev, D, n_rows, p):
    """mmg_gather_rows_next_bn: one relation (the last layer's backward) and three, accumulate or not; D = 64 takes the
    separate pass inside the call."""
    gen = torch.Generator().manual_seed(3 * D + n_rows)
    grels = []
    for nc, md in zip([50, 114, 100], [50, 9, 25]):
        ei = simple_edges(gen, n_rows, nc, md)
        rp, col = _csr(ops, dev, ei, n_rows)
        _, inv = ops.row_degree(rp)
        _, cinv = ops.col_degree(col, nc)
        _, mask_r = ops.rel_mask_build(rp, col, nc)
        grels.append(ops.Rel(rp, col, nc, rowscale=inv, colscale=cinv, table=(torch.randn(nc, D, generator=gen) * 2).to(dev),
                             simple=True, mask_r=mask_r))
    y, pro, fold = _bn_below(ops, dev, gen, n_rows, D, p)
    base = torch.randn(n_rows, D, generator=gen).to(dev)
    for rels in (grels[:1], grels):
        for acc in (True, False):
            ref = base.clone()
            ops.gather_rows(rels, n_rows, D, ref, accumulate=acc)
            o = base.clone()
            _, sums = ops.gather_rows(rels, n_rows, D, o, accumulate=acc, next_bn=ops.NextBN(y, pro, fold))
            assert torch.equal(o, ref)
            _stats_close(ops, sums, o, y, pro, fold)


@pytest.mark.parametrize("n", [1, 3, 63, 65, 5829, 100003])
def test_zero_fill_recorded_into_a_hipgraph_zeroes_at_every_replay(ops, dev, n):
    """ops.zeros inside a captured step: the buffer is zero at EVERY replay, whatever it held before.  (hipMemsetAsync recorded
    into a hipGraph replays a stale fill pattern on this ROCm from the second replay on -- profiles/probes/hipgraph_memset_node.py;
    the library fills with a kernel.)  Also unaligned views: a zero-fill of bytes 4 .. 4 + 4 n of a buffer touches nothing else."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out = torch.empty(n, device=dev)
        g = torch.cuda.CUDAGraph()
        g.capture_begin()
        z = ops.zeros(n, device=dev)
        out.copy_(z)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    for rep in range(5):
        z.fill_(float("inf") if rep % 2 else 7.0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert int((out != 0).sum()) == 0, (rep, out[out != 0][:4].tolist())
    del g
    from mmgnn import _lib
    buf = torch.full((n + 8,), 5.0, device=dev)
    _lib.check(_lib.load().mmg_fill_zero(ops._p(buf[1:1 + n], torch.float32), n * 4, ops._stream()), "mmg_fill_zero")
    torch.cuda.synchronize()
    assert float(buf[0]) == 5.0 and bool((buf[1 + n:] == 5.0).all()) and int((buf[1:1 + n] != 0).sum()) == 0
