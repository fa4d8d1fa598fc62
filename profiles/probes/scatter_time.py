"""Stand-alone timing of mmg_scatter_rows (kernel + slab sum) on the x100 eICU-shape graph, with and without a rowscale,
next to an fp64 index_add_ check of the same call.  Usage: python profiles/probes/scatter_time.py [scale] [D] [iters] [eicu|mimic]"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import mmgnn  # noqa: F401
from mmgnn import _lib
if os.environ.get("MMG_AB_LIB"):                       # an ablation build (profiles/probes/scatter_ab.sh)
    _lib.LIB_PATH = os.path.join(REPO, os.environ["MMG_AB_LIB"])
from mmgnn import ops
from mmgnn.data import build_plan
from mmgnn.synth import make_graph, EICU, MIMIC_LIKE

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 100
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
shape = MIMIC_LIKE if (len(sys.argv) > 4 and sys.argv[4] == "mimic") else EICU
g = make_graph(scale, seed=0, device=dev, shape=shape)
plan = build_plan(g, dev)
P = plan.n_rows
x = torch.randn(P, D, device=dev) * 1.5 + 0.25
rout = plan.rels_from_patient()
n_items = sum(r.n_cols for r in rout)
edges = sum(int(r.col.numel()) for r in rout)
alg_bytes = P * D * 4 + edges * 4 + n_items * D * 4


def rels_for(rowscale):
    buf = torch.empty(n_items, D, device=dev)
    rels, off = [], 0
    for r in rout:
        o = buf[off:off + r.n_cols]; off += r.n_cols
        rels.append(ops.Rel(r.rowptr, r.col, r.n_cols, colscale=None if rowscale else r.inv_col, out=o, simple=r.simple,
                            mask_t=r.mask_t, rowscale=r.inv_row if rowscale else None))
    return rels, buf


def reference(rowscale):
    outs = []
    for r in rout:
        rows = torch.repeat_interleave(torch.arange(P, device=dev), (r.rowptr[1:] - r.rowptr[:-1]).long())
        xs = x.double()
        if rowscale:
            xs = xs * r.inv_row.double()[:, None]
        ref = torch.zeros(r.n_cols, D, dtype=torch.float64, device=dev).index_add_(0, r.col.long(), xs[rows])
        if not rowscale:
            ref = ref * r.inv_col.double()[:, None]
        outs.append(ref)
    return torch.cat(outs)


for rowscale in (False, True):
    rels, buf = rels_for(rowscale)
    ops.scatter_rows(rels, P, D, x)
    ref = reference(rowscale)
    err = float((buf.double() - ref).abs().max() / ref.abs().max())
    first = buf.clone()
    for _ in range(20):
        ops.scatter_rows(rels, P, D, x)
    same = bool(torch.equal(first, buf))
    torch.cuda.synchronize()
    # 20 calls per captured graph: the host (ctypes marshalling, ~25 us per call) must not pace the GPU
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(20):
                ops.scatter_rows(rels, P, D, x)
    graph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(max(iters // 20, 1)):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (max(iters // 20, 1) * 20)
    print(os.environ.get("MMG_AB_LIB", "shipped"), f"scale {scale} D {D} rowscale {int(rowscale)}: {us:7.2f} us per call (kernel + slab sum), "
          f"{alg_bytes / 1e6:.1f} MB algorithmic -> {alg_bytes / us / 1e6:.2f} TB/s = {alg_bytes / us / 8e6:.3f} of 8 TB/s; "
          f"max err vs fp64 {err:.2e}; repeat bit-identical {same}", flush=True)
