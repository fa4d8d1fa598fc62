"""Phase stamps of k_scatter_roles' multiplier waves: s_memtime (shader clock) and s_memrealtime (100 MHz) at kernel entry,
before / after the main loop and after the cross-wave sum -> cycles per phase AND the clock the chip held meanwhile.
Needs an ablation build with -DMMG_STAMPS:  MMG_AB_LIB=ab/libmmgnn_<n>s.so python profiles/probes/scatter_roles_stamps.py"""
import os, sys, ctypes, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import mmgnn  # noqa: F401
from mmgnn import _lib
_lib.LIB_PATH = os.path.join(REPO, os.environ["MMG_AB_LIB"])
from mmgnn import ops
from mmgnn.data import build_plan
from mmgnn.synth import make_graph
dev = torch.device("cuda:0")
g = make_graph(100, seed=0, device=dev); plan = build_plan(g, dev); P = plan.n_rows; D = 128
x = torch.randn(P, D, device=dev)
rout = plan.rels_from_patient()
buf = torch.empty(sum(r.n_cols for r in rout), D, device=dev)
rels, off = [], 0
for r in rout:
    o = buf[off:off + r.n_cols]; off += r.n_cols
    rels.append(ops.Rel(r.rowptr, r.col, r.n_cols, colscale=r.inv_col, out=o, simple=r.simple, mask_t=r.mask_t))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        for _ in range(20):
            ops.scatter_rows(rels, P, D, x)
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
lib = _lib.load()
n = 1024 * 8
host = np.zeros(n, np.uint64)
lib.mmg_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mmg_debug_stamps(host.ctypes.data, n)
st = host.reshape(1024, 8).astype(np.int64)
names = ["entry -> loop", "main loop", "cross-wave sum + stores"]
for i, nm in enumerate(names):
    cyc = st[:, i + 1] - st[:, i]
    rt = (st[:, 4 + i + 1] - st[:, 4 + i]) * 10.0        # ns at 100 MHz
    ok = rt > 0
    print(f"{os.environ['MMG_AB_LIB']:24s} {nm:26s} median {np.median(cyc):9.0f} cycles = {np.median(rt) / 1e3:6.2f} us"
          f" -> {np.median(cyc[ok] / rt[ok]):5.2f} GHz")
