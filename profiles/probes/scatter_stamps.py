"""Phase stamps of k_scatter_strip (s_memtime per wave at: kernel entry, LUT barrier, first operands ready, end of the main
loop, end of the slab store).  Needs a DIAGNOSTIC build of the library with -DMMG_STAMPS (never the shipped one):

    cd multi-modal-gnn_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DMMG_STAMPS -c aggregate.hip -o /tmp/aggregate_stamps.o \\
      && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../profiles/probes/libmmgnn_stamps.so api.o csr.o /tmp/aggregate_stamps.o gemm.o elementwise.o pairs.o evalred.o optim.o small.o

Output of the round-2 run: scatter_stamps_mi355x.log."""
import os, sys, ctypes, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import mmgnn
from mmgnn import _lib
_lib.LIB_PATH = os.path.join(REPO, "profiles", "probes", "libmmgnn_stamps.so")
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
for _ in range(30):
    ops.scatter_rows(rels, P, D, x)
torch.cuda.synchronize()
lib = _lib.load()
n = 1024 * 8
host = np.zeros(n, np.uint64)
lib.mmg_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.mmg_debug_stamps(host.ctypes.data, n)
st = host.reshape(1024, 8).astype(np.int64)
t0 = st[:, 0].min()
names = ["start->lut barrier", "prologue (masks, ring prime, first operands)", "main loop", "LDS sum + stores"]
for i, nm in enumerate(names):
    d = st[:, i + 1] - st[:, i]
    print(f"{nm:48s} median {np.median(d):9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f} ticks")
tot = st[:, 4] - st[:, 0]
print("wave total median", np.median(tot), "max", tot.max(), "; kernel span (first start -> last end)", st[:, 4].max() - t0, "; start skew p90", np.percentile(st[:,0]-t0, 90))
print("(s_memtime tick rate not calibrated here: only the RATIOS of the phases are used)")
