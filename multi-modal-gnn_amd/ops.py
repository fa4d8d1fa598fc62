"""Typed Python entry points over the C ABI (include/mmgnn.h): one function per exported op.

PyTorch is plumbing here -- it owns device memory and the stream; every computation below is a
hand-written HIP kernel in libmmgnn.so.  All tensors must live on a HIP device, be contiguous and
fp32 (indices int32 unless stated); violations raise instead of silently copying.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import (BnFinT, HeadGradT, HeadT, NextBnT, PairSavedT, PrologueT, RelT, SmallBnBwdT, SmallBnT, SmallFwdT, SmallWgradT, SumJobT, WgradReduceT,
                   check)

BN_MOMENTUM = 0.1
BN_EPS = 1e-5
L2_EPS = 1e-12


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor], dtype=torch.float32, name="tensor"):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MmgError(f"{name}: expected a HIP device tensor, got {t.device} (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------------------
# optional per-op timing (bench.py): HIP events recorded on the stream the kernels are launched on
class OpProfiler:
    """Collects (op, ms, algorithmic bytes, flops) per call; `only` restricts it to some op names."""

    def __init__(self, only=None):
        self.only = set(only) if only else None
        self.rows = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, nbytes, flops in self.rows:
            d = out.setdefault(name, dict(calls=0, ms=0.0, bytes=0, flops=0, shapes={}))
            ms = e0.elapsed_time(e1)
            d["calls"] += 1
            d["ms"] += ms
            d["bytes"] += nbytes
            d["flops"] += flops
            # launches of one op with the same algorithmic size are one kernel shape
            sh = d["shapes"].setdefault((nbytes, flops), dict(calls=0, ms=0.0, bytes=nbytes, flops=flops))
            sh["calls"] += 1
            sh["ms"] += ms
        return out


_PROF: Optional[OpProfiler] = None


def set_profiler(p: Optional[OpProfiler]):
    global _PROF
    _PROF = p


def _pb(name):
    if _PROF is None or (_PROF.only is not None and name not in _PROF.only):
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _pe(tok, name, nbytes=0, flops=0):
    if tok is None:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    _PROF.rows.append((name, tok, e, int(nbytes), int(flops)))


PROBE_TAGS = {1: "linear_fwd", 2: "linear_wgrad", 3: "linear_wgrad_reduce", 4: "gather_rows", 5: "scatter_rows",
              6: "scatter_reduce", 7: "pair_head_fwd", 8: "pair_head_bwd", 9: "bn_bwd_stats", 10: "bn_bwd_apply",
              11: "elementwise"}


def probe_arm(n: int):
    """Measurement hook: the next n big-kernel launches (any host thread) carry their own HIP start / stop event pair."""
    check(_lib.load().mmg_probe_arm(int(n)), "mmg_probe_arm")


PROBE_NAME_LEN = 128


def probe_read(cap: int = 1 << 16):
    """-> list of (ms, family name, M, N, K, flags, kernel symbol) of the probed launches (flags: 1 accumulate, 4 prologue, 8 rowscale, 16 BatchNorm backward staged in the GEMM, 32 L2-norm epilogue, 64 L2-norm backward staged, 128 two upstream gradients, 256 row-list upstream gradient)."""
    import numpy as np
    ms = np.zeros(cap, np.float32); tag = np.zeros(cap, np.int32); M = np.zeros(cap, np.int64)
    N = np.zeros(cap, np.int32); K = np.zeros(cap, np.int32); fl = np.zeros(cap, np.int32)
    names = np.zeros(cap * PROBE_NAME_LEN, np.uint8)
    n = _lib.load().mmg_probe_read(ms.ctypes.data, tag.ctypes.data, M.ctypes.data, N.ctypes.data, K.ctypes.data,
                                   fl.ctypes.data, names.ctypes.data, cap)
    nm = names.reshape(cap, PROBE_NAME_LEN)
    return [(float(ms[i]), PROBE_TAGS.get(int(tag[i]), str(int(tag[i]))), int(M[i]), int(N[i]), int(K[i]), int(fl[i]),
             bytes(nm[i]).split(b"\0", 1)[0].decode("ascii", "replace")) for i in range(n)]


_WS = {}              # key -> [buffer, handed out during a stream capture?]
_WS_RETIRED = []      # outgrown workspaces a captured hipGraph may hold the address of: never handed back


def workspace(nbytes: int, device) -> torch.Tensor:
    """Growable scratch per (device, host thread, stream).  Reuse is stream-ordered: an op runs on the current stream,
    and neither two host threads nor two streams (the model overlaps its vocab-side work on a side stream) share a
    buffer.  Growth is geometric (at least twice the old size: a sweep over growing shapes retires O(log n) buffers), and
    an outgrown buffer is only kept alive if it was ever handed out DURING a stream capture -- a graph recorded then replays
    kernels that write to it; one that no capture has seen goes back to the allocator."""
    import threading
    d = torch.device(device)
    key = (d.index if d.index is not None else torch.cuda.current_device(), threading.get_ident(),
           torch.cuda.current_stream().cuda_stream)
    ent = _WS.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None or ent[0].numel() < nbytes:
        old = 0
        if ent is not None:
            old = ent[0].numel()
            if ent[1]:
                # a step captured earlier on this stream replays kernels that write to the old buffer: if it went back to
                # the allocator, the next tensor placed there (an index list, say) would be scribbled over by the replay
                _WS_RETIRED.append(ent[0])
        ent = [torch.empty(max(int(nbytes), 2 * old, 1 << 20), dtype=torch.uint8, device=device), False]
        _WS[key] = ent
    if capturing:
        ent[1] = True
    return ent[0]


@dataclass
class Pro:
    """Prologue dropout(act(x*scale+shift)) applied on load (mmg_prologue_t).  relu: False / True, or an MMG_ACT_* code
    (2 = leaky_relu, 3 = elu: the materialising and backward kernels only)."""
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    relu: int = 0
    p: float = 0.0
    seed: int = 0
    site: int = 0
    row_offset: int = 0
    seed_dev: Optional[torch.Tensor] = None     # int64 [1] on the device: overrides `seed` at run time (hipGraph replays)

    def c(self):
        return PrologueT(_p(self.scale, name="pro.scale"), _p(self.shift, name="pro.shift"), int(self.relu),
                         float(self.p), int(self.seed) & 0xFFFFFFFFFFFFFFFF, int(self.site), int(self.row_offset),
                         _p(self.seed_dev, torch.int64, "pro.seed_dev"))


def _pro(pro: Optional[Pro]):
    return C.byref(pro.c()) if pro is not None else None


# ------------------------------------------------------------------------------------------ CSR
def csr_build(edge_index: torch.Tensor, n_rows: int, sort_row: int):
    """-> (rowptr int32 [n_rows+1], col int32 [E], perm int32 [E]); see mmg_csr_build."""
    lib = _lib.load()
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be [2,E], got {tuple(edge_index.shape)}")
    E = int(edge_index.shape[1])
    dev = edge_index.device
    rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    col = torch.empty(E, dtype=torch.int32, device=dev)
    perm = torch.empty(E, dtype=torch.int32, device=dev)
    nb = lib.mmg_csr_build_ws_bytes(E, n_rows)
    ws = workspace(nb, dev)
    check(lib.mmg_csr_build(_p(edge_index, torch.int64, "edge_index"), E, n_rows, sort_row, _p(rowptr, torch.int32),
                            _p(col, torch.int32), _p(perm, torch.int32), _p(ws, torch.uint8), ws.numel(), _stream()),
          "mmg_csr_build")
    return rowptr, col, perm


def row_degree(rowptr: torch.Tensor):
    lib = _lib.load()
    n = rowptr.numel() - 1
    deg = torch.empty(n, dtype=torch.int32, device=rowptr.device)
    inv = torch.empty(n, dtype=torch.float32, device=rowptr.device)
    check(lib.mmg_row_degree(_p(rowptr, torch.int32), n, _p(deg, torch.int32), _p(inv), _stream()), "mmg_row_degree")
    return deg, inv


def col_degree(col: torch.Tensor, n_cols: int):
    lib = _lib.load()
    cnt = torch.empty(n_cols, dtype=torch.int32, device=col.device)
    inv = torch.empty(n_cols, dtype=torch.float32, device=col.device)
    check(lib.mmg_col_degree(_p(col, torch.int32), col.numel(), n_cols, _p(cnt, torch.int32), _p(inv), _stream()),
          "mmg_col_degree")
    return cnt, inv


# ------------------------------------------------------------------------------------- aggregates
@dataclass
class Rel:
    rowptr: torch.Tensor
    col: torch.Tensor
    n_cols: int
    rowscale: Optional[torch.Tensor] = None
    colscale: Optional[torch.Tensor] = None
    table: Optional[torch.Tensor] = None
    out: Optional[torch.Tensor] = None
    simple: bool = False          # no duplicate (row, col) pair (MMG_REL_SIMPLE)
    mask_t: Optional[torch.Tensor] = None     # bit planes of the adjacency (rel_mask_build), simple relations only
    mask_r: Optional[torch.Tensor] = None     # the same, row-major (gather side)


def _agg_bytes(rels, n_rows, D, accumulate):
    """Algorithmic (compulsory) bytes of one fused aggregate launch: every index once, every distinct
    feature row once (the patient tensor is counted ONCE for the fused relations)."""
    b = 4 * D * n_rows * (2 if accumulate else 1)
    for r in rels:
        b += 4 * r.col.numel() + 4 * (n_rows + 1) + 4 * D * r.n_cols
        if r.rowscale is not None:
            b += 4 * n_rows
        if r.colscale is not None:
            b += 4 * r.n_cols
    return b


def _rels(rels: Sequence[Rel], D: int, need_table=False, need_out=False):
    if not 1 <= len(rels) <= _lib.MMG_MAX_REL:
        raise ValueError(f"1..{_lib.MMG_MAX_REL} relations per launch")
    arr = (RelT * len(rels))()
    for i, r in enumerate(rels):
        if need_table and (r.table is None or tuple(r.table.shape) != (r.n_cols, D)):
            raise ValueError(f"relation {i}: table must be [{r.n_cols},{D}]")
        if need_out and (r.out is None or tuple(r.out.shape) != (r.n_cols, D)):
            raise ValueError(f"relation {i}: out must be [{r.n_cols},{D}]")
        arr[i] = RelT(_p(r.rowptr, torch.int32), _p(r.col, torch.int32), _p(r.rowscale), _p(r.colscale),
                      _p(r.table), _p(r.out), r.n_cols, 1 if r.simple else 0,
                      _p(r.mask_t, torch.int64) if r.simple else None,
                      _p(r.mask_r, torch.int64) if r.simple else None)
    return arr


def rel_mask_build(rowptr: torch.Tensor, col: torch.Tensor, n_cols: int):
    """Bit planes of a SIMPLE CSR relation (mmg_rel_mask_build) -> (mask_t, mask_r), int64 storage each:
    mask_t [ceil(n_rows/64)][pad32(n_cols)][2] words (scatter side), mask_r [n_rows][2][pad32(n_cols)/16] uint16
    fields (gather side)."""
    lib = _lib.load()
    n_rows = rowptr.numel() - 1
    words = lib.mmg_rel_mask_words(n_rows, n_cols)
    mask_t = torch.empty(max(words, 1), dtype=torch.int64, device=rowptr.device)
    mask_r = torch.empty(max(words, 1), dtype=torch.int64, device=rowptr.device)
    check(lib.mmg_rel_mask_build(_p(rowptr, torch.int32), _p(col, torch.int32), n_rows, n_cols, _p(mask_t, torch.int64),
                                 _p(mask_r, torch.int64), _stream()), "mmg_rel_mask_build")
    return mask_t, mask_r


def _bn_fin(count: int, N: int, device, bn):
    """bn = (gamma, beta, running_mean, running_var, n_updates) -> (mmg_bn_fin_t, the BNFold its launch fills)."""
    gamma, beta, rm, rv, n_updates = bn
    st = torch.empty(4, N, dtype=torch.float32, device=device)
    fin = BnFinT(int(count), _p(gamma).value, _p(beta).value, _p(rm).value if rm is not None else None,
                 _p(rv).value if rv is not None else None, int(n_updates), BN_MOMENTUM, BN_EPS,
                 _p(st[0]).value, _p(st[1]).value, _p(st[2]).value, _p(st[3]).value)
    return fin, BNFold(st[0], st[1], st[2], st[3], int(count), True)


def gather_rows(rels: Sequence[Rel], n_rows: int, D: int, out: torch.Tensor, accumulate: bool, with_stats: bool = False,
                bn=None, next_bn: Optional["NextBN"] = None):
    """with_stats: also return fp64 [2,D] = (column sums, column sums of squares) of the final `out`.
    bn = (gamma, beta, running_mean, running_var, n_updates): also fold the training-mode BatchNorm of `out` in the launch
    that sums the statistics (mmg_gather_rows_stats_bn) -> (out, sums, BNFold)."""
    lib = _lib.load()
    if tuple(out.shape) != (n_rows, D):
        raise ValueError("gather_rows: out shape")
    for r in rels:
        if r.rowptr.numel() != n_rows + 1:
            raise ValueError("gather_rows: rowptr length")
    arr = _rels(rels, D, need_table=True)
    _tok = _pb("gather_rows")
    fold = None
    if next_bn is not None:       # -> (out, the statistics of the BatchNorm backward that consumes out), see NextBN
        if bn is not None or with_stats:
            raise ValueError("gather_rows: next_bn excludes the forward statistics")
        nbt, sums = _next_bn(next_bn, n_rows, D)
        check(lib.mmg_gather_rows_next_bn(arr, len(rels), n_rows, D, _p(out), int(accumulate), C.byref(nbt), _stream()),
              "mmg_gather_rows_next_bn")
        _pe(_tok, "gather_rows", _agg_bytes(rels, n_rows, D, accumulate) + 4 * D * n_rows, 0)
        return out, sums
    if bn is not None:
        sums = torch.empty(2, D, dtype=torch.float64, device=out.device)
        ws = workspace(lib.mmg_gather_rows_stats_ws_bytes(n_rows, D), out.device)
        fin, fold = _bn_fin(n_rows, D, out.device, bn)
        check(lib.mmg_gather_rows_stats_bn(arr, len(rels), n_rows, D, _p(out), int(accumulate), _p(sums, torch.float64),
                                           _p(ws, torch.uint8), ws.numel(), C.byref(fin), _stream()), "mmg_gather_rows_stats_bn")
    elif with_stats:
        sums = torch.empty(2, D, dtype=torch.float64, device=out.device)
        ws = workspace(lib.mmg_gather_rows_stats_ws_bytes(n_rows, D), out.device)
        check(lib.mmg_gather_rows_stats(arr, len(rels), n_rows, D, _p(out), int(accumulate), _p(sums, torch.float64),
                                        _p(ws, torch.uint8), ws.numel(), _stream()), "mmg_gather_rows_stats")
    else:
        check(lib.mmg_gather_rows(arr, len(rels), n_rows, D, _p(out), int(accumulate), _stream()), "mmg_gather_rows")
    _pe(_tok, "gather_rows", _agg_bytes(rels, n_rows, D, accumulate), 0)
    if bn is not None:
        return out, sums, fold
    return (out, sums) if with_stats else out


def scatter_rows(rels: Sequence[Rel], n_rows: int, D: int, x: torch.Tensor):
    """Writes every rel.out ([n_cols, D])."""
    lib = _lib.load()
    if tuple(x.shape) != (n_rows, D):
        raise ValueError("scatter_rows: x shape")
    for r in rels:
        if r.rowptr.numel() != n_rows + 1:
            raise ValueError("scatter_rows: rowptr length")
    arr = _rels(rels, D, need_out=True)
    nb = lib.mmg_scatter_rows_ws_bytes(arr, len(rels), n_rows, D)
    ws = workspace(nb, x.device)
    _tok = _pb("scatter_rows")
    check(lib.mmg_scatter_rows(arr, len(rels), n_rows, D, _p(x), _p(ws, torch.uint8), ws.numel(), _stream()),
          "mmg_scatter_rows")
    _pe(_tok, "scatter_rows", _agg_bytes(rels, n_rows, D, False), 0)


# ------------------------------------------------------------------------------------------ dense
def linear_fwd(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, pro: Optional[Pro] = None,
               out: Optional[torch.Tensor] = None, accumulate: bool = False, w_kn: bool = False,
               with_stats: bool = False, bn=None, next_bn: Optional["NextBN"] = None):
    """out[M,N] (+)= pro(x)[M,K] @ W[N,K]^T + bias;  w_kn: W is stored [K,N] (out = x @ W), read in place.
    next_bn: -> (out, the statistics of the BatchNorm backward that consumes out), see NextBN.
    with_stats: also return fp64 [2,N] = (column sums, column sums of squares) of out, from the GEMM epilogue.
    bn = (gamma, beta, running_mean, running_var, n_updates): also fold the training-mode BatchNorm of `out` in the launch
    that sums the statistics (mmg_linear_fwd_stats_bn) -> (out, sums, BNFold)."""
    lib = _lib.load()
    M, K = x.shape
    N = W.shape[1] if w_kn else W.shape[0]
    if (W.shape[0] if w_kn else W.shape[1]) != K:
        raise ValueError(f"linear_fwd: W {tuple(W.shape)} vs x {tuple(x.shape)}")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs out")
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (M, N):
        raise ValueError("linear_fwd: out shape")
    _tok = _pb("linear_fwd")
    flags = int(accumulate) | (2 if w_kn else 0)
    fold = None
    if next_bn is not None:
        if bn is not None or with_stats:
            raise ValueError("linear_fwd: next_bn excludes the forward statistics")
        nbt, sums = _next_bn(next_bn, M, N)
        check(lib.mmg_linear_fwd_next_bn(_p(x, name="x"), _pro(pro), _p(W, name="W"), _p(bias, name="bias"),
                                         _p(out, name="out"), M, N, K, flags, C.byref(nbt), _stream()), "mmg_linear_fwd_next_bn")
        _pe(_tok, "linear_fwd", 4 * (M * K + N * K + M * N * (3 if accumulate else 2)), 2 * M * N * K)
        return out, sums
    if bn is not None:
        sums = torch.empty(2, N, dtype=torch.float64, device=x.device)
        ws = workspace(lib.mmg_linear_fwd_stats_ws_bytes(M, N), x.device)
        fin, fold = _bn_fin(M, N, x.device, bn)
        check(lib.mmg_linear_fwd_stats_bn(_p(x, name="x"), _pro(pro), _p(W, name="W"), _p(bias, name="bias"),
                                          _p(out, name="out"), M, N, K, flags, _p(sums, torch.float64), _p(ws, torch.uint8),
                                          ws.numel(), C.byref(fin), _stream()), "mmg_linear_fwd_stats_bn")
    elif with_stats:
        sums = torch.empty(2, N, dtype=torch.float64, device=x.device)
        ws = workspace(lib.mmg_linear_fwd_stats_ws_bytes(M, N), x.device)
        check(lib.mmg_linear_fwd_stats(_p(x, name="x"), _pro(pro), _p(W, name="W"), _p(bias, name="bias"),
                                       _p(out, name="out"), M, N, K, flags, _p(sums, torch.float64), _p(ws, torch.uint8),
                                       ws.numel(), _stream()), "mmg_linear_fwd_stats")
    else:
        check(lib.mmg_linear_fwd(_p(x, name="x"), _pro(pro), _p(W, name="W"), _p(bias, name="bias"), _p(out, name="out"),
                                 M, N, K, flags, _stream()), "mmg_linear_fwd")
    _pe(_tok, "linear_fwd", 4 * (M * K + N * K + M * N * (2 if accumulate else 1)), 2 * M * N * K)
    if bn is not None:
        return out, sums, fold
    return (out, sums) if with_stats else out


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, pro: Optional[Pro] = None, out: Optional[torch.Tensor] = None,
                 accumulate: bool = False, with_bias: bool = False, bias_out: Optional[torch.Tensor] = None,
                 defer: Optional[list] = None):
    """out[N,K] (+)= dy[M,N]^T @ pro(x)[M,K].  with_bias: also return the column sums of dy ([N] float, the bias
    gradient of the same layer), computed in the same pass over dy.
    defer (a list): the partial slabs are NOT summed now -- a job is appended to the list and `out` is complete only after
    wgrad_reduce_flush(list), which sums the slabs of many layers in one launch (nobody reads a weight gradient before the
    optimizer does)."""
    lib = _lib.load()
    M, N = dy.shape
    K = x.shape[1]
    if x.shape[0] != M:
        raise ValueError("linear_wgrad: row mismatch")
    if out is None:
        out = torch.empty(N, K, dtype=torch.float32, device=x.device)
        accumulate = False
    dbias = None
    if with_bias:
        dbias = bias_out if (bias_out is not None and accumulate) else torch.empty(N, dtype=torch.float32, device=x.device)
    nb = lib.mmg_linear_wgrad_ws_bytes(M, N, K)
    _tok = _pb("linear_wgrad")
    if defer is None:
        ws = workspace(nb, x.device)
        check(lib.mmg_linear_wgrad(_p(dy), _p(x), _pro(pro), _p(out), _p(dbias), M, N, K, int(accumulate),
                                   _p(ws, torch.uint8), ws.numel(), _stream()), "mmg_linear_wgrad")
    else:
        if accumulate and lib.mmg_linear_wgrad_is_direct(M, N, K) and any(j[2] is out for j in defer):
            wgrad_reduce_flush(defer)        # a small launch adds into `out` in place: its earlier slabs must be summed first
        ws = torch.empty(max(int(nb), 256), dtype=torch.uint8, device=x.device)      # its own slabs: they live until the flush
        job = WgradReduceT()
        check(lib.mmg_linear_wgrad_deferred(_p(dy), _p(x), _pro(pro), _p(out), _p(dbias), M, N, K, int(accumulate),
                                            _p(ws, torch.uint8), ws.numel(), _stream(), C.byref(job)),
              "mmg_linear_wgrad_deferred")
        defer.append((job, ws, out, dbias))
    _pe(_tok, "linear_wgrad", 4 * (M * N + M * K + N * K), 2 * M * N * K)
    return (out, dbias) if with_bias else out


def wgrad_reduce_flush(jobs: list):
    """Sum the slabs of every deferred weight gradient (linear_wgrad(defer=...)): one launch per <= 16 jobs; a job that
    accumulates into a gradient an earlier job of the list writes goes into a later launch.  Empties the list."""
    lib = _lib.load()
    todo = [j for j in jobs if j[0].slab]
    jobs.clear()
    while todo:
        group, later, seen = [], [], set()
        for j in todo:
            if j[0].dW in seen or len(group) == 16:
                later.append(j)
            else:
                group.append(j)
            seen.add(j[0].dW)            # (also blocks every LATER job of that gradient: the order of its sums is kept)
        arr = (WgradReduceT * len(group))(*[j[0] for j in group])
        check(lib.mmg_wgrad_reduce_group(arr, len(group), _stream()), "mmg_wgrad_reduce_group")
        todo = later


def col_reduce2(a: torch.Tensor, b: Optional[torch.Tensor] = None):
    """-> fp64 [2,N]: (sum_m a, sum_m a*b) with b = a when omitted."""
    lib = _lib.load()
    M, N = a.shape
    out = torch.empty(2, N, dtype=torch.float64, device=a.device)
    nb = lib.mmg_col_reduce2_ws_bytes(M, N)
    ws = workspace(nb, a.device)
    _tok = _pb("col_reduce2")
    check(lib.mmg_col_reduce2(_p(a), _p(b), _p(out, torch.float64), M, N, _p(ws, torch.uint8), ws.numel(), _stream()),
          "mmg_col_reduce2")
    _pe(_tok, "col_reduce2", 4 * M * N * (2 if b is not None else 1), 0)
    return out


@dataclass
class BNFold:
    scale: torch.Tensor
    shift: torch.Tensor
    mean: torch.Tensor
    rstd: torch.Tensor
    count: int
    training: bool


@dataclass
class NextBN:
    """The BatchNorm backward that CONSUMES an op's output (mmg_next_bn_t): y = its pre-BatchNorm activation, pro = its
    fold / activation / dropout, fold = its BNFold (mean, rstd).  An op that is handed one also returns the fp64 [2,N]
    statistics bn_bwd_stats(out, y, pro, fold) would compute -- from its epilogue where the shape has a fused form, from
    the separate pass elsewhere.  sums: ADD to these sums instead (two producers through one BatchNorm)."""
    y: torch.Tensor
    pro: "Pro"
    fold: "BNFold"
    sums: Optional[torch.Tensor] = None


def _next_bn(nb: NextBN, M: int, N: int):
    """-> (NextBnT for the C call (keeps its prologue alive), the sums tensor the call fills)."""
    lib = _lib.load()
    if tuple(nb.y.shape) != (M, N):
        raise ValueError(f"next_bn: y is {tuple(nb.y.shape)}, the producer writes [{M},{N}]")
    acc = nb.sums is not None
    sums = nb.sums if acc else torch.empty(2, N, dtype=torch.float64, device=nb.y.device)
    if tuple(sums.shape) != (2, N) or sums.dtype != torch.float64:
        raise ValueError("next_bn: sums must be fp64 [2,N]")
    ws = workspace(lib.mmg_next_bn_ws_bytes(M, N), nb.y.device)
    pc = nb.pro.c()
    t = NextBnT(_p(nb.y).value, C.pointer(pc), _p(nb.fold.mean).value, _p(nb.fold.rstd).value,
                _p(sums, torch.float64).value, int(acc), _p(ws, torch.uint8).value, ws.numel())
    t._keep = (pc, ws)
    return t, sums


def bn_finalize(sums: Optional[torch.Tensor], count: int, gamma, beta, running_mean, running_var, training: bool,
                n_updates: int = 1) -> BNFold:
    lib = _lib.load()
    N = gamma.numel()
    dev = gamma.device
    st = torch.empty(4, N, dtype=torch.float32, device=dev)
    check(lib.mmg_bn_finalize(_p(sums, torch.float64), count, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                              int(training), n_updates, BN_MOMENTUM, BN_EPS, _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]),
                              N, _stream()), "mmg_bn_finalize")
    return BNFold(st[0], st[1], st[2], st[3], count, training)


def affine_act_drop(y: torch.Tensor, pro: Pro, out: Optional[torch.Tensor] = None):
    lib = _lib.load()
    M, N = y.shape
    out = torch.empty_like(y) if out is None else out
    _tok = _pb("affine_act_drop")
    check(lib.mmg_affine_act_drop(_p(y), _pro(pro), _p(out), M, N, _stream()), "mmg_affine_act_drop")
    _pe(_tok, "affine_act_drop", 8 * M * N, 0)
    return out


def affine_act_drop_rows(y: torch.Tensor, pro: Pro, rows: torch.Tensor):
    """dropout(relu(y[rows]*scale+shift)) for the selected rows (int64 ids); the dropout mask is the one the full tensor
    would get at those rows."""
    lib = _lib.load()
    N = y.shape[1]
    n = rows.numel()
    out = torch.empty(n, N, dtype=torch.float32, device=y.device)
    check(lib.mmg_affine_act_drop_rows(_p(y), _pro(pro), _p(rows, torch.int64), n, _p(out), N, _stream()),
          "mmg_affine_act_drop_rows")
    return out


def bn_bwd_stats(g: torch.Tensor, y: torch.Tensor, pro: Pro, fold: BNFold):
    lib = _lib.load()
    M, N = y.shape
    out = torch.empty(2, N, dtype=torch.float64, device=y.device)
    nb = lib.mmg_col_reduce2_ws_bytes(M, N)
    ws = workspace(nb, y.device)
    _tok = _pb("bn_bwd_stats")
    check(lib.mmg_bn_bwd_stats(_p(g), _p(y), _pro(pro), _p(fold.mean), _p(fold.rstd), _p(out, torch.float64), M, N,
                               _p(ws, torch.uint8), ws.numel(), _stream()), "mmg_bn_bwd_stats")
    _pe(_tok, "bn_bwd_stats", 8 * M * N, 0)
    return out


def bn_bwd_stats2(g: torch.Tensor, g2: torch.Tensor, y: torch.Tensor, pro: Pro, pro2: Pro, fold: BNFold):
    """bn_bwd_stats of two upstream gradients through the same BatchNorm + ReLU with their own dropout masks."""
    lib = _lib.load()
    M, N = y.shape
    out = torch.empty(2, N, dtype=torch.float64, device=y.device)
    nb = lib.mmg_col_reduce2_ws_bytes(M, N)
    ws = workspace(nb, y.device)
    _tok = _pb("bn_bwd_stats")
    check(lib.mmg_bn_bwd_stats2(_p(g), _p(g2), _p(y), _pro(pro), _pro(pro2), _p(fold.mean), _p(fold.rstd),
                                _p(out, torch.float64), M, N, _p(ws, torch.uint8), ws.numel(), _stream()), "mmg_bn_bwd_stats2")
    _pe(_tok, "bn_bwd_stats", 12 * M * N, 0)
    return out


def bn_bwd_apply2(g: torch.Tensor, g2: torch.Tensor, y: torch.Tensor, pro: Pro, pro2: Pro, fold: BNFold, sums, count,
                  dbeta=None, dgamma=None):
    lib = _lib.load()
    M, N = y.shape
    out = torch.empty_like(y)
    _tok = _pb("bn_bwd_apply")
    check(lib.mmg_bn_bwd_apply2(_p(g), _p(g2), _p(y), _pro(pro), _pro(pro2), _p(fold.mean), _p(fold.rstd),
                                _p(sums, torch.float64), 1.0 / float(count), _p(dbeta), _p(dgamma), _p(out), M, N, _stream()),
          "mmg_bn_bwd_apply2")
    _pe(_tok, "bn_bwd_apply", 16 * M * N, 0)
    return out


def bn_bwd_stats_rows(g_rows: torch.Tensor, y: torch.Tensor, rows: torch.Tensor, pro: Pro, fold: BNFold):
    """bn_bwd_stats for an upstream gradient that is zero outside `rows` (g_rows = its rows, in list order)."""
    lib = _lib.load()
    N = y.shape[1]
    out = torch.empty(2, N, dtype=torch.float64, device=y.device)
    ws = workspace(lib.mmg_bn_bwd_stats_rows_ws_bytes(N), y.device)
    check(lib.mmg_bn_bwd_stats_rows(_p(g_rows), _p(y), _p(rows, torch.int64), rows.numel(), _pro(pro), _p(fold.mean),
                                    _p(fold.rstd), _p(out, torch.float64), N, _p(ws, torch.uint8), ws.numel(), _stream()),
          "mmg_bn_bwd_stats_rows")
    return out


def bn_bwd_apply_rows(g_rows: torch.Tensor, y: torch.Tensor, rows: torch.Tensor, pro: Pro, dy: torch.Tensor):
    """dy[rows] += scale * g'  -- completes a bn_bwd_apply(None, ...) for the listed (distinct) rows."""
    lib = _lib.load()
    check(lib.mmg_bn_bwd_apply_rows(_p(g_rows), _p(y), _p(rows, torch.int64), rows.numel(), _pro(pro), _p(dy),
                                    y.shape[1], _stream()), "mmg_bn_bwd_apply_rows")
    return dy


def bn_bwd_apply(g: Optional[torch.Tensor], y: torch.Tensor, pro: Pro, fold: Optional[BNFold], sums=None, count: float = 1.0,
                 dbeta=None, dgamma=None, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """sums: the fp64 [2,N] output of bn_bwd_stats (None in eval mode); dbeta / dgamma ([N] float) receive its rows.
    accumulate: out += (needs out).  g = None: an all-zero upstream gradient (see bn_bwd_apply_rows)."""
    lib = _lib.load()
    M, N = y.shape
    if accumulate and out is None:
        raise ValueError("accumulate needs out")
    out = torch.empty_like(y) if out is None else out
    _tok = _pb("bn_bwd_apply")
    check(lib.mmg_bn_bwd_apply(_p(g), _p(y), _pro(pro), _p(fold.mean) if fold else None,
                               _p(fold.rstd) if fold else None, _p(sums, torch.float64), 1.0 / float(count),
                               _p(dbeta), _p(dgamma), _p(out), M, N, int(accumulate), _stream()),
          "mmg_bn_bwd_apply")
    _pe(_tok, "bn_bwd_apply", (16 if accumulate else 12) * M * N, 0)
    return out


def linear_l2norm_fwd(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, pro: Optional[Pro] = None):
    """l2norm_fwd(linear_fwd(x, W, bias, pro)) -> (normalised rows, rn): ONE kernel where the GEMM's workgroup holds whole
    rows (mmg_linear_fwd_l2norm: the norm is taken in the epilogue), the two launches elsewhere."""
    lib = _lib.load()
    M, K = x.shape
    N = W.shape[0]
    if not lib.mmg_linear_fwd_l2norm_supported(M, N, K):
        return l2norm_fwd(linear_fwd(x, W, bias, pro=pro))
    out = torch.empty(M, N, device=x.device)
    rn = torch.empty(M, device=x.device)
    _tok = _pb("linear_fwd")
    check(lib.mmg_linear_fwd_l2norm(_p(x), _pro(pro), _p(W), _p(bias), _p(out), _p(rn), M, N, K, L2_EPS, _stream()),
          "mmg_linear_fwd_l2norm")
    _pe(_tok, "linear_fwd", 4 * (M * K + N * K + M * N), 2 * M * N * K)
    return out, rn


def linear_bnbwd_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().mmg_linear_bnbwd_supported(int(M), int(N), int(K)))


def linear_bnbwd(g: torch.Tensor, y: torch.Tensor, pro: Pro, fold: Optional[BNFold], W: torch.Tensor, sums=None,
                 count: float = 1.0, dbeta=None, dgamma=None, next_bn: Optional["NextBN"] = None):
    """bn_bwd_apply(g, y, ...) and the data gradient dz @ W of the linear in front of that BatchNorm in ONE pass over g and
    y (mmg_linear_bnbwd): -> (dz [M,K], dx [M,N]).  W [K, N] is the forward weight of the linear, read in place."""
    lib = _lib.load()
    M, K = y.shape
    if W.shape[0] != K:
        raise ValueError(f"linear_bnbwd: W has {W.shape[0]} rows, y has {K} columns")
    N = W.shape[1]
    dz = torch.empty_like(y)
    dx = torch.empty(M, N, device=y.device)
    _tok = _pb("linear_bnbwd")
    if next_bn is not None:       # -> (dz, dx, the statistics of the BatchNorm backward that consumes dx), see NextBN
        nbt, nsums = _next_bn(next_bn, M, N)
        check(lib.mmg_linear_bnbwd_next_bn(_p(g), _p(y), _pro(pro), _p(fold.mean) if fold else None,
                                           _p(fold.rstd) if fold else None, _p(sums, torch.float64), 1.0 / float(count),
                                           _p(dbeta), _p(dgamma), _p(W), _p(dz), _p(dx), M, N, K, C.byref(nbt), _stream()),
              "mmg_linear_bnbwd_next_bn")
        _pe(_tok, "linear_bnbwd", 4 * (3 * M * K + 2 * M * N), 2 * M * N * K)
        return dz, dx, nsums
    check(lib.mmg_linear_bnbwd(_p(g), _p(y), _pro(pro), _p(fold.mean) if fold else None, _p(fold.rstd) if fold else None,
                               _p(sums, torch.float64), 1.0 / float(count), _p(dbeta), _p(dgamma), _p(W), _p(dz), _p(dx),
                               M, N, K, _stream()), "mmg_linear_bnbwd")
    _pe(_tok, "linear_bnbwd", 4 * (3 * M * K + M * N), 2 * M * N * K)
    return dz, dx


def linear_bnbwd2_supported(M: int, N: int, K: int) -> bool:
    return N == 128 and K == 128 and linear_bnbwd_supported(M, N, K)


def linear_bnbwd2(g: torch.Tensor, g2: torch.Tensor, y: torch.Tensor, pro: Pro, pro2: Pro, fold: BNFold, W: torch.Tensor,
                  sums, count, dbeta=None, dgamma=None):
    """bn_bwd_apply2 (two upstream gradients, own dropout masks) + the data gradient dz @ W in one pass -> (dz, dx)."""
    lib = _lib.load()
    M, K = y.shape
    N = W.shape[1]
    dz = torch.empty_like(y)
    dx = torch.empty(M, N, device=y.device)
    _tok = _pb("linear_bnbwd")
    check(lib.mmg_linear_bnbwd2(_p(g), _p(g2), _p(y), _pro(pro), _pro(pro2), _p(fold.mean), _p(fold.rstd),
                                _p(sums, torch.float64), 1.0 / float(count), _p(dbeta), _p(dgamma), _p(W), _p(dz), _p(dx),
                                M, N, K, _stream()), "mmg_linear_bnbwd2")
    _pe(_tok, "linear_bnbwd", 4 * (4 * M * K + M * N), 2 * M * N * K)
    return dz, dx


def linear_bnbwd_rows(g_rows: torch.Tensor, row_pos: torch.Tensor, y: torch.Tensor, pro: Pro, fold: BNFold, W: torch.Tensor,
                      sums, count, dbeta=None, dgamma=None, next_bn: Optional["NextBN"] = None):
    """bn_bwd_apply(None, ...) + bn_bwd_apply_rows(g_rows, ...) + the data gradient dz @ W in one pass -> (dz, dx): the
    upstream gradient is zero outside the listed rows; row_pos [M] int32 = position of a row in the list or -1."""
    lib = _lib.load()
    M, K = y.shape
    N = W.shape[1]
    dz = torch.empty_like(y)
    dx = torch.empty(M, N, device=y.device)
    _tok = _pb("linear_bnbwd")
    if next_bn is not None:       # -> (dz, dx, the statistics of the BatchNorm backward that consumes dx), see NextBN
        nbt, nsums = _next_bn(next_bn, M, N)
        check(lib.mmg_linear_bnbwd_rows_next_bn(_p(g_rows) if g_rows.numel() else None, _p(row_pos, torch.int32),
                                                g_rows.shape[0], _p(y), _pro(pro), _p(fold.mean), _p(fold.rstd),
                                                _p(sums, torch.float64), 1.0 / float(count), _p(dbeta), _p(dgamma), _p(W),
                                                _p(dz), _p(dx), M, N, K, C.byref(nbt), _stream()),
              "mmg_linear_bnbwd_rows_next_bn")
        _pe(_tok, "linear_bnbwd", 4 * (2 * M * K + 2 * M * N), 2 * M * N * K)
        return dz, dx, nsums
    check(lib.mmg_linear_bnbwd_rows(_p(g_rows) if g_rows.numel() else None, _p(row_pos, torch.int32), g_rows.shape[0], _p(y),
                                    _pro(pro), _p(fold.mean), _p(fold.rstd), _p(sums, torch.float64), 1.0 / float(count),
                                    _p(dbeta), _p(dgamma), _p(W), _p(dz), _p(dx), M, N, K, _stream()),
          "mmg_linear_bnbwd_rows")
    _pe(_tok, "linear_bnbwd", 4 * (2 * M * K + M * N), 2 * M * N * K)
    return dz, dx


def linear_l2bwd(g: torch.Tensor, out: torch.Tensor, rn: torch.Tensor, W: torch.Tensor, next_bn: Optional["NextBN"] = None):
    """l2norm_bwd(g, out, rn) and the data gradient dz @ W of the linear in front of the normalisation -> (dz, dx): ONE
    kernel where mmg_linear_bnbwd_supported (W [K, N] = the forward weight in place), the two launches elsewhere."""
    lib = _lib.load()
    M, K = out.shape
    N = W.shape[1]
    if W.shape[0] != K:
        raise ValueError(f"linear_l2bwd: W has {W.shape[0]} rows, out has {K} columns")
    if not lib.mmg_linear_bnbwd_supported(M, N, K):
        dz = l2norm_bwd(g, out, rn)
        if next_bn is not None:
            return (dz,) + tuple(linear_fwd(dz, W, w_kn=True, next_bn=next_bn))
        return dz, linear_fwd(dz, W, w_kn=True)
    dz = torch.empty_like(out)
    dx = torch.empty(M, N, device=out.device)
    _tok = _pb("linear_l2bwd")
    if next_bn is not None:       # -> (dz, dx, the statistics of the BatchNorm backward that consumes dx), see NextBN
        nbt, nsums = _next_bn(next_bn, M, N)
        check(lib.mmg_linear_l2bwd_next_bn(_p(g), _p(out), _p(rn), _p(W), _p(dz), _p(dx), M, N, K, L2_EPS, C.byref(nbt),
                                           _stream()), "mmg_linear_l2bwd_next_bn")
        _pe(_tok, "linear_l2bwd", 4 * (3 * M * K + 2 * M * N), 2 * M * N * K)
        return dz, dx, nsums
    check(lib.mmg_linear_l2bwd(_p(g), _p(out), _p(rn), _p(W), _p(dz), _p(dx), M, N, K, L2_EPS, _stream()), "mmg_linear_l2bwd")
    _pe(_tok, "linear_l2bwd", 4 * (3 * M * K + M * N), 2 * M * N * K)
    return dz, dx


def l2norm_fwd(z: torch.Tensor):
    lib = _lib.load()
    M, N = z.shape
    out = torch.empty_like(z)
    rn = torch.empty(M, dtype=torch.float32, device=z.device)
    _tok = _pb("l2norm_fwd")
    check(lib.mmg_l2norm_fwd(_p(z), _p(out), _p(rn), M, N, L2_EPS, _stream()), "mmg_l2norm_fwd")
    _pe(_tok, "l2norm_fwd", 8 * M * N, 0)
    return out, rn


def l2norm_bwd(g: torch.Tensor, out: torch.Tensor, rn: torch.Tensor):
    lib = _lib.load()
    M, N = out.shape
    dz = torch.empty_like(out)
    _tok = _pb("l2norm_bwd")
    check(lib.mmg_l2norm_bwd(_p(g), _p(out), _p(rn), _p(dz), M, N, L2_EPS, _stream()), "mmg_l2norm_bwd")
    _pe(_tok, "l2norm_bwd", 12 * M * N, 0)
    return dz


def dropout_mask(seed: int, site: int, n_rows: int, width: int, p: float, device, row_offset: int = 0, seed_dev=None):
    """The keep-mask ([n_rows, width] uint8) the kernels draw for (seed, site) -- used to inject the
    same masks into the CPU oracle in parity tests."""
    lib = _lib.load()
    m = torch.empty(n_rows, width, dtype=torch.uint8, device=device)
    check(lib.mmg_dropout_mask(seed & 0xFFFFFFFFFFFFFFFF, _p(seed_dev, torch.int64), site, row_offset * width,
                               n_rows * width, float(p), _p(m, torch.uint8), _stream()), "mmg_dropout_mask")
    return m


# ------------------------------------------------------------------------------------------ heads
@dataclass
class Head:
    A: torch.Tensor     # [P,64]
    B: torch.Tensor     # [L,64]
    W2: torch.Tensor    # [32,64]
    b2: torch.Tensor    # [32]
    W3: torch.Tensor    # [32] (mlp.6.weight flattened)
    b3: torch.Tensor    # [1]

    def c(self):
        return HeadT(_p(self.A), _p(self.B), _p(self.W2), _p(self.b2), _p(self.W3), _p(self.b3))


def pair_select(pi, deg, thr: int, dpred=None, io_perm=None, dpred_sorted=None, out=None):
    """Stable compaction of pair positions by head (mmg_pair_select) -> (sel_low, sel_high, counts[2] on device).
    With `dpred`, positions whose upstream gradient is exactly 0 are dropped (dpred is read through io_perm; if
    `dpred_sorted` ([n] float) is given it receives dpred in pair order).  out: the three tensors of an earlier call, to
    be overwritten in place (a captured step holds their addresses)."""
    lib = _lib.load()
    n = pi.numel()
    if out is not None:
        sel_low, sel_high, counts = out
    else:
        sel_low = torch.empty(max(n, 1), dtype=torch.int32, device=pi.device)
        sel_high = torch.empty(max(n, 1), dtype=torch.int32, device=pi.device)
        counts = torch.empty(2, dtype=torch.int32, device=pi.device)
    ws = workspace(lib.mmg_pair_select_ws_bytes(n), pi.device)
    _tok = _pb("pair_select")
    check(lib.mmg_pair_select(_p(pi, torch.int32), _p(deg, torch.int32), thr, _p(dpred), _p(io_perm, torch.int64),
                              _p(dpred_sorted), n, _p(sel_low, torch.int32),
                              _p(sel_high, torch.int32), _p(counts, torch.int32), _p(ws, torch.uint8), ws.numel(),
                              _stream()), "mmg_pair_select")
    _pe(_tok, "pair_select", n * 24)
    return sel_low, sel_high, counts


def _pair_sizes(head: Head, pi, li, deg, pair_id, io_perm, per_pair):
    """The array lengths the kernels range-check against (mmg_pair_head_*: n_total, n_patients, n_labs) must be the real
    ones: every per-pair array has pi's length."""
    n = pi.numel()
    for name, t in (("li", li), ("pair_id", pair_id), ("io_perm", io_perm), ("pred / dpred", per_pair)):
        if t is not None and t.numel() != n:
            raise ValueError(f"pair head: {name} has {t.numel()} entries, pi has {n}")
    if head.A.dim() != 2 or head.A.shape[1] != 64 or head.B.dim() != 2 or head.B.shape[1] != 64:
        raise ValueError("pair head: A and B must be [rows, 64]")


def pair_saved_alloc(n_total: int, device):
    """Buffers of mmg_pair_saved_t: (h1 sign bits int32 [n, 2], layer-2 activations float32 [n, 32]), 136 B per entry.
    n = the pairs of the pair set (entries indexed by pair), or the launch bound of the head's pair list when the entries
    are indexed by list position (by_position).  The forward fills the entries it visits, the backward reads those."""
    return (torch.empty(max(n_total, 1), 2, dtype=torch.int32, device=device),
            torch.empty(max(n_total, 1), 32, dtype=torch.float32, device=device))


def _pair_saved(saved, n_total: int, n_launch: int):
    """saved = (bits, h2) or (bits, h2, by_position): by_position -- entries indexed by the position in the pair list
    instead of the pair index (dense); forward and backward must then run over the same list.  Sizes: one entry per pair
    of the pair set (indexed by pair), or per list position the launch can reach (n_launch, its bound) when by_position."""
    if saved is None:
        return None
    bits, h2 = saved[0], saved[1]
    by_pos = bool(saved[2]) if len(saved) > 2 else False
    need = max(int(n_launch) if by_pos else int(n_total), 1)
    if bits.dtype != torch.int32 or bits.dim() != 2 or bits.shape[1] != 2 or h2.dtype != torch.float32 or h2.dim() != 2 or \
            h2.shape[1] != 32 or bits.shape[0] != h2.shape[0] or not bits.is_contiguous() or not h2.is_contiguous():
        raise ValueError("pair head: saved = (int32 [n, 2], float32 [n, 32]) -- see pair_saved_alloc")
    if bits.shape[0] < need:
        raise ValueError(f"pair head: saved buffers hold {bits.shape[0]} entries, the launch reaches {need}"
                         + (" list positions" if by_pos else " pairs"))
    return PairSavedT(_p(bits, torch.int32).value, _p(h2).value, int(by_pos), int(bits.shape[0]))


def pair_head_fwd(head: Head, pi, li, deg, thr: int, want_low: bool, p: float, seed: int, pair_id, pred, seed_dev=None,
                  sel=None, n_sel=None, n_bound: Optional[int] = None, io_perm=None, save=None):
    """sel / n_sel: compacted positions (pair_select) and their device-resident count; n_bound >= that count.
    io_perm: pred is written to pred[io_perm[k]] (the caller's pair order).
    save = pair_saved_alloc(...): also leave what the backward needs per visited pair (mmg_pair_saved_t)."""
    lib = _lib.load()
    n = pi.numel() if sel is None else int(n_bound)
    if n == 0:
        return
    h = head.c()
    _pair_sizes(head, pi, li, deg, pair_id, io_perm, pred)
    sv = _pair_saved(save, pi.numel(), n)
    _tok = _pb("pair_head_fwd")
    check(lib.mmg_pair_head_fwd_save(C.byref(h), _p(pi, torch.int32), _p(li, torch.int32), _p(deg, torch.int32), thr,
                                     int(want_low), n, pi.numel(), min(int(head.A.shape[0]), deg.numel()),
                                     int(head.B.shape[0]), float(p), seed & 0xFFFFFFFFFFFFFFFF,
                                     _p(seed_dev, torch.int64),
                                     _p(pair_id, torch.int64), _p(pred), _p(sel, torch.int32), _p(n_sel, torch.int32),
                                     _p(io_perm, torch.int64), C.byref(sv) if sv is not None else None, _stream()),
          "mmg_pair_head_fwd_save")
    _pe(_tok, "pair_head_fwd", n * (12 + (136 if sv is not None else 0)) + 256 * (head.A.shape[0] + head.B.shape[0]),
        n * 2 * (64 * 32 + 32 + 64))


def pair_head_bwd(head: Head, grads: Head, pi, li, deg, thr: int, want_low: bool, n_labs: int, p: float, seed: int,
                  pair_id, dpred, seed_dev=None, sel=None, n_sel=None, n_bound: Optional[int] = None, io_perm=None,
                  saved=None):
    """`grads` mirrors `head` (dA,dB,dW2,db2,dW3,db3), accumulated in place.
    saved: what pair_head_fwd(..., save=...) of the SAME head, gate, seed and pair arrays left (it must have visited every
    pair this call visits); None = recompute."""
    lib = _lib.load()
    n = pi.numel() if sel is None else int(n_bound)
    if n == 0:
        return
    h = head.c()
    g = HeadGradT(_p(grads.A), _p(grads.B), _p(grads.W2), _p(grads.b2), _p(grads.W3), _p(grads.b3))
    ws = workspace(lib.mmg_pair_head_bwd_ws_bytes(n, n_labs), head.A.device)
    _pair_sizes(head, pi, li, deg, pair_id, io_perm, dpred)
    if tuple(grads.A.shape) != tuple(head.A.shape) or tuple(grads.B.shape) != tuple(head.B.shape):
        raise ValueError("pair_head_bwd: gradient tables must have the shapes of A and B")
    sv = _pair_saved(saved, pi.numel(), n)
    _tok = _pb("pair_head_bwd")
    check(lib.mmg_pair_head_bwd_saved(C.byref(h), C.byref(g), _p(pi, torch.int32), _p(li, torch.int32),
                                      _p(deg, torch.int32), thr, int(want_low), n, pi.numel(),
                                      min(int(head.A.shape[0]), deg.numel()), n_labs, float(p), seed & 0xFFFFFFFFFFFFFFFF,
                                      _p(seed_dev, torch.int64), _p(pair_id, torch.int64), _p(dpred), _p(sel, torch.int32),
                                      _p(n_sel, torch.int32), _p(io_perm, torch.int64),
                                      C.byref(sv) if sv is not None else None, _p(ws, torch.uint8), ws.numel(), _stream()),
          "mmg_pair_head_bwd_saved")
    _pe(_tok, "pair_head_bwd", n * 12 + 2 * 256 * (head.A.shape[0] + head.B.shape[0]), n * 2 * (4 * 64 * 32))


# ------------------------------------------------------------------------------------------ loss
LOSS_TYPES = {"mae": 0, "mse": 1, "huber": 2}


def sup_mask_draw(n: int, fraction: float, device, seed: int = 0, seed_dev=None, ids=None, sup=None, count=None,
                  inv_den=None, count_only: bool = False):
    """The per-epoch supervision subset drawn on the device (mmg_sup_mask_draw; train.py:150-176 of the reference)
    -> (sup float [n], count fp64 [1], inv_den fp64 [1] = 1 / max(count, 1)); the three may be passed in (a captured step
    overwrites them in place).  seed_dev: int64 device tensor whose first element is the seed at run time."""
    lib = _lib.load()
    if count_only:                       # the size of the subset of ids 0 .. n-1, nothing written per pair
        sup = None
    else:
        sup = torch.empty(max(n, 1), dtype=torch.float32, device=device)[:n] if sup is None else sup
    count = torch.empty(1, dtype=torch.float64, device=device) if count is None else count
    inv_den = torch.empty(1, dtype=torch.float64, device=device) if inv_den is None else inv_den
    ws = workspace(lib.mmg_sup_mask_ws_bytes(n), device)
    check(lib.mmg_sup_mask_draw(_p(seed_dev, torch.int64), int(seed) & 0xFFFFFFFFFFFFFFFF, _p(ids, torch.int64), n,
                                float(fraction), _p(sup) if (n and sup is not None) else None, _p(count, torch.float64),
                                _p(inv_den, torch.float64), _p(ws, torch.uint8), ws.numel(), _stream()), "mmg_sup_mask_draw")
    return sup, count, inv_den


def pair_loss(pred, y, w=None, sup=None, inv_den: float = 1.0, loss_type: str = "mae", inv_den_dev=None, loss_out=None,
              want_dpred: bool = True):
    """-> (loss fp64 scalar tensor, dpred [n]) in one pass (mmg_pair_loss).  inv_den_dev: fp64 [1] device tensor that
    overrides inv_den at run time (a captured step whose supervision subset changes size between replays).
    loss_out: fp64 device scalar to write the loss to (a slot of the caller's buffer); want_dpred = False: loss only."""
    lib = _lib.load()
    n = pred.numel()
    dpred = torch.empty_like(pred) if want_dpred else None
    loss = torch.empty((), dtype=torch.float64, device=pred.device) if loss_out is None else loss_out
    if loss.numel() != 1:
        raise ValueError("pair_loss: loss_out must hold one fp64 value")
    ws = workspace(lib.mmg_pair_loss_ws_bytes(n), pred.device)
    if loss_type not in LOSS_TYPES:
        raise ValueError(f"Unknown loss type: {loss_type}")            # (model.py:610 of the reference)
    lt = LOSS_TYPES[loss_type]
    _tok = _pb("pair_loss")
    check(lib.mmg_pair_loss(_p(pred), _p(y), _p(w), _p(sup), n, float(inv_den), _p(inv_den_dev, torch.float64), lt,
                            _p(dpred), _p(loss, torch.float64),
                            _p(ws, torch.uint8), ws.numel(), _stream()), "mmg_pair_loss")
    _pe(_tok, "pair_loss", 20 * n)
    return loss, dpred


class _PairLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, y, w, sup, inv_den, loss_type, inv_den_dev=None):
        loss, dpred = pair_loss(pred.detach().contiguous(), y, w, sup, inv_den, loss_type, inv_den_dev)
        ctx.save_for_backward(dpred)
        return loss.float()

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None, None, None, None


def weighted_pair_loss(pred, y, w=None, sup=None, inv_den: float = 1.0, loss_type: str = "mae", inv_den_dev=None):
    """Differentiable fused loss: inv_den * sum sup*w*|pred-y| (or squared); w/sup are per-pair float vectors."""
    return _PairLossFn.apply(pred, y, w, sup, inv_den, loss_type, inv_den_dev)


# ------------------------------------------------------------------------------------------ evaluation reducers
def seg_sums(pred: torch.Tensor, target: torch.Tensor, seg: torch.Tensor, n_seg: int, n_sigma: float = 0.0,
             want_adjusted: bool = False):
    """Segment sums of the evaluation metrics (mmg_seg_moments + mmg_seg_metrics) -> (sums fp64 [n_seg, 8], adjusted
    predictions or None).  sums[s] = (n, sum|e|, sum e^2, sum t, sum t^2, sum|e/t| over t != 0, count(t != 0), clipped).
    n_sigma > 0: residuals are clipped to mean +- n_sigma * std of their segment first (evaluate.py:417-440)."""
    lib = _lib.load()
    n = pred.numel()
    dev = pred.device
    ws = workspace(lib.mmg_seg_reduce_ws_bytes(n, n_seg), dev)
    moments = None
    if n_sigma > 0:
        moments = torch.empty(n_seg, 3, dtype=torch.float64, device=dev)
        check(lib.mmg_seg_moments(_p(pred), _p(target), _p(seg, torch.int64), n, n_seg, _p(moments, torch.float64),
                                  _p(ws, torch.uint8), ws.numel(), _stream()), "mmg_seg_moments")
    sums = torch.empty(n_seg, 8, dtype=torch.float64, device=dev)
    adj = torch.empty_like(pred) if want_adjusted else None
    check(lib.mmg_seg_metrics(_p(pred), _p(target), _p(seg, torch.int64), n, n_seg, _p(moments, torch.float64),
                              float(n_sigma), _p(adj), _p(sums, torch.float64), _p(ws, torch.uint8), ws.numel(), _stream()),
          "mmg_seg_metrics")
    return sums, adj


def _p_any(t: torch.Tensor):
    """device pointer of an fp32 tensor that may be a strided view (the caller passes its strides on)."""
    if not t.is_cuda:
        raise _lib.MmgError(f"expected a HIP device tensor, got {t.device} (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"expected torch.float32, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _rows_view(t: torch.Tensor):
    """(cols, row stride) of a tensor that is a flat vector or a 2-D matrix with unit column stride."""
    if t.is_contiguous():
        return 0, 0
    if t.dim() == 2 and t.stride(1) == 1:
        return t.shape[1], t.stride(0)
    raise ValueError("vec_sums: tensors must be contiguous or 2-D with unit column stride")


def vec_sums(jobs):
    """jobs: list of (dst, [src tensors, 1..4]) -- dst = sum of the sources in list order, ONE launch (mmg_vec_sums).
    2-D tensors may be column slices of wider matrices (row stride > columns): the launch that sums gradient
    contributions also splits / joins the halves of an edge head's first-layer weight."""
    lib = _lib.load()
    arr = (SumJobT * len(jobs))()
    for j, (dst, srcs) in enumerate(jobs):
        if not 1 <= len(srcs) <= 4:
            raise ValueError("vec_sums: 1..4 sources per job")
        views = [_rows_view(t) for t in [dst] + list(srcs)]
        strided = any(c for c, _ in views)
        cols = 0
        if strided:
            if dst.dim() != 2 or any(t.shape != dst.shape for t in srcs):
                raise ValueError("vec_sums: strided jobs take 2-D tensors of one shape")
            cols = dst.shape[1]
        sp = (C.c_void_p * 4)()
        ld = (C.c_int * 4)()
        for q, t in enumerate(srcs):
            if t.numel() != dst.numel():
                raise ValueError("vec_sums: size mismatch")
            sp[q] = _p_any(t).value
            ld[q] = (views[q + 1][1] or cols) if strided else 0
        arr[j] = SumJobT(_p_any(dst).value, sp, len(srcs), dst.numel(), cols,
                         (views[0][1] or cols) if strided else 0, ld)
    check(lib.mmg_vec_sums(arr, len(jobs), _stream()), "mmg_vec_sums")


def counters_add(counters, incs):
    """*counters[i] += incs[i] (int64 device scalars) in one launch (mmg_counters_add)."""
    lib = _lib.load()
    for i0 in range(0, len(counters), 32):
        cs, ins = counters[i0:i0 + 32], incs[i0:i0 + 32]
        ptrs = (C.c_void_p * len(cs))(*[_p(c, torch.int64).value for c in cs])
        inc = (C.c_int64 * len(cs))(*[int(v) for v in ins])
        check(lib.mmg_counters_add(ptrs, inc, len(cs), _stream()), "mmg_counters_add")


def seed_advance(state: torch.Tensor):
    """state: int64 [2] on the device = (dropout seed the kernels read, stream position); one SplitMix64 step."""
    check(_lib.load().mmg_seed_advance(_p(state, torch.int64), _stream()), "mmg_seed_advance")


def zeros(*shape, device, dtype=torch.float32):
    """torch.zeros through mmg_fill_zero: a zero-fill KERNEL on the current stream (not hipMemsetAsync -- its hipGraph node
    does not reliably replay the recorded pattern on this ROCm: csrc/common.h, profiles/probes/hipgraph_memset_node.py)."""
    t = torch.empty(*shape, device=device, dtype=dtype)
    check(_lib.load().mmg_fill_zero(_p(t, dtype), t.numel() * t.element_size(), _stream()), "mmg_fill_zero")
    return t


# ------------------------------------------------------------------------------------------ grouped small launches
SMALL_MAX_ROWS = 4096


@dataclass
class SmallFwd:
    """One problem of small_fwd_group: out[M,N] (+)= x @ W^T (+ x2 @ W2^T) + bias (w_kn: W, W2 stored [K,N])."""
    x: torch.Tensor
    W: torch.Tensor
    out: Optional[torch.Tensor] = None
    bias: Optional[torch.Tensor] = None
    x2: Optional[torch.Tensor] = None
    W2: Optional[torch.Tensor] = None
    accumulate: bool = False
    w_kn: bool = False


def small_fwd_group(probs: Sequence[SmallFwd]):
    """Runs the problems (same N and K) in ceil(len / 8) launches; allocates missing outputs; returns the outputs."""
    lib = _lib.load()
    if not probs:
        return []
    K = probs[0].x.shape[1]
    N = probs[0].W.shape[1] if probs[0].w_kn else probs[0].W.shape[0]
    outs = []
    for p in probs:
        M, k = p.x.shape
        n = p.W.shape[1] if p.w_kn else p.W.shape[0]
        if k != K or n != N or (p.W.shape[0] if p.w_kn else p.W.shape[1]) != K or M > SMALL_MAX_ROWS:
            raise ValueError("small_fwd_group: every problem must share (N, K) and have M <= 4096")
        if p.out is None:
            if p.accumulate:
                raise ValueError("accumulate needs out")
            p.out = torch.empty(M, N, dtype=torch.float32, device=p.x.device)
        elif tuple(p.out.shape) != (M, N):
            raise ValueError("small_fwd_group: out shape")
        if (p.x2 is None) != (p.W2 is None) or (p.x2 is not None and (tuple(p.x2.shape) != (M, K) or p.W2.shape != p.W.shape)):
            raise ValueError("small_fwd_group: second term shape")
        outs.append(p.out)
    live = [p for p in probs if p.x.shape[0] > 0]              # (an empty table has an empty output)
    for i0 in range(0, len(live), 8):
        chunk = live[i0:i0 + 8]
        arr = (SmallFwdT * len(chunk))()
        for i, p in enumerate(chunk):
            arr[i] = SmallFwdT(_p(p.x, name="x").value, _p(p.W, name="W").value,
                               _p(p.x2).value if p.x2 is not None else None,
                               _p(p.W2).value if p.W2 is not None else None,
                               _p(p.bias).value if p.bias is not None else None,
                               _p(p.out).value, p.x.shape[0], int(p.accumulate) | (2 if p.w_kn else 0))
        check(lib.mmg_small_fwd_group(arr, len(chunk), N, K, _stream()), "mmg_small_fwd_group")
    return outs


@dataclass
class SmallWgrad:
    """One problem of small_wgrad_group: dW[N,K] (+)= dy^T @ x; dbias (+)= column sums of dy (with_bias)."""
    dy: torch.Tensor
    x: torch.Tensor
    dW: Optional[torch.Tensor] = None
    dbias: Optional[torch.Tensor] = None
    with_bias: bool = False
    accumulate: bool = False


def small_wgrad_group(probs: Sequence[SmallWgrad]):
    """Runs the problems (same N and K) in ceil(len / 8) launches; allocates missing outputs; returns [(dW, dbias)]."""
    lib = _lib.load()
    if not probs:
        return []
    N, K = probs[0].dy.shape[1], probs[0].x.shape[1]
    res = []
    for p in probs:
        M = p.dy.shape[0]
        if p.dy.shape[1] != N or p.x.shape[1] != K or p.x.shape[0] != M or M > SMALL_MAX_ROWS:
            raise ValueError("small_wgrad_group: every problem must share (N, K), rows must match, M <= 4096")
        if p.dW is None:
            p.dW = torch.empty(N, K, dtype=torch.float32, device=p.x.device)
            p.accumulate = False
        if p.with_bias and p.dbias is None:
            if p.accumulate:
                raise ValueError("accumulating the bias gradient needs dbias")
            p.dbias = torch.empty(N, dtype=torch.float32, device=p.x.device)
        res.append((p.dW, p.dbias))
    for i0 in range(0, len(probs), 8):
        chunk = probs[i0:i0 + 8]
        arr = (SmallWgradT * len(chunk))()
        for i, p in enumerate(chunk):
            arr[i] = SmallWgradT(_p(p.dy).value if p.dy.numel() else None, _p(p.x).value if p.x.numel() else None,
                                 _p(p.dW).value, _p(p.dbias).value if p.dbias is not None else None, p.dy.shape[0],
                                 int(p.accumulate))
        check(lib.mmg_small_wgrad_group(arr, len(chunk), N, K, _stream()), "mmg_small_wgrad_group")
    return res


def small_bn_act_group(items, training: bool):
    """items: list of (y [M,N], bn module or None, Pro with the activation / dropout fields) -- the per-type epilogue of
    a HeteroConv layer for every small node type in ONE launch (mmg_small_bn_act_group).
    -> list of (out, BNFold or None); the Pro objects get their scale / shift filled in."""
    lib = _lib.load()
    res = []
    if not items:
        return res
    N = items[0][0].shape[1]
    for i0 in range(0, len(items), 8):
        chunk = items[i0:i0 + 8]
        arr = (SmallBnT * len(chunk))()
        keep = []
        for i, (y, mod, pro) in enumerate(chunk):
            M = y.shape[0]
            if y.shape[1] != N or M > SMALL_MAX_ROWS:
                raise ValueError("small_bn_act_group: same width, M <= 4096")
            out = torch.empty_like(y)
            st = torch.empty(4, N, dtype=torch.float32, device=y.device) if mod is not None else None
            arr[i] = SmallBnT(_p(y).value if M else None, _p(out).value if M else None,
                              _p(mod.weight.detach()).value if mod is not None else None,
                              _p(mod.bias.detach()).value if mod is not None else None,
                              _p(mod.running_mean).value if mod is not None else None,
                              _p(mod.running_var).value if mod is not None else None,
                              _p(st).value if st is not None else None, M, int(training), int(pro.relu), float(pro.p),
                              int(pro.seed) & 0xFFFFFFFFFFFFFFFF, int(pro.site), int(pro.row_offset),
                              _p(pro.seed_dev, torch.int64).value if pro.seed_dev is not None else None)
            keep.append((out, st))
            fold = None
            if mod is not None:
                fold = BNFold(st[0], st[1], st[2], st[3], M, training)
                pro.scale, pro.shift = st[0], st[1]
            res.append((out, fold))
        check(lib.mmg_small_bn_act_group(arr, len(chunk), N, BN_MOMENTUM, BN_EPS, _stream()), "mmg_small_bn_act_group")
    return res


def small_bn_bwd_group(items):
    """items: list of (g [M,N], y [M,N], Pro, BNFold or None) -> list of (dy, dbeta or None, dgamma or None)
    (mmg_small_bn_bwd_group: the backward of small_bn_act_group, one launch)."""
    lib = _lib.load()
    res = []
    if not items:
        return res
    N = items[0][1].shape[1]
    for i0 in range(0, len(items), 8):
        chunk = items[i0:i0 + 8]
        arr = (SmallBnBwdT * len(chunk))()
        for i, (g, y, pro, fold) in enumerate(chunk):
            M = y.shape[0]
            dy = torch.empty_like(y)
            dbg = torch.empty(2, N, dtype=torch.float32, device=y.device) if fold is not None else None
            arr[i] = SmallBnBwdT(_p(g).value if M else None, _p(y).value if M else None, _p(dy).value if M else None,
                                 _p(fold.scale).value if fold is not None else None,
                                 _p(fold.shift).value if fold is not None else None,
                                 _p(fold.mean).value if fold is not None else None,
                                 _p(fold.rstd).value if fold is not None else None,
                                 _p(dbg[0]).value if dbg is not None else None, _p(dbg[1]).value if dbg is not None else None,
                                 M, int(fold.training) if fold is not None else 0, int(pro.relu), float(pro.p),
                                 int(pro.seed) & 0xFFFFFFFFFFFFFFFF, int(pro.site), int(pro.row_offset),
                                 _p(pro.seed_dev, torch.int64).value if pro.seed_dev is not None else None)
            res.append((dy, dbg[0] if dbg is not None else None, dbg[1] if dbg is not None else None))
        check(lib.mmg_small_bn_bwd_group(arr, len(chunk), N, _stream()), "mmg_small_bn_bwd_group")
    return res
