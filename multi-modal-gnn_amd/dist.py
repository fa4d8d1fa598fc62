"""Patient-axis sharding over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference is single-process and full-batch; the faithful decomposition of its ONE
graph is by contiguous patient ranges: every relation has exactly one patient endpoint
(src/graph_build.py:128-141), so no patient<->patient halo exists.  Each rank owns its patients'
embedding rows, their CSR rows of every relation and the supervision pairs of those patients; vocab
tables and all weights are replicated.  Exchanges per step (all SUM all-reduces, all tiny, latency-bound):
  * per conv layer forward : the [sum V_t, D] patient->vocab partial sums        (1 message)
  * per conv layer backward: the [sum V_t, D] grads of the transformed vocab tables (1 message)
  * per patient-axis BatchNorm: [2, D] fp64 statistics, forward and backward (Sync-BN semantics,
    required for parity with the single-device reference)
  * heads backward: [V_lab, 64] lab-side grads (+ the [64, D] first-layer weight grad)
  * end of backward: one flat bucket with every per-shard partial parameter gradient.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .data import GraphPlan, HeteroGraph, ROW_TYPE


_EAGER_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def eager_collective_stream(device_index: int):
    """The stream every EAGER RCCL collective of this package is issued on: one per device and process, created by the
    library (mmg_stream_create) and therefore outside torch's stream pools -- no hipGraph capture ever runs on it.

    Why: ProcessGroupNCCL hands every eager collective to its watchdog thread, which polls the collective's end event with
    hipEventQuery every 100 ms until it has completed.  HIP refuses that query ("operation not permitted on an event last
    recorded in a capturing stream") while the stream the event was recorded on is capturing -- also when the event itself
    was recorded BEFORE the capture began -- invalidates the capture, and the watchdog's exception ends the process
    (profiles/probes/rccl_event_cache_abort.py: eager all-reduces on S, then a capture on S within the watchdog's poll
    period aborts; the same with the capture on another stream survives).  A step warms up with eager collectives and is
    recorded right after: on one stream that is exactly the sequence.  With the eager collectives on a stream that never
    captures, no event the watchdog holds can meet a capturing stream.  Never destroyed: the watchdog may still hold events
    of it at teardown."""
    st = _EAGER_STREAMS.get(device_index)
    if st is None:
        import ctypes
        from . import _lib
        lib = _lib.load()
        out = ctypes.c_void_p()
        with torch.cuda.device(device_index):
            _lib.check(lib.mmg_stream_create(ctypes.byref(out)), "mmg_stream_create")
        st = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", device_index))
        _EAGER_STREAMS[device_index] = st
    return st


class ShardComm:
    """Collectives used by the sharded step.  ``n_calls``/``n_bytes`` are counted for the tests/bench."""

    def __init__(self, group=None, pair_ids: Optional[torch.Tensor] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.pair_ids = pair_ids          # global pair ids of this rank's pairs (keys the head dropout RNG)
        self.n_calls = 0
        self.n_bytes = 0
        self.on_collective = None         # set while a step is being captured piecewise (train.PiecewiseGraphedTrainStep)
        self._capturable = None           # capturable(): decided once, by a probe capture
        self._nccl = str(dist.get_backend(group)) == "nccl"

    # ---- recording collectives into a hipGraph
    def backend(self) -> str:
        return str(dist.get_backend(self.group))

    def capturable(self) -> bool:
        """True when the all-reduces of this group can be RECORDED into a hipGraph together with the kernels around them
        (train.PiecewiseGraphedTrainStep then replays a sharded step as one launch instead of a chain of segments with
        the collectives issued from Python between them): the RCCL backend on a HIP device, and a probe capture + replay
        of one tiny all-reduce that went through.  Decided ONCE per ShardComm, outside any other capture;
        MMG_CAPTURE_COLLECTIVES=0 / 1 overrides the probe's verdict."""
        if self._capturable is None:
            self._capturable = self._probe_capture()
        return self._capturable

    def _probe_capture(self) -> bool:
        """Every rank runs the same probe and the verdicts are combined (MIN over the group): either ALL ranks record their
        collectives or none does -- a rank that replays a graph with an all-reduce inside while another issues the chain's
        eager one would wait for it forever."""
        env = os.environ.get("MMG_CAPTURE_COLLECTIVES")
        if env is not None and env.strip() in ("0", "1"):
            return env.strip() == "1"
        if self.backend() != "nccl" or not torch.cuda.is_available():
            return False
        ok = 1.0
        dev = torch.device("cuda", torch.cuda.current_device())
        try:
            t = torch.zeros(64, device=dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.raw_all_reduce(t)                 # the communicator exists before anything is recorded
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    self.raw_all_reduce(t)
                g.replay()
                torch.cuda.synchronize()
            torch.cuda.current_stream().wait_stream(side)
            del g
        except Exception as e:                         # a backend that cannot record: the segment chain stays
            import sys
            print(f"[mmgnn.dist] collectives are not capturable here ({type(e).__name__}: {e}); using the segment chain",
                  file=sys.stderr)
            ok = 0.0
        if self.world > 1:
            verdict = torch.tensor([ok], device=dev)
            work_stream = eager_collective_stream(dev.index)
            work_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(work_stream):
                dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.group)
            torch.cuda.current_stream().wait_stream(work_stream)
            ok = float(verdict.item())
        return ok == 1.0

    # ---- teardown
    def close(self, *holders, destroy: bool = True):
        """The ONE teardown order of a sharded run (bench.py, the RCCL test, user code):
          1. `holders` (captured steps: anything with release()) drop their hipGraphs, the all-reduce tensors of their
             segment chain and their side-stream events -- what a step keeps alive that references the communicator;
          2. collect garbage, synchronise the device: no replay or collective of this process is in flight;
          3. barrier + synchronise: no OTHER rank is still inside a collective that needs this one;
          4. destroy the process group (destroy=False: the caller keeps the group for its next measurement).
        (The aborts round 3 saw around here were the watchdog meeting a capturing stream: eager_collective_stream.)"""
        for h in holders:
            rel = getattr(h, "release", None)
            if rel is not None:
                rel()
        self.on_collective = None
        import gc
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist.is_initialized():
            if self.world > 1:
                dist.barrier(group=self.group)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            if destroy:
                g, self.group = self.group, None
                dist.destroy_process_group(g)

    def raw_all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """SUM all-reduce in place.  RCCL on a HIP tensor: recorded on the current stream while that stream is capturing;
        issued on the package's eager-collective stream -- ordered after the current stream's work and before what it does
        next, two event waits -- when the current stream is a side stream (see eager_collective_stream for why not there)."""
        cur = torch.cuda.current_stream(t.device) if t.is_cuda else None
        # (the default stream can never capture: a collective issued there -- the replays of a segment chain, a training
        #  loop's own reductions -- stays where it is; only a side stream, which a step's construction records on right
        #  after its warm-up, sends its eager collectives over)
        if t.is_cuda and self._nccl and not torch.cuda.is_current_stream_capturing() and \
                cur.cuda_stream != torch.cuda.default_stream(t.device).cuda_stream:
            es = eager_collective_stream(t.device.index)
            es.wait_stream(cur)
            with torch.cuda.stream(es):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            cur.wait_stream(es)
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.on_collective is not None:
            self.on_collective(t)         # ends the current graph segment, runs the collective eagerly, starts the next
        else:
            self.raw_all_reduce(t)
        self.n_calls += 1
        self.n_bytes += t.numel() * t.element_size()
        return t

    def all_reduce_bucket(self, ts: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """One bucket for many small tensors (the end-of-backward gradient exchange): they are packed into ONE flat fp32
        tensor (a single concatenation launch), all-reduced, and handed back as VIEWS of that bucket, in order -- no copy
        back (63 separate device copies per step for this model: inside a captured step each is a graph node of its own,
        ~0.2 ms together, most of what a sharded step cost over the unsharded one on one rank)."""
        ts = list(ts)
        if not ts:
            return []
        flat = torch.cat([t.reshape(-1).float() for t in ts])
        self.all_reduce(flat)
        out, off = [], 0
        for t in ts:
            n = t.numel()
            out.append(flat[off:off + n].view(t.shape))
            off += n
        return out

    def all_reduce_list(self, ts: Sequence[torch.Tensor]):
        """all_reduce_bucket with the sums copied back into the caller's tensors (in place; None entries skipped)."""
        ts = [t for t in ts if t is not None]
        for t, v in zip(ts, self.all_reduce_bucket(ts)):
            t.copy_(v)


def partition_rows(row_weights: torch.Tensor, world: int) -> List[int]:
    """Contiguous row ranges balanced by weight (nnz): returns world+1 boundaries.
    boundaries[r] = first row of rank r; greedy on the prefix sum, every rank gets >= 1 row when possible."""
    n = int(row_weights.numel())
    if world <= 0:
        raise ValueError("world must be positive")
    csum = torch.cumsum(row_weights.to(torch.float64).cpu(), 0)
    total = float(csum[-1]) if n else 0.0
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(torch.searchsorted(csum, torch.tensor(target, dtype=torch.float64), right=True).item()) if n else 0
        b = max(b, bounds[-1] + (1 if bounds[-1] < n else 0))
        b = min(b, n - (world - r) if n >= world else n)
        b = max(b, bounds[-1])
        bounds.append(b)
    bounds.append(n)
    return bounds


def patient_weights(data) -> torch.Tensor:
    """nnz per patient over all relations (the partition's balance criterion)."""
    P = int(data[ROW_TYPE].num_nodes)
    w = None
    for et in data.edge_types:
        s, _, d = et
        ei = data[et].edge_index
        idx = ei[0] if s == ROW_TYPE else ei[1]
        c = torch.bincount(idx, minlength=P)
        w = c if w is None else w + c
    return w if w is not None else torch.zeros(P, dtype=torch.long)


def shard_graph(data, lo: int, hi: int) -> HeteroGraph:
    """The sub-graph of patients [lo, hi) with LOCAL patient indices; vocab node sets unchanged.
    Edge order inside the shard keeps the original relative order."""
    g = HeteroGraph()
    for t in data.node_types:
        g[t].num_nodes = (hi - lo) if t == ROW_TYPE else int(data[t].num_nodes)
    for et in data.edge_types:
        s, _, d = et
        ei = data[et].edge_index
        prow = 0 if s == ROW_TYPE else 1
        keep = (ei[prow] >= lo) & (ei[prow] < hi)
        sub = ei[:, keep].clone()
        sub[prow] -= lo
        g[et].edge_index = sub.contiguous()
        st = data[et]
        if "edge_attr" in st:
            g[et].edge_attr = st.edge_attr[keep]
    g.row_range = (lo, hi)
    return g


def shard_pairs(pi: torch.Tensor, li: torch.Tensor, lo: int, hi: int):
    """Pairs of patients [lo,hi): (local pi, li, global pair ids)."""
    keep = (pi >= lo) & (pi < hi)
    ids = torch.nonzero(keep).flatten()
    return (pi[keep] - lo).contiguous(), li[keep].contiguous(), ids


def shard_plan(plan: GraphPlan, comm: ShardComm, row_offset: int, n_rows_global: int) -> GraphPlan:
    """Turn a rank-local plan into a shard of the global graph: vocab in-degrees become global
    (one all-reduce of the [sum V_t] counts at setup), rows get their global offset."""
    from .data import drop_cached_plan
    drop_cached_plan(plan)               # rewritten in place below: the cache must not hand it to an unsharded caller
    done = set()
    for et, rel in plan.rels.items():
        if id(rel.col_cnt) in done:
            continue
        done.add(id(rel.col_cnt))
        cnt = rel.col_cnt.to(torch.int64)
        comm.all_reduce(cnt)
        inv = 1.0 / cnt.clamp(min=1).to(torch.float32)
        rel.inv_col.copy_(inv)
    plan.row_offset = int(row_offset)
    plan.n_rows_global = int(n_rows_global)
    return plan


def shard_model(model, comm: ShardComm):
    model._comm = comm
    return model


def shard_state(sd: Dict[str, torch.Tensor], lo: int, hi: int) -> Dict[str, torch.Tensor]:
    """Slice the patient embedding rows of a global state_dict for one shard."""
    out = dict(sd)
    k = f"embeddings.{ROW_TYPE}.weight"
    if k in out:
        out[k] = out[k][lo:hi].clone()
    return out
