"""Drop-in replacement of the reference's ``src/model.py`` hot path on MI355X.

Same call surface (``build_model``, ``HeteroRGCN.{_init_embeddings, encode_nodes, forward,
predict_lab_values}``, ``EdgeRegressionHead``, ``compute_regression_loss``), same attribute names and
the same ``state_dict`` key layout -- but every tensor op on the path is a hand-written
HIP kernel of libmmgnn.so (include/mmgnn.h) with a hand-written backward; PyTorch only owns memory,
parameters and the autograd hook.  Citations are into /root/reference/src/model.py.

Algebra used (exact up to fp32 re-association, inside the 1e-4 parity bar):
  * mean-aggregation is linear, so lin_l(mean_j x_j) = mean_j (x_j W_l^T) + b: the vocab tables
    (50..200 rows) are transformed first and the 128-d *transformed* rows are gathered (model.py:125-131);
  * for dst=patient the three lin_r terms share x_patient: x_P (sum_r W_r)^T -- one GEMM instead of four;
  * the head's Linear(2D,64) on cat[h_P[pi], h_lab[li]] splits into per-node A = h_P W[:, :D]^T and
    B = h_lab W[:, D:]^T + b, so each pair costs a 64-wide gather-add (model.py:305-333, 373-386).
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .data import GraphPlan, LAB_EDGE, ROW_TYPE, RelCSR, build_plan
from .ops import Pro

EdgeType = Tuple[str, str, str]
SITE_CONV = 16
# Vocab-side work of a layer on a side stream beside the patient-side kernels: "auto" = from 16 k patient rows up (below
# that every kernel is launch-sized and the fork / join events cost more than they hide), "off" = one stream, "on" = always.
OVERLAP_MODE = "auto"


def set_overlap(mode: str):
    """'auto' | 'off' | 'on' -- see OVERLAP_MODE (bench.py's single-stream probe pass and the tests switch it)."""
    global OVERLAP_MODE
    if mode not in ("auto", "off", "on"):
        raise ValueError(f"overlap mode must be 'auto', 'off' or 'on', got {mode!r}")
    OVERLAP_MODE = mode


# Which BatchNorm-backward statistics passes ride on the kernel that produces their upstream gradient (ops.NextBN) instead
# of a separate pass over gradient + activation: "heads" = the last conv layer's patient BatchNorm (from the heads' data-
# gradient GEMM), "conv" = the patient BatchNorm of a lower conv layer (from the gather that finishes the gradient),
# "enc2" = the encoder's second BatchNorm (from the L2-backward GEMM), "enc1" = its first, shared by the two passes of a
# training step (from the two BatchNorm-backward GEMMs above it).  "enc1" is off by default: two epilogues (with the
# register spills they cause in those two kernels) cost more than the one joint pass they replace -- x100 eICU shape, same
# box: none 2.468 ms, heads+conv+enc2 2.431 ms, all four 2.465 ms per step.
NEXT_BN_SITES = frozenset({"heads", "conv", "enc2"})
# In training the pair heads' forward saves, per visited pair, the sign bits of the first layer and the activations of the
# second (ops.pair_saved_alloc: 136 B per pair of the pair set), and the backward reads them instead of recomputing twelve
# dropout hashes, a 64 x 32 product and its epilogue per pair ("store the mask words in the forward ...
# measure both").  The recomputing backward takes that product in the forward's own order, so both variants return the same
# bits.  Used where the forward visits the supervision subset alone (a forward over ALL 4.3 M pairs would pay 68 us for the
# 30 us the backward gains).
SAVE_PAIR_STATE = True


def set_next_bn(sites):
    """An iterable of site names out of NEXT_BN_SITES' vocabulary (bench.py's A/B runs and the tests switch it)."""
    global NEXT_BN_SITES
    sites = frozenset(sites)
    if not sites <= {"heads", "conv", "enc2", "enc1"}:
        raise ValueError(f"unknown next-BatchNorm site(s): {sorted(sites - {'heads', 'conv', 'enc2', 'enc1'})}")
    NEXT_BN_SITES = sites


def _mangle(et: EdgeType) -> str:
    return "<" + "___".join(et) + ">"


# =============================================================================================
# parameter containers (names = the reference's state_dict layout)
# =============================================================================================
class _SAGEParams(nn.Module):
    """SAGEConv(in, out, aggr='mean'): lin_l with bias (neighbours), lin_r without (root)."""

    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.lin_l = nn.Linear(d_in, d_out, bias=True)
        self.lin_r = nn.Linear(d_in, d_out, bias=False)


class _HeteroConvParams(nn.Module):
    def __init__(self, edge_types: List[EdgeType], d: int):
        super().__init__()
        self.convs = nn.ModuleDict({_mangle(et): _SAGEParams(d, d) for et in edge_types})


class EdgeRegressionHead(nn.Module):
    """model.py:342-396 -- MLP 2D -> 64 -> 32 -> 1 with ReLU + Dropout."""

    def __init__(self, input_dim: int, hidden_dims: list = [64, 32], output_dim: int = 1, dropout: float = 0.2):
        super().__init__()
        if list(hidden_dims) != [64, 32] or output_dim != 1:
            raise NotImplementedError("the HIP head is specialised to hidden_dims=[64,32], output_dim=1 "
                                      "(hard-coded in the reference, model.py:161,174)")
        layers, prev = [], input_dim
        for h in hidden_dims:
            layers += [nn.Linear(prev, h), nn.ReLU(), nn.Dropout(dropout)]
            prev = h
        layers.append(nn.Linear(prev, output_dim))
        self.mlp = nn.Sequential(*layers)
        self.input_dim = input_dim
        self.p = dropout

    @torch.no_grad()
    def forward(self, edge_embeds: torch.Tensor) -> torch.Tensor:
        """Inference on explicit [n, 2D] embeddings (the reference's smoke test, model.py:654-655)."""
        n, two_d = edge_embeds.shape
        d = two_d // 2
        dev = edge_embeds.device
        w1 = self.mlp[0].weight
        a = ops.linear_fwd(edge_embeds[:, :d].contiguous(), w1[:, :d].contiguous())
        b = ops.linear_fwd(edge_embeds[:, d:].contiguous(), w1[:, d:].contiguous(), self.mlp[0].bias.detach())
        head = ops.Head(a, b, self.mlp[3].weight.detach(), self.mlp[3].bias.detach(),
                        self.mlp[6].weight.detach().reshape(-1).contiguous(), self.mlp[6].bias.detach())
        idx = torch.arange(n, dtype=torch.int32, device=dev)
        deg = torch.zeros(n, dtype=torch.int32, device=dev)
        pred = torch.empty(n, device=dev)
        p = self.p if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        ops.pair_head_fwd(head, idx, idx, deg, 1, True, p, seed, None, pred)
        return pred.unsqueeze(-1)


# =============================================================================================
# the model
# =============================================================================================
class HeteroRGCN(nn.Module):
    """model.py:33-335."""

    def __init__(self, metadata: Tuple, hidden_dim: int = 128, num_layers: int = 2, dropout: float = 0.2,
                 patient_feature_dim: int = None, use_batch_norm: bool = True, activation: str = "relu"):
        super().__init__()
        if hidden_dim not in (64, 128, 256):
            raise NotImplementedError(f"hidden_dim={hidden_dim}: the HIP kernels are built for 64, 128, 256")
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.dropout = dropout
        self.use_batch_norm = use_batch_norm
        node_types, edge_types = metadata
        self._node_types = list(node_types)
        self._edge_types = [tuple(e) for e in edge_types]

        self.embeddings = nn.ModuleDict()          # filled by _init_embeddings (lazy, model.py:180-204)
        self.embedding_dims = {}
        D = hidden_dim
        self.patient_transform = nn.Sequential(    # model.py:93-103 (containers; compute is HIP)
            nn.Linear(D, D), nn.BatchNorm1d(D), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(D, D), nn.BatchNorm1d(D), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(D, D))
        self.patient_l2_norm = F.normalize
        logging.info(f"Initialized HeteroRGCN with hidden_dim={hidden_dim}, num_layers={num_layers}")

        self.convs = nn.ModuleList()
        self.batch_norms = nn.ModuleList() if use_batch_norm else None
        for _ in range(num_layers):
            self.convs.append(_HeteroConvParams(self._edge_types, D))
            if use_batch_norm:
                self.batch_norms.append(nn.ModuleDict({t: nn.BatchNorm1d(D) for t in self._node_types}))

        if activation == "relu":                                                         # model.py:145-152
            self.activation = F.relu
        elif activation == "elu":
            self.activation = F.elu
        elif activation == "leaky_relu":
            self.activation = F.leaky_relu
        else:
            raise ValueError(f"Unknown activation: {activation}")
        # activation code of the HIP epilogues (MMG_ACT_*): conv layers only -- patient_transform and the heads are
        # nn.ReLU by construction (model.py:93-103, 373-386)
        self._act_code = {"relu": 1, "leaky_relu": 2, "elu": 3}[activation]

        self.edge_predictor = EdgeRegressionHead(2 * D, [64, 32], 1, dropout)           # model.py:159-164
        self.tabular_mlp = EdgeRegressionHead(2 * D, [64, 32], 1, dropout)              # model.py:172-177
        self.degree_threshold = 6                                                        # model.py:178

        self._comm = None            # set by dist.shard_model(): patient-axis sharding
        self._dropout_seed = None    # tests pin the dropout stream through this
        self._seed_dev = None        # int64[1] device tensor: dropout seed read at run time (hipGraph replays)
        self._pair_cache = {}
        self._last_run = None

    # ------------------------------------------------------------------------------ embeddings
    def _init_embeddings(self, data):
        """model.py:180-204 (created on the model's device; the reference leaves them on CPU)."""
        dev = next(self.parameters()).device
        for node_type in data.node_types:
            if node_type not in self.embeddings:
                num_nodes = int(data[node_type].num_nodes)
                emb = nn.Embedding(num_nodes, self.hidden_dim)
                nn.init.xavier_uniform_(emb.weight)
                self.embeddings[node_type] = emb.to(dev)
                self.embedding_dims[node_type] = num_nodes
                logging.info(f"Created embedding for {node_type}: {num_nodes} nodes")

    # ------------------------------------------------------------------------------ state_dict
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Accept both PyG key manglings for HeteroConv ('<a___b___c>' and 'a__b__c')."""
        fixed = {}
        for k, v in state_dict.items():
            parts = k.split(".")
            if len(parts) > 3 and parts[0] == "convs" and parts[2] == "convs" and not parts[3].startswith("<"):
                parts[3] = "<" + "___".join(parts[3].split("__")) + ">"
                k = ".".join(parts)
            fixed[k] = v
        return super().load_state_dict(fixed, strict=strict, assign=assign)

    # ------------------------------------------------------------------------------ public API
    def encode_nodes(self, data):
        """model.py:206-234 -> dict node_type -> [N_t, D]."""
        run = _Run(self, data)
        return run.apply("encode")

    def forward(self, data):
        """model.py:236-271 -> dict node_type -> final embeddings."""
        if len(self.embeddings) == 0:
            self._init_embeddings(data)
        run = _Run(self, data)
        return run.apply("forward")

    def predict_lab_values(self, data, patient_indices, lab_indices):
        """model.py:273-335 -> [n] predictions (degree-gated dual heads)."""
        if len(self.embeddings) == 0:
            self._init_embeddings(data)
        run = _Run(self, data)
        return run.apply("predict", patient_indices, lab_indices)

    def configure_execution(self, overlap: Optional[str] = None, next_bn=None, save_pair_state: Optional[bool] = None):
        """Per-model execution switches (None = keep / fall back to the module-level default): `overlap` 'auto' | 'off' |
        'on' (vocab-side work of a layer on a side stream), `next_bn` = the BatchNorm-backward statistics taken from
        producer epilogues (subset of {'heads', 'conv', 'enc2', 'enc1'}), `save_pair_state` = the heads' forward leaves its
        layer states for the backward.  Results do not depend on any of them; a step captured earlier keeps what it was
        recorded with."""
        ex = dict(getattr(self, "_exec", None) or {})
        if overlap is not None:
            if overlap not in ("auto", "off", "on"):
                raise ValueError(f"overlap mode must be 'auto', 'off' or 'on', got {overlap!r}")
            ex["overlap"] = overlap
        if next_bn is not None:
            sites = frozenset(next_bn)
            if not sites <= {"heads", "conv", "enc2", "enc1"}:
                raise ValueError(f"unknown next-BatchNorm site(s): {sorted(sites - {'heads', 'conv', 'enc2', 'enc1'})}")
            ex["next_bn"] = sites
        if save_pair_state is not None:
            ex["save_pair_state"] = bool(save_pair_state)
        self._exec = ex
        return self

    # pairs sorted by patient (cached: the split is static across epochs)
    def _pairs(self, pi: torch.Tensor, li: torch.Tensor, n_rows: int, pair_ids: Optional[torch.Tensor] = None,
               deg: Optional[torch.Tensor] = None, thr: int = 0):
        """(pi, li) sorted by patient -> (pi32, li32, perm, rng ids, head lists).  Cached per tensor OBJECT (the
        cache holds the tensors, so their storage cannot be recycled under a stale entry).  head lists = the
        positions served by tabular_mlp / edge_predictor (deg[pi] < thr or not; static per pair set):
        (sel_low, sel_high, counts_dev, n_low, n_high)."""
        if pair_ids is not None and pair_ids.numel() != pi.numel():
            pair_ids = None     # ids of ANOTHER pair set (a shard's train-pair ids while its validation pairs are scored)
        key = (id(pi), id(li), pi._version, li._version, n_rows, id(pair_ids), id(deg), int(thr))
        hit = self._pair_cache.get(key)
        if hit is not None:
            return hit[0]
        n = pi.numel()
        if n and (int(pi.min()) < 0 or int(pi.max()) >= n_rows):
            raise IndexError("patient_indices out of range")
        n_lab = self.embeddings["lab"].weight.shape[0] if "lab" in self.embeddings else None
        if n and n_lab is not None and (int(li.min()) < 0 or int(li.max()) >= n_lab):
            raise IndexError("lab_indices out of range")          # (the reference's final_embeds['lab'][lab_indices])
        ei = torch.stack([pi.to(torch.int64), li.to(torch.int64)]).contiguous()
        _, li_sorted, perm = ops.csr_build(ei, n_rows, 0)
        perm64 = perm.to(torch.int64)
        pi_sorted = pi.to(torch.int32)[perm64].contiguous()
        ids = perm64 if pair_ids is None else pair_ids.to(torch.int64)[perm64].contiguous()
        lists = None
        if deg is not None:
            sel_low, sel_high, counts = ops.pair_select(pi_sorted, deg, int(thr))
            n_low, n_high = counts.tolist()             # one-off sync per pair set (sizes the launches)
            # the tabular head only ever sees the low-degree patients (~1.5 %): it works on their compacted rows
            low_rows = torch.nonzero(deg < int(thr)).squeeze(1)
            low_pos = torch.full((n_rows,), -1, dtype=torch.int32, device=pi.device)
            low_pos[low_rows] = torch.arange(low_rows.numel(), dtype=torch.int32, device=pi.device)
            pi_low = low_pos[pi_sorted.to(torch.int64)].contiguous() if n else pi_sorted
            deg_low = torch.zeros(max(int(low_rows.numel()), 1), dtype=torch.int32, device=pi.device)
            lists = (sel_low[:max(n_low, 1)].clone(), sel_high[:max(n_high, 1)].clone(), counts, n_low, n_high,
                     low_rows, pi_low, deg_low, low_pos)
        out = (pi_sorted, li_sorted, perm64, ids, lists)
        if len(self._pair_cache) >= 6:
            self._pair_cache.clear()
        self._pair_cache[key] = (out, pi, li, pair_ids, deg)
        return out


# =============================================================================================
# one forward(+backward) of the hot path: manual tape over the C-ABI ops
# =============================================================================================
class _LazyAct:
    """dropout(act(BN(y))) that is NOT materialised: its consumers fold it into their load (GEMM / weight-gradient
    prologue).  Used for the last conv layer's patient activations inside predict_lab_values, which only feed the edge
    head's first linear and its weight gradient (one write + two reads of a [P, D] tensor less per step)."""

    def __init__(self, y: torch.Tensor, pro: Pro, fold=None):
        self.y, self.pro, self.fold = y, pro, fold

    @property
    def shape(self):
        return self.y.shape


BN_SUMS = "__patient_bn_sums__"   # key of a gradient dict: local BatchNorm-backward statistics that came with g[ROW_TYPE]


class _NextStats:
    """The statistics pass of a BatchNorm backward, taken from the epilogue(s) of the kernel(s) that produce its upstream
    gradient (ops.NextBN / mmg_next_bn_t): y, fold = that BatchNorm's pre-activation and fold; every producer passes
    `self.next(pro)` (its own dropout mask) and reports the sums it got back.  `sums` is complete once `n` producers --
    one per gradient that flows into the BatchNorm -- have added their share."""

    def __init__(self, y, fold):
        self.y, self.fold, self.sums, self.n = y, fold, None, 0

    def next(self, pro):
        return ops.NextBN(self.y, pro, self.fold, self.sums)

    def took(self, sums):
        self.sums = sums
        self.n += 1


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, run, mode, n_out, *params):
        ctx.run = run
        outs = run.run_forward(mode)
        ctx.mark_non_differentiable(*[o for o in outs if not o.is_floating_point()])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        grads = ctx.run.run_backward(gouts)
        return (None, None, None) + tuple(grads)


class _Run:
    def __init__(self, model: HeteroRGCN, data):
        self.m = model
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda":
            raise ops._lib.MmgError("HeteroRGCN runs on a HIP device only: call model.to('cuda') (no CPU fallback)")
        self.plan: GraphPlan = data if isinstance(data, GraphPlan) else build_plan(data, self.dev)
        self.T = model.training
        self.p = float(model.dropout) if self.T else 0.0
        self.seed_dev = model._seed_dev if self.p > 0 else None
        if self.p > 0 and self.seed_dev is None:
            self.seed = model._dropout_seed if model._dropout_seed is not None else int(
                torch.randint(0, 2 ** 62, (1,)).item())
        else:
            self.seed = 0
        self.comm = model._comm
        self.D = model.hidden_dim
        self.names = [n for n, _ in model.named_parameters()]
        self.params = {n: p for n, p in model.named_parameters()}
        self.grads: Dict[str, torch.Tensor] = {}
        self.partial = set()         # param grads that are per-shard partial sums (need the final all-reduce)
        self.tape = {}
        self._nbt = {}               # BatchNorm step counters to advance: {id(module): [buffer, increment]}
        self.pending = {}            # parameter name -> further gradient contributions, summed at the end of the backward
        self.wgrad_jobs = []         # deferred slab sums of the big weight gradients: one launch at the end of the backward
        # (sel_low, sel_high, counts) of the pairs the backward must visit, when the caller knows them in advance (a
        # captured step with a per-epoch supervision mask: train.PiecewiseGraphedTrainStep); None: selected from dpred != 0
        self.static_select = None
        # the same lists for the FORWARD of the heads: a training step whose loss reads the supervised predictions only
        # (train.py:366-370: predictions[supervision_mask]) does not need the others -- every prediction it does compute,
        # the loss and every gradient are bit for bit what the full sweep gives (per-pair arithmetic, counter RNG keyed
        # on the pair id); predictions outside the lists are 0.  Only the captured training steps set this.
        self.forward_select = None
        self.lists_ready = None      # event behind which the lists above (and the supervision mask) are valid, or None
        self.pairs = None
        self.lazy_final = False      # predict mode: the final patient activations stay folded (see _LazyAct)
        # The vocab-side work of a layer (tables of 50..200 rows: ~40 launches of a few microseconds each, a pure
        # dependency chain) runs on a side stream underneath the patient-side kernels of the same layer
        # (OVERLAP_MODE / set_overlap).  Sharded runs keep every collective on the main stream.
        # Execution switches: the MODEL's own setting where it has one (HeteroRGCN.configure_execution: two models of a
        # process can differ), else the module-level default (set_overlap / set_next_bn / SAVE_PAIR_STATE).
        ex = getattr(model, "_exec", None) or {}
        mode = ex.get("overlap") or OVERLAP_MODE
        self.next_bn_sites = ex["next_bn"] if ex.get("next_bn") is not None else NEXT_BN_SITES
        self.save_pair_state = ex["save_pair_state"] if ex.get("save_pair_state") is not None else SAVE_PAIR_STATE
        self.overlap = mode == "on" or (mode == "auto" and self.plan.n_rows >= 16384)
        if self.overlap:
            if getattr(model, "_side_stream", None) is None:
                model._side_stream = torch.cuda.Stream(device=self.dev)
            self.side = model._side_stream
        for t in self.plan.node_types:
            if t not in model.embeddings:
                raise KeyError(f"no embedding table for node type '{t}': call model._init_embeddings(data) first")
            if model.embeddings[t].weight.shape[0] != self.plan.num_nodes[t]:
                raise ValueError(f"embedding table for '{t}' has {model.embeddings[t].weight.shape[0]} rows, graph has "
                                 f"{self.plan.num_nodes[t]}")

    # ---- helpers
    def W(self, name) -> torch.Tensor:
        return self.params[name].detach()

    def acc(self, name, g, partial=False):
        if name in self.grads:
            if g is not self.grads[name]:            # (a producer that accumulated in place hands the same tensor back)
                self.pending.setdefault(name, []).append(g)      # summed by flush_grad_sums: one launch for all of them
        else:
            self.grads[name] = g
        if partial:
            self.partial.add(name)

    def flush_grad_sums(self):
        """The deferred slab sums of the weight gradients (one launch), then grads[name] = grads[name] + every later
        contribution recorded by acc(), all parameters in ONE mmg_vec_sums launch per 8.  The sum goes to a NEW tensor:
        grads[name] may be shared (dWsum serves three lin_r names) or belong to the caller (an autograd grad_output in
        encode / forward mode).  More than three further contributions chain: each level reads the previous level's
        result, so levels are separate launches (jobs of one launch run concurrently)."""
        ops.wgrad_reduce_flush(self.wgrad_jobs)
        levels: List[list] = []
        for name, more in self.pending.items():
            cur, lvl = self.grads[name], 0
            more = [m.reshape(cur.shape) for m in more]
            while more:                               # (4 sources per job: the running sum + 3)
                dst = torch.empty_like(cur, memory_format=torch.contiguous_format)
                if len(levels) <= lvl:
                    levels.append([])
                levels[lvl].append((dst, [cur] + more[:3]))
                cur, more, lvl = dst, more[3:], lvl + 1
            self.grads[name] = cur
        self.pending = {}
        for jobs in levels:
            for j0 in range(0, len(jobs), 8):
                ops.vec_sums(jobs[j0:j0 + 8])

    def allreduce(self, t):
        if self.comm is not None:
            self.comm.all_reduce(t)
        return t

    def apply(self, mode, pi=None, li=None):
        if mode == "predict":
            if pi.numel() != li.numel():
                raise ValueError("patient_indices and lab_indices differ in length")
            if pi.device != self.dev or li.device != self.dev:
                raise ops._lib.MmgError("patient_indices / lab_indices must live on the model's device")
            self.pairs = self.m._pairs(pi, li, self.plan.n_rows,
                                       getattr(self.comm, "pair_ids", None) if self.comm else None,
                                       self.plan.lab_deg, int(self.m.degree_threshold))
            self.n_pairs = pi.numel()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.params.values())
        self.need_grad = need_grad
        self.m._last_run = self if need_grad else None      # (tests drive run_backward by hand through this)
        plist = [self.params[n] for n in self.names]
        if need_grad:
            outs = _StepFn.apply(self, mode, 0, *plist)
        else:
            outs = tuple(self.run_forward(mode))
        if mode == "predict":
            return outs[0]
        return {t: o for t, o in zip(self.out_types, outs)}

    # ======================================================================== forward driver
    def _bump_counters(self):
        ents = [e for e in self._nbt.values() if e[1]]
        if ents:                                         # every num_batches_tracked buffer in ONE launch
            ops.counters_add([b for b, _ in ents], [i for _, i in ents])
        self._nbt = {}

    def run_forward(self, mode):
        outs = self._run_forward(mode)
        self._bump_counters()
        return outs

    def _run_forward(self, mode):
        T, plan = self.T, self.plan
        if mode == "encode":
            enc = self.enc_fwd(0, 1)
            x = self.enc_dict(enc)
            self.tape["mode"] = ("encode", enc)
            self.out_types = list(x.keys())
            return [x[t] if t == ROW_TYPE else x[t].clone() for t in self.out_types]
        if mode == "forward":
            enc = self.enc_fwd(1, 1)
            x, layers = self.layers_fwd(self.enc_dict(enc))
            self.tape["mode"] = ("forward", enc, layers)
            self.out_types = list(x.keys())
            return [x[t] for t in self.out_types]
        self.lazy_final = True
        # predict: encode_nodes runs twice (model.py:294 and :301->251).  The two passes differ only by
        # their dropout masks, so with p == 0 (or eval) one pass is computed and BN running stats are
        # advanced twice.
        if self.p > 0:
            first = self.enc_first(2)            # shared by both passes (no dropout before the first BatchNorm)
            m0, m1 = self.enc_mid(0, first), self.enc_mid(1, first)
            if self.comm is not None and self.T:     # sharded: the statistics of both passes in ONE all-reduce
                both = torch.stack([m0[2], m1[2]])
                self.allreduce(both)
                m0[2], m1[2], m0[3], m1[3] = both[0], both[1], True, True
            enc0 = self.enc_fwd(0, 1, first, rows=self.pairs[4][5], mid=m0)   # feeds the tabular head only: low-degree rows
            enc0["row_pos"] = self.pairs[4][8]
            enc1 = self.enc_fwd(1, 1, first, mid=m1)
        else:
            enc0 = enc1 = self.enc_fwd(0, 2)
        init = self.enc_dict(enc0)
        self._init_compact = enc0.get("rows") is not None
        fin, layers = self.layers_fwd(self.enc_dict(enc1))
        pred, hrec = self.heads_fwd(init, fin)
        self.tape["mode"] = ("predict", enc0, enc1, layers, hrec)
        return [pred]

    def run_backward(self, gouts):
        mode = self.tape["mode"]
        D = self.D
        if mode[0] == "encode":
            enc = mode[1]
            g = {t: go for t, go in zip(self.out_types, gouts)}
            self.enc_bwd(enc, g.get(ROW_TYPE))
            for t in self.out_types:
                if t != ROW_TYPE and g[t] is not None:
                    self.acc(f"embeddings.{t}.weight", g[t].contiguous())
        elif mode[0] == "forward":
            _, enc, layers = mode
            g = {t: (go.contiguous() if go is not None else None) for t, go in zip(self.out_types, gouts)}
            g = self.layers_bwd(layers, g)
            self.enc_bwd(enc, g.get(ROW_TYPE))
            for t, gt in g.items():
                if t not in (ROW_TYPE, BN_SUMS) and gt is not None:
                    self.acc(f"embeddings.{t}.weight", gt)
        else:
            _, enc0, enc1, layers, hrec = mode
            dpred = gouts[0].contiguous()
            g_init, g_fin = self.heads_bwd(hrec, dpred)
            g = self.layers_bwd(layers, g_fin)
            gi = g_init.get(ROW_TYPE)            # tabular head: (row ids, gradient rows) of the low-degree patients

            def dense(rows_grads):
                if rows_grads is None:
                    return None
                d = torch.zeros(self.plan.n_rows, D, device=self.dev)
                d[rows_grads[0]] = rows_grads[1]
                return d

            if enc0 is enc1:
                gp = g.get(ROW_TYPE)
                if gp is None:
                    tot = dense(gi)
                else:
                    tot = gp if gi is None else gp.index_add_(0, gi[0], gi[1])
                self.enc_bwd(enc0, tot)
            else:
                a1 = self.enc_bwd_a(enc1, g.get(ROW_TYPE))
                a0 = self.enc_bwd_a(enc0, gi if enc0.get("rows") is not None else dense(gi))
                if self.comm is not None:        # sharded: the BatchNorm statistics of both passes in ONE all-reduce
                    live = [a for a in (a1, a0) if a is not None]
                    if len(live) == 2:
                        both = torch.stack([a1[1], a0[1]])
                        self.allreduce(both)
                        a1, a0 = (a1[0], both[0]), (a0[0], both[1])
                    elif live:
                        self.allreduce(live[0][1])
                nstats = _NextStats(enc1["z1"], enc1["f1"])      # the shared first BatchNorm: both passes add their share
                self.enc_bwd_shared(enc1, self.enc_bwd_b(enc1, a1, nstats), enc0, self.enc_bwd_b(enc0, a0, nstats), nstats)
            for t, gt in g.items():
                if t not in (ROW_TYPE, BN_SUMS) and gt is not None:
                    self.acc(f"embeddings.{t}.weight", gt)
            if g_init.get("lab") is not None:
                self.acc("embeddings.lab.weight", g_init["lab"])
        self.flush_grad_sums()
        if self.comm is not None and self.partial:
            names = sorted(self.partial)
            bucket = getattr(self.comm, "all_reduce_bucket", None)
            if bucket is not None:           # the summed gradients are views of the one bucket: nothing is copied back
                for n, g in zip(names, bucket([self.grads[n] for n in names])):
                    self.grads[n] = g
            else:
                self.comm.all_reduce_list([self.grads[n] for n in names])
        out = []
        for n in self.names:
            g = self.grads.pop(n, None)       # drop our reference: autograd can then adopt the tensor as .grad (no clone)
            if g is not None and g.shape != self.params[n].shape:
                g = g.reshape(self.params[n].shape)
            out.append(g)
        return out

    # ======================================================================== encode_nodes
    def bn_spec(self, mod: nn.BatchNorm1d, n_updates: int, rows: int, sharded: bool):
        """(gamma, beta, running_mean, running_var, n_updates) for a producer that folds the training-mode BatchNorm of its
        output in the launch that sums its statistics -- or None where that is not possible (eval mode; a patient-sharded
        tensor, whose sums are all-reduced first; a single row, which raises like torch)."""
        if not self.T or rows <= 1 or (sharded and self.comm is not None):
            return None
        return (mod.weight.detach(), mod.bias.detach(), mod.running_mean, mod.running_var, int(n_updates))

    def count_bn(self, mod: nn.BatchNorm1d, n_updates: int):
        ent = self._nbt.setdefault(id(mod), [mod.num_batches_tracked, 0])           # bumped once, together
        ent[1] += int(n_updates)

    def bn_fold(self, y, mod: nn.BatchNorm1d, n_updates=1, sharded=False, sums=None, reduced=False) -> ops.BNFold:
        """Batch statistics (train) or running statistics (eval) folded to scale/shift.  sums: the column sums of
        y and y^2 when the producing kernel already took them (reduced: already summed over the shards)."""
        count = y.shape[0]
        if isinstance(sums, ops.BNFold):         # the producer already folded it (bn_spec)
            self.count_bn(mod, n_updates)
            return sums
        if self.T:
            if sums is None:
                sums = ops.col_reduce2(y)
            if sharded and self.comm is not None:
                if not reduced:
                    self.allreduce(sums)
                count = self.plan.n_rows_global
            if count <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {list(y.shape)}")
            fold = ops.bn_finalize(sums, count, mod.weight.detach(), mod.bias.detach(), mod.running_mean,
                                   mod.running_var, True, n_updates)
            ent = self._nbt.setdefault(id(mod), [mod.num_batches_tracked, 0])           # bumped once, together
            ent[1] += int(n_updates)
            return fold
        return ops.bn_finalize(None, count, mod.weight.detach(), mod.bias.detach(), mod.running_mean, mod.running_var,
                               False, 0)

    def enc_first(self, n_updates):
        """First linear + BatchNorm statistics of patient_transform.  No dropout sits in front of them, so the two
        encode_nodes passes of a training step (F7) share them bit for bit: computed once, the BatchNorm running
        statistics advanced n_updates times."""
        pt = self.m.patient_transform
        E = self.W(f"embeddings.{ROW_TYPE}.weight")
        # the batch statistics of z1 / z2 come out of the GEMM epilogue (training mode)
        spec = self.bn_spec(pt[1], n_updates, E.shape[0], True)
        if spec is not None:                     # ... and so does the BatchNorm fold (single GPU)
            z1, _, s1 = ops.linear_fwd(E, pt[0].weight.detach(), pt[0].bias.detach(), bn=spec)
        else:
            z1, s1 = ops.linear_fwd(E, pt[0].weight.detach(), pt[0].bias.detach(), with_stats=True) if self.T else \
                (ops.linear_fwd(E, pt[0].weight.detach(), pt[0].bias.detach()), None)
        f1 = self.bn_fold(z1, pt[1], n_updates, sharded=True, sums=s1)
        return E, z1, f1

    def enc_mid(self, call, first, n_updates=1):
        """Second linear of pass `call` with the BatchNorm statistics of its output (not yet summed over the shards; on a
        single GPU already folded: a BNFold instead of the sums)."""
        pt = self.m.patient_transform
        E, z1, f1 = first
        pro1 = Pro(f1.scale, f1.shift, True, self.p, self.seed, 2 * call, self.plan.row_offset, self.seed_dev)
        spec = self.bn_spec(pt[5], n_updates, z1.shape[0], True)
        if spec is not None:
            z2, _, s2 = ops.linear_fwd(z1, pt[4].weight.detach(), pt[4].bias.detach(), pro=pro1, bn=spec)
        else:
            z2, s2 = ops.linear_fwd(z1, pt[4].weight.detach(), pt[4].bias.detach(), pro=pro1, with_stats=True) if self.T else \
                (ops.linear_fwd(z1, pt[4].weight.detach(), pt[4].bias.detach(), pro=pro1), None)
        return [pro1, z2, s2, False]

    def enc_fwd(self, call, n_updates, first=None, rows=None, mid=None):
        """rows (int64 ids): everything after the last BatchNorm -- third linear, L2 norm -- is evaluated for these rows
        only and x0 / rn are compact [len(rows), .] (the first encode_nodes pass of a training step only feeds the
        tabular head, which only sees the low-degree patients; its BatchNorm statistics still need every row)."""
        pt = self.m.patient_transform
        off = self.plan.row_offset
        if first is None:
            first = self.enc_first(n_updates)
        E, z1, f1 = first
        pro1, z2, s2, reduced = mid if mid is not None else self.enc_mid(call, first, n_updates)
        f2 = self.bn_fold(z2, pt[5], n_updates, sharded=True, sums=s2, reduced=reduced)
        pro2 = Pro(f2.scale, f2.shift, True, self.p, self.seed, 2 * call + 1, off, self.seed_dev)
        if rows is not None:
            act = ops.affine_act_drop_rows(z2, pro2, rows)       # the dropout masks of the ORIGINAL rows
            x0, rn = ops.linear_l2norm_fwd(act, pt[8].weight.detach(), pt[8].bias.detach())
            return dict(E=E, z1=z1, z2=z2, x0=x0, rn=rn, f1=f1, f2=f2, pro1=pro1, pro2=pro2, rows=rows, act=act)
        x0, rn = ops.linear_l2norm_fwd(z2, pt[8].weight.detach(), pt[8].bias.detach(), pro=pro2)    # third linear + L2 norm
        return dict(E=E, z1=z1, z2=z2, x0=x0, rn=rn, f1=f1, f2=f2, pro1=pro1, pro2=pro2)

    def enc_dict(self, enc):
        x = {}
        for t in self.plan.node_types:
            x[t] = enc["x0"] if t == ROW_TYPE else self.W(f"embeddings.{t}.weight")
        return x

    def bn_bwd_sums(self, g, y, pro: Pro, fold: Optional[ops.BNFold], sharded: bool):
        """first half of bn_bwd: the column statistics (all-reduced over the shards); None without batch norm."""
        if fold is None:
            return None
        sums = ops.bn_bwd_stats(g, y, pro, fold)
        if sharded and self.comm is not None:
            self.allreduce(sums)
        return sums

    def bn_bwd(self, g, y, pro: Pro, fold: Optional[ops.BNFold], bn_prefix: Optional[str], sharded: bool, sums=None,
               add_into=None):
        """grad wrt the pre-BN tensor y of  x' = dropout(relu(BN(y)));  accumulates d gamma / d beta.
        add_into: add the result to this tensor (inside the kernel) instead of returning a new one."""
        acc = add_into is not None
        if fold is None:        # no batch norm: relu/dropout only
            return ops.bn_bwd_apply(g, y, pro, None, out=add_into, accumulate=acc)
        if sums is None:
            sums = self.bn_bwd_sums(g, y, pro, fold, sharded)
        N = y.shape[1]
        dbg = torch.empty(2, N, device=y.device)       # d beta | d gamma, written by the apply kernel
        if fold.training:
            dy = ops.bn_bwd_apply(g, y, pro, fold, sums, fold.count, dbg[0], dbg[1], out=add_into, accumulate=acc)
        else:
            dbg.copy_(sums)
            dy = ops.bn_bwd_apply(g, y, pro, fold, out=add_into, accumulate=acc)
        self.acc(bn_prefix + ".bias", dbg[0])
        self.acc(bn_prefix + ".weight", dbg[1])
        return dy

    def bn_lin_bwd(self, g, y, pro: Pro, fold: Optional[ops.BNFold], bn_prefix: Optional[str], sharded: bool, sums, W,
                   nxt: Optional[Tuple["_NextStats", Pro]] = None):
        """bn_bwd followed by the data gradient  dz @ W  through the linear in front of that BatchNorm, as ONE kernel
        (mmg_linear_bnbwd: g and y are read once, dz is written once for the weight gradient).  -> (dz, dx), or None where
        the fused kernel does not apply (the caller then runs bn_bwd and the GEMM).
        nxt = (stats holder, prologue) of the BatchNorm below, whose backward consumes dx: its statistics ride along."""
        if g is None or pro.relu not in (0, 1) or not ops.linear_bnbwd_supported(y.shape[0], W.shape[1], y.shape[1]):
            return None
        nb = nxt[0].next(nxt[1]) if nxt is not None else None

        def done(out):
            if nb is not None:
                nxt[0].took(out[2])
            return out[0], out[1]

        if fold is None:
            return done(ops.linear_bnbwd(g, y, pro, None, W, next_bn=nb))
        if sums is None:
            sums = self.bn_bwd_sums(g, y, pro, fold, sharded)
        dbg = torch.empty(2, y.shape[1], device=y.device)       # d beta | d gamma, written by the kernel
        if fold.training:
            out = ops.linear_bnbwd(g, y, pro, fold, W, sums, fold.count, dbg[0], dbg[1], next_bn=nb)
        else:
            dbg.copy_(sums)
            out = ops.linear_bnbwd(g, y, pro, fold, W, next_bn=nb)
        self.acc(bn_prefix + ".bias", dbg[0])
        self.acc(bn_prefix + ".weight", dbg[1])
        return done(out)

    def lin_bwd(self, dy, x, pro, wname, bname, need_dx=True, partial=False, dx_into=None):
        """grads of  y = pro(x) W^T + b.  dx_into: accumulate dX into this tensor (inside the GEMM) instead of a new one."""
        gw = self.grads.get(wname)       # a second contribution (the encoder runs twice) accumulates inside the kernel
        if bname is not None:            # the bias gradient (column sums of dy) comes out of the same pass over dy
            gb = self.grads.get(bname)
            if gw is not None and gb is not None:
                ops.linear_wgrad(dy, x, pro, out=gw, accumulate=True, with_bias=True, bias_out=gb, defer=self.wgrad_jobs)
                dW, db = gw, gb
            else:
                dW, db = ops.linear_wgrad(dy, x, pro, with_bias=True, defer=self.wgrad_jobs)
            self.acc(wname, dW, partial)
            self.acc(bname, db, partial)
        else:
            self.acc(wname, ops.linear_wgrad(dy, x, pro, out=gw, accumulate=gw is not None, defer=self.wgrad_jobs), partial)
        if need_dx:
            if dx_into is not None:
                return ops.linear_fwd(dy, self.W(wname), w_kn=True, out=dx_into, accumulate=True)
            return ops.linear_fwd(dy, self.W(wname), w_kn=True)          # dX = dY . W, W read in place
        return None

    def enc_bwd_a(self, enc, g_x0):
        """Backward of one encoder pass down to the statistics of its last BatchNorm (local sums, not yet all-reduced):
        -> (upstream gradient of that BatchNorm [dense, or the listed rows], fp64 sums) or None."""
        if g_x0 is None:
            return None
        pt = "patient_transform"
        if enc.get("rows") is not None:      # compact tail: g_x0 = (row ids, gradient rows)
            rows, g_rows = g_x0
            # L2-norm backward inside the data-gradient GEMM of the third linear; the weight gradient reads dz3 afterwards
            dz3, g = ops.linear_l2bwd(g_rows.contiguous(), enc["x0"], enc["rn"], self.W(f"{pt}.8.weight"))
            self.lin_bwd(dz3, enc["act"], None, f"{pt}.8.weight", f"{pt}.8.bias", need_dx=False, partial=True)
            sums = ops.bn_bwd_stats_rows(g, enc["z2"], rows, enc["pro2"], enc["f2"])
        else:
            # the statistics of the second BatchNorm's backward come out of the epilogue of the GEMM that produces its
            # upstream gradient (the separate pass read g and z2 once more)
            if "enc2" in self.next_bn_sites:
                dz3, g, sums = ops.linear_l2bwd(g_x0.contiguous(), enc["x0"], enc["rn"], self.W(f"{pt}.8.weight"),
                                                next_bn=ops.NextBN(enc["z2"], enc["pro2"], enc["f2"]))
            else:
                dz3, g = ops.linear_l2bwd(g_x0.contiguous(), enc["x0"], enc["rn"], self.W(f"{pt}.8.weight"))
                sums = ops.bn_bwd_stats(g, enc["z2"], enc["pro2"], enc["f2"])
            self.lin_bwd(dz3, enc["z2"], enc["pro2"], f"{pt}.8.weight", f"{pt}.8.bias", need_dx=False, partial=True)
        return g, sums

    def enc_bwd_b(self, enc, state, nstats: Optional["_NextStats"] = None):
        """...and from the (all-reduced) statistics on to the upstream gradient of the first BatchNorm.
        nstats: holder of that BatchNorm's backward statistics; a fused producer adds its share (this pass's mask)."""
        if state is None:
            return None
        pt = "patient_transform"
        g, sums = state
        y, pro, fold = enc["z2"], enc["pro2"], enc["f2"]
        nxt = (nstats, enc["pro1"]) if nstats is not None and nstats.fold is not None and "enc1" in self.next_bn_sites else None
        if enc.get("rows") is not None:
            # an upstream gradient that is zero outside the listed rows (training statistics): the dense apply pass never
            # reads a gradient tensor, the listed rows are patched afterwards
            dbg = torch.empty(2, y.shape[1], device=y.device)   # d beta | d gamma, written by the apply kernel
            W4 = self.W(f"{pt}.4.weight")
            if enc.get("row_pos") is not None and pro.relu in (0, 1) and fold.training and \
                    ops.linear_bnbwd2_supported(y.shape[0], W4.shape[1], y.shape[1]):
                # dense pass, row patch and the data-gradient GEMM of the second linear in ONE kernel
                if nxt is not None:
                    dz2, dx, nsums = ops.linear_bnbwd_rows(g, enc["row_pos"], y, pro, fold, W4, sums, fold.count, dbg[0], dbg[1],
                                                           next_bn=nstats.next(enc["pro1"]))
                    nstats.took(nsums)
                else:
                    dz2, dx = ops.linear_bnbwd_rows(g, enc["row_pos"], y, pro, fold, W4, sums, fold.count, dbg[0], dbg[1])
                self.acc(f"{pt}.5.bias", dbg[0])
                self.acc(f"{pt}.5.weight", dbg[1])
                self.lin_bwd(dz2, enc["z1"], enc["pro1"], f"{pt}.4.weight", f"{pt}.4.bias", need_dx=False, partial=True)
                return dx
            dz2 = ops.bn_bwd_apply(None, y, pro, fold, sums, fold.count, dbg[0], dbg[1])
            ops.bn_bwd_apply_rows(g, y, enc["rows"], pro, dz2)
            self.acc(f"{pt}.5.bias", dbg[0])
            self.acc(f"{pt}.5.weight", dbg[1])
        else:
            fused = self.bn_lin_bwd(g, y, pro, fold, f"{pt}.5", True, sums, self.W(f"{pt}.4.weight"), nxt=nxt)
            if fused is not None:                # BatchNorm backward inside the data-gradient GEMM of the second linear
                dz2, dx = fused
                self.lin_bwd(dz2, enc["z1"], enc["pro1"], f"{pt}.4.weight", f"{pt}.4.bias", need_dx=False, partial=True)
                return dx
            dz2 = self.bn_bwd(g, y, pro, fold, f"{pt}.5", sharded=True, sums=sums)
        return self.lin_bwd(dz2, enc["z1"], enc["pro1"], f"{pt}.4.weight", f"{pt}.4.bias", partial=True)

    def enc_bwd(self, enc, g_x0, upto_bn1=False):
        """upto_bn1: stop in front of the first BatchNorm and return its upstream gradient (the two passes of a training
        step share that BatchNorm and the linear in front of it: enc_bwd_shared differentiates them once for both)."""
        state = self.enc_bwd_a(enc, g_x0)
        if state is None:
            return None
        if self.comm is not None:
            self.allreduce(state[1])
        nstats = _NextStats(enc["z1"], enc["f1"]) if not upto_bn1 else None
        g = self.enc_bwd_b(enc, state, nstats)
        if upto_bn1:
            return g
        pt = "patient_transform"
        sums = nstats.sums if nstats.n == 1 else None        # from the epilogue of the kernel that produced g
        if sums is not None and self.comm is not None:
            self.allreduce(sums)
        dz1 = self.bn_bwd(g, enc["z1"], enc["pro1"], enc["f1"], f"{pt}.1", sharded=True, sums=sums)
        self.enc_bwd_first(enc, dz1)
        return None

    def enc_bwd_shared(self, enc_a, g_a, enc_b, g_b, nstats: Optional["_NextStats"] = None):
        """First BatchNorm + first linear of two passes that share them (same z1, same statistics, own dropout masks):
        g_out = g_out(g_a; pro1_a) + g_out(g_b; pro1_b) -- one statistics pass, one apply pass, one weight / data gradient.
        nstats: the statistics the producers of g_a / g_b summed in their epilogues (complete when every gradient that is
        present added its share; the separate pass over g_a, g_b and z1 runs otherwise)."""
        pt = "patient_transform"
        if g_a is None and g_b is None:
            return
        n_live = (g_a is not None) + (g_b is not None)
        pre = nstats.sums if nstats is not None and nstats.n == n_live else None
        if g_a is None or g_b is None:
            enc, g = (enc_a, g_a) if g_b is None else (enc_b, g_b)
            if pre is not None and self.comm is not None:
                self.allreduce(pre)
            dz1 = self.bn_bwd(g, enc["z1"], enc["pro1"], enc["f1"], f"{pt}.1", sharded=True, sums=pre)
        else:
            y, fold = enc_a["z1"], enc_a["f1"]
            sums = pre if pre is not None else ops.bn_bwd_stats2(g_a, g_b, y, enc_a["pro1"], enc_b["pro1"], fold)
            if self.comm is not None:
                self.allreduce(sums)
            dbg = torch.empty(2, y.shape[1], device=y.device)
            ename, W0 = f"embeddings.{ROW_TYPE}.weight", self.W(f"{pt}.0.weight")
            if self.params[ename].requires_grad and ename not in self.grads and \
                    ops.linear_bnbwd2_supported(y.shape[0], W0.shape[1], y.shape[1]):
                # the joint BatchNorm backward inside the data-gradient GEMM of the first linear (dE)
                dz1, dE = ops.linear_bnbwd2(g_a, g_b, y, enc_a["pro1"], enc_b["pro1"], fold, W0, sums, fold.count,
                                            dbg[0], dbg[1])
                self.acc(f"{pt}.1.bias", dbg[0])
                self.acc(f"{pt}.1.weight", dbg[1])
                self.lin_bwd(dz1, enc_a["E"], None, f"{pt}.0.weight", f"{pt}.0.bias", need_dx=False, partial=True)
                self.acc(ename, dE)
                return
            dz1 = ops.bn_bwd_apply2(g_a, g_b, y, enc_a["pro1"], enc_b["pro1"], fold, sums, fold.count, dbg[0], dbg[1])
            self.acc(f"{pt}.1.bias", dbg[0])
            self.acc(f"{pt}.1.weight", dbg[1])
        self.enc_bwd_first(enc_a, dz1)

    def enc_bwd_first(self, enc, dz1):
        pt = "patient_transform"
        ename = f"embeddings.{ROW_TYPE}.weight"
        need_dE = self.params[ename].requires_grad
        # the second encoder pass (dropout: encode_nodes runs twice, F7) adds its dE inside the GEMM
        dE = self.lin_bwd(dz1, enc["E"], None, f"{pt}.0.weight", f"{pt}.0.bias", need_dx=need_dE, partial=True,
                          dx_into=self.grads.get(ename))
        if dE is not None and ename not in self.grads:
            self.acc(ename, dE)

    # ======================================================================== HeteroConv layers
    def layers_fwd(self, x):
        recs = []
        for l in range(self.m.num_layers):
            x, rec = self.layer_fwd(l, x)
            recs.append(rec)
        return x, recs

    def conv_name(self, l, et):
        return f"convs.{l}.convs.{_mangle(et)}"

    def layer_fwd(self, l, x):
        plan, D, P = self.plan, self.D, self.plan.n_rows
        xP = x.get(ROW_TYPE)
        rin = [r for r in plan.rels_into_patient() if r.other in x and xP is not None]
        rout = [r for r in plan.rels_from_patient() if r.other in x and xP is not None]
        y: Dict[str, torch.Tensor] = {}
        rec = dict(x=x, rin=rin, rout=rout)
        last = l == self.m.num_layers - 1
        p = 0.0 if last else self.p
        out, folds, pros = {}, {}, {}

        def bn_act(t):       # per-type BN -> ReLU -> Dropout (model.py:258-269)
            ti = plan.node_types.index(t)
            sharded = t == ROW_TYPE
            fold = self.bn_fold(y[t], self.m.batch_norms[l][t], 1, sharded,
                                sums=rec.get("ysums") if t == ROW_TYPE else None) if self.m.use_batch_norm else None
            pro = Pro(fold.scale if fold else None, fold.shift if fold else None, self.m._act_code, p, self.seed,
                      SITE_CONV + 8 * l + ti, plan.row_offset if sharded else 0, self.seed_dev)
            if last and t == ROW_TYPE and self.lazy_final and self.m._act_code == 1:
                out[t] = _LazyAct(y[t], pro, fold)               # consumed through the heads' GEMM prologues
            else:
                out[t] = ops.affine_act_drop(y[t], pro)
            folds[t], pros[t] = fold, pro

        names = [self.conv_name(l, r.edge_type) for r in rin]
        tables = []

        def vocab_tables():  # T_v = x_v W_l^T: the transformed vocab rows the patient-side gather reads (ONE launch)
            if all(x[r.other].shape[0] <= ops.SMALL_MAX_ROWS for r in rin):
                tables.extend(ops.small_fwd_group([ops.SmallFwd(x[r.other], self.W(nme + ".lin_l.weight"))
                                                   for r, nme in zip(rin, names)]))
                return
            for r, nme in zip(rin, names):
                tables.append(ops.linear_fwd(x[r.other], self.W(nme + ".lin_l.weight")))

        scat = {}

        def vocab_scatter():
            # ---- dst = vocab type v: y_v = mean_scatter(x_P) W_l^T + b + x_v W_r^T   (summed over relations into v)
            if not rout:
                return
            aggs, rels, off = [], [], 0
            buf = torch.empty(sum(r.n_cols for r in rout), D, device=self.dev)   # one buffer = one all-reduce
            for r in rout:
                agg = buf[off:off + r.n_cols]
                off += r.n_cols
                rels.append(ops.Rel(r.rowptr, r.col, r.n_cols, colscale=r.inv_col, out=agg, simple=r.simple, mask_t=r.mask_t))
                aggs.append(agg)
            ops.scatter_rows(rels, P, D, xP)
            self.allreduce(buf)                          # partial sums over patient shards
            scat["aggs"] = aggs

        def vocab_small():
            if rout:
                others = [r.other for r in rout]
                if len(set(others)) == len(others) and all(r.n_cols <= ops.SMALL_MAX_ROWS for r in rout):
                    # every vocab type has ONE incoming relation: y_v = agg W_l^T + b + x_v W_r^T is a two-term problem,
                    # all of them in one launch
                    outs = ops.small_fwd_group([
                        ops.SmallFwd(agg, self.W(self.conv_name(l, r.edge_type) + ".lin_l.weight"),
                                     bias=self.W(self.conv_name(l, r.edge_type) + ".lin_l.bias"), x2=x[r.other],
                                     W2=self.W(self.conv_name(l, r.edge_type) + ".lin_r.weight"))
                        for r, agg in zip(rout, scat["aggs"])])
                    for r, o in zip(rout, outs):
                        y[r.other] = o
                else:
                    for r, agg in zip(rout, scat["aggs"]):
                        nme = self.conv_name(l, r.edge_type)
                        first = r.other not in y
                        y[r.other] = ops.linear_fwd(agg, self.W(nme + ".lin_l.weight"), self.W(nme + ".lin_l.bias"),
                                                    out=None if first else y[r.other], accumulate=not first)
                        ops.linear_fwd(x[r.other], self.W(nme + ".lin_r.weight"), out=y[r.other], accumulate=True)
                rec["aggs"] = scat["aggs"]
            vts = [t for t in plan.node_types if t != ROW_TYPE and t in y]
            if vts and all(y[t].shape[0] <= ops.SMALL_MAX_ROWS for t in vts) and \
                    not (self.T and self.m.use_batch_norm and any(y[t].shape[0] <= 1 for t in vts)):
                # BatchNorm statistics + fold + running update + activation + dropout of every vocab type: ONE launch
                items = []
                for t in vts:
                    ti = plan.node_types.index(t)
                    mod = self.m.batch_norms[l][t] if self.m.use_batch_norm else None
                    items.append((y[t], mod, Pro(None, None, self.m._act_code, p, self.seed, SITE_CONV + 8 * l + ti, 0,
                                                 self.seed_dev)))
                for t, (o, fold), it in zip(vts, ops.small_bn_act_group(items, self.T), items):
                    out[t], folds[t], pros[t] = o, fold, it[2]
                    if self.T and it[1] is not None:
                        ent = self._nbt.setdefault(id(it[1]), [it[1].num_batches_tracked, 0])
                        ent[1] += 1
            else:
                for t in vts:
                    bn_act(t)

        def patient_gather(wait_tables):
            # ---- dst = patient: y_P = x_P (sum_r W_r)^T + sum_r b_r + sum_r mean_gather(x_v W_l^T)
            if not rin:
                return
            if len(names) == 1:
                Wsum, bsum = self.W(names[0] + ".lin_r.weight"), self.W(names[0] + ".lin_l.bias")
            elif len(names) <= 4:                # the lin_r weights / lin_l biases that share x_patient: one launch
                Wsum = torch.empty(D, D, device=self.dev)
                bsum = torch.empty(D, device=self.dev)
                ops.vec_sums([(Wsum, [self.W(n_ + ".lin_r.weight") for n_ in names]),
                              (bsum, [self.W(n_ + ".lin_l.bias") for n_ in names])])
            else:
                Wsum = self.W(names[0] + ".lin_r.weight")
                bsum = self.W(names[0] + ".lin_l.bias")
                for nme in names[1:]:
                    Wsum = Wsum + self.W(nme + ".lin_r.weight")
                    bsum = bsum + self.W(nme + ".lin_l.bias")
            yP = ops.linear_fwd(xP, Wsum.contiguous(), bsum.contiguous())
            wait_tables()
            rels = [ops.Rel(r.rowptr, r.col, r.n_cols, rowscale=r.inv_row, table=Tv, simple=r.simple, mask_r=r.mask_r)
                    for r, Tv in zip(rin, tables)]
            ysums = None
            spec = self.bn_spec(self.m.batch_norms[l][ROW_TYPE], 1, P, True) if self.m.use_batch_norm else None
            if spec is not None:                     # BatchNorm statistics AND fold of y_P from the gather's launches
                _, _, ysums = ops.gather_rows(rels, P, D, yP, accumulate=True, bn=spec)
            elif self.T and self.m.use_batch_norm and P > 0:       # BatchNorm statistics of y_P from the gather epilogue
                _, ysums = ops.gather_rows(rels, P, D, yP, accumulate=True, with_stats=True)
            else:
                ops.gather_rows(rels, P, D, yP, accumulate=True)
            y[ROW_TYPE] = yP
            rec["Wsum"] = Wsum
            rec["ysums"] = ysums

        def patient_bn():
            if rin:
                bn_act(ROW_TYPE)                             # (sharded: all-reduce of the column sums inside)

        if self.overlap and self.comm is None:
            # single GPU: the whole vocab path, its scatter included, runs beside the patient path
            main = torch.cuda.current_stream()
            self.side.wait_stream(main)
            ev = torch.cuda.Event()
            with torch.cuda.stream(self.side):
                vocab_tables()
                ev.record(self.side)
                vocab_scatter()
                vocab_small()
            patient_gather(lambda: main.wait_event(ev))
            patient_bn()
            main.wait_stream(self.side)
        elif self.overlap:
            # sharded: collectives stay on the main stream (they cut the hipGraph segments, so every fork is joined
            # before the next one); the small vocab chains run beside the patient GEMM + gather between two of them
            main = torch.cuda.current_stream()
            vocab_tables()
            vocab_scatter()
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                vocab_small()
            patient_gather(lambda: None)
            main.wait_stream(self.side)
            patient_bn()
        else:
            vocab_tables()
            patient_gather(lambda: None)
            patient_bn()
            vocab_scatter()
            vocab_small()
        rec.update(y=y, folds=folds, pros=pros, l=l, tables=tables)
        return out, rec

    def layers_bwd(self, recs, g):
        for i in range(len(recs) - 1, -1, -1):
            g = self.layer_bwd(recs[i], g, below=recs[i - 1] if i > 0 else None)
        return g

    def layer_bwd(self, rec, g_out, below=None):
        """g_out[BN_SUMS] (optional): the LOCAL statistics of this layer's patient BatchNorm backward, summed by the kernel
        that produced g_out[ROW_TYPE]; below: the record of the layer underneath, whose patient BatchNorm consumes the
        patient gradient this layer hands down (its statistics then ride on the last kernel that writes that gradient)."""
        plan, D, P, l = self.plan, self.D, self.plan.n_rows, rec["l"]
        x, y = rec["x"], rec["y"]
        dy = {}

        def bn_bwd_t(t, sums=None):
            gt = g_out.get(t)
            if gt is None or t not in y:
                return
            bn_prefix = f"batch_norms.{l}.{t}" if self.m.use_batch_norm else None
            dy[t] = self.bn_bwd(gt.contiguous(), y[t], rec["pros"][t], rec["folds"][t], bn_prefix,
                                sharded=(t == ROW_TYPE), sums=sums)

        def patient_sums():      # statistics of the patient rows' BN backward (sharded: a collective)
            gt = g_out.get(ROW_TYPE)
            if gt is None or ROW_TYPE not in y:
                return None
            pre = g_out.get(BN_SUMS)
            if pre is not None and rec["folds"][ROW_TYPE] is not None:
                if self.comm is not None:
                    self.allreduce(pre)
                return pre
            return self.bn_bwd_sums(gt.contiguous(), y[ROW_TYPE], rec["pros"][ROW_TYPE], rec["folds"][ROW_TYPE], True)

        g_in: Dict[str, Optional[torch.Tensor]] = {t: None for t in x}

        def add_dgrad(t, dy_, W_):
            """g_in[t] += dy_ . W_  (W_ is the forward weight [N,K], read in place; accumulated by the GEMM itself)."""
            if g_in[t] is None:
                g_in[t] = ops.linear_fwd(dy_, W_, w_kn=True)
            else:
                ops.linear_fwd(dy_, W_, w_kn=True, out=g_in[t], accumulate=True)

        def vocab_1():
            """BN backward of the vocab types and everything that only depends on it (vocab destinations)."""
            vts = [t for t in y if t != ROW_TYPE and g_out.get(t) is not None]
            if vts and all(y[t].shape[0] <= ops.SMALL_MAX_ROWS for t in vts):
                items = [(g_out[t].contiguous(), y[t], rec["pros"][t], rec["folds"][t]) for t in vts]
                for t, (dyt, dbeta, dgamma) in zip(vts, ops.small_bn_bwd_group(items)):
                    dy[t] = dyt
                    if dbeta is not None:
                        self.acc(f"batch_norms.{l}.{t}.bias", dbeta)
                        self.acc(f"batch_norms.{l}.{t}.weight", dgamma)
            else:
                for t in y:
                    if t != ROW_TYPE:
                        bn_bwd_t(t)
            rels = []
            live = [(r, agg) for r, agg in zip(rec["rout"], rec.get("aggs", [])) if dy.get(r.other) is not None]
            if live and all(r.n_cols <= ops.SMALL_MAX_ROWS for r, _ in live) and \
                    all(g_in[r.other] is None for r, _ in live) and len({r.other for r, _ in live}) == len(live):
                # grouped: the weight gradients of every relation in one launch, their data gradients in another
                wg = []
                for r, agg in live:
                    dyv = dy[r.other]
                    wg += [ops.SmallWgrad(dyv, agg, with_bias=True), ops.SmallWgrad(dyv, x[r.other])]
                res = ops.small_wgrad_group(wg)
                fw = []
                for r, agg in live:
                    nme = self.conv_name(l, r.edge_type)
                    dyv = dy[r.other]
                    fw += [ops.SmallFwd(dyv, self.W(nme + ".lin_r.weight"), w_kn=True),      # -> g_in[other]
                           ops.SmallFwd(dyv, self.W(nme + ".lin_l.weight"), w_kn=True)]      # -> d agg
                outs = ops.small_fwd_group(fw)
                for i, (r, agg) in enumerate(live):
                    nme = self.conv_name(l, r.edge_type)
                    self.acc(nme + ".lin_l.weight", res[2 * i][0])
                    self.acc(nme + ".lin_l.bias", res[2 * i][1])
                    self.acc(nme + ".lin_r.weight", res[2 * i + 1][0])
                    g_in[r.other] = outs[2 * i]
                    rels.append(ops.Rel(r.rowptr, r.col, r.n_cols, colscale=r.inv_col, table=outs[2 * i + 1],
                                        simple=r.simple, mask_r=r.mask_r))
                return rels
            for r, agg in live:
                dyv = dy[r.other]
                nme = self.conv_name(l, r.edge_type)
                dWl, dbl = ops.linear_wgrad(dyv, agg, with_bias=True)     # the bias gradient rides in the same pass
                self.acc(nme + ".lin_l.weight", dWl)
                self.acc(nme + ".lin_l.bias", dbl)
                self.acc(nme + ".lin_r.weight", ops.linear_wgrad(dyv, x[r.other]))
                add_dgrad(r.other, dyv, self.W(nme + ".lin_r.weight"))
                dagg = ops.linear_fwd(dyv, self.W(nme + ".lin_l.weight"), w_kn=True)
                rels.append(ops.Rel(r.rowptr, r.col, r.n_cols, colscale=r.inv_col, table=dagg, simple=r.simple, mask_r=r.mask_r))
            return rels

        def patient_1(sums=None, reduce=True):
            """BN backward of the patient rows, their weight / data gradients, scatter of dy_P onto the vocab rows."""
            gt, fused = g_out.get(ROW_TYPE), None
            if gt is not None and ROW_TYPE in y and rec["rin"] and g_in[ROW_TYPE] is None:
                # BatchNorm backward inside the data-gradient GEMM of the self-loop weights (sum of the three lin_r)
                fused = self.bn_lin_bwd(gt.contiguous(), y[ROW_TYPE], rec["pros"][ROW_TYPE], rec["folds"][ROW_TYPE],
                                        f"batch_norms.{l}.{ROW_TYPE}" if self.m.use_batch_norm else None, True, sums,
                                        rec["Wsum"])
            if fused is not None:
                dy[ROW_TYPE], g_in[ROW_TYPE] = fused
            else:
                bn_bwd_t(ROW_TYPE, sums)
            dyP = dy.get(ROW_TYPE)
            if dyP is None or not rec["rin"]:
                return None
            xP = x[ROW_TYPE]
            dWsum, dbsum = ops.linear_wgrad(dyP, xP, with_bias=True, defer=self.wgrad_jobs)
            if fused is None:
                add_dgrad(ROW_TYPE, dyP, rec["Wsum"])
            rels, dTs, off = [], [], 0
            buf = torch.empty(sum(r.n_cols for r in rec["rin"]), D, device=self.dev)
            for r in rec["rin"]:
                dT = buf[off:off + r.n_cols]
                off += r.n_cols
                rels.append(ops.Rel(r.rowptr, r.col, r.n_cols, rowscale=r.inv_row, out=dT, simple=r.simple, mask_t=r.mask_t))
                dTs.append(dT)
            ops.scatter_rows(rels, P, D, dyP)
            if reduce:
                self.allreduce(buf)
            return dTs, dWsum, dbsum, buf

        def vocab_2(res):
            """Per-relation work behind the scatter: gradients of the vocab-table transforms T_v = x_v W_l^T."""
            if res is None:
                return
            dTs, dWsum, dbsum, _ = res
            rin_ = rec["rin"]
            if all(r.n_cols <= ops.SMALL_MAX_ROWS for r in rin_) and len({r.other for r in rin_}) == len(rin_):
                # grouped: gradients of the vocab-table transforms T_v = x_v W_l^T, one launch each for dW and dX
                wres = ops.small_wgrad_group([ops.SmallWgrad(dT, x[r.other]) for r, dT in zip(rin_, dTs)])
                fw = []
                for r, dT in zip(rin_, dTs):
                    nme = self.conv_name(l, r.edge_type)
                    fw.append(ops.SmallFwd(dT, self.W(nme + ".lin_l.weight"), out=g_in[r.other],
                                           accumulate=g_in[r.other] is not None, w_kn=True))
                outs = ops.small_fwd_group(fw)
                for (r, dT), (dW, _), o in zip(zip(rin_, dTs), wres, outs):
                    nme = self.conv_name(l, r.edge_type)
                    self.acc(nme + ".lin_r.weight", dWsum, partial=True)
                    self.acc(nme + ".lin_l.bias", dbsum, partial=True)
                    self.acc(nme + ".lin_l.weight", dW)
                    g_in[r.other] = o
                return
            for r, dT in zip(rin_, dTs):
                nme = self.conv_name(l, r.edge_type)
                self.acc(nme + ".lin_r.weight", dWsum, partial=True)
                self.acc(nme + ".lin_l.bias", dbsum, partial=True)
                self.acc(nme + ".lin_l.weight", ops.linear_wgrad(dT, x[r.other]))
                add_dgrad(r.other, dT, self.W(nme + ".lin_l.weight"))

        def patient_2(rels):
            if rels:
                nb = None
                if below is not None and ROW_TYPE in below["y"] and below["folds"].get(ROW_TYPE) is not None and \
                        "conv" in self.next_bn_sites:
                    nb = ops.NextBN(below["y"][ROW_TYPE], below["pros"][ROW_TYPE], below["folds"][ROW_TYPE])
                acc_ = g_in[ROW_TYPE] is not None
                if not acc_:
                    g_in[ROW_TYPE] = torch.empty(P, D, device=self.dev)
                if nb is not None:           # the gather is the last writer of the gradient the BatchNorm below consumes
                    _, g_in[BN_SUMS] = ops.gather_rows(rels, P, D, g_in[ROW_TYPE], accumulate=acc_, next_bn=nb)
                else:
                    ops.gather_rows(rels, P, D, g_in[ROW_TYPE], accumulate=acc_)

        if self.overlap and self.comm is None:
            main, side = torch.cuda.current_stream(), self.side
            side.wait_stream(main)
            ev_tables, ev_scatter = torch.cuda.Event(), torch.cuda.Event()
            with torch.cuda.stream(side):
                rels = vocab_1()
                ev_tables.record(side)
            res = patient_1(patient_sums() if g_out.get(BN_SUMS) is not None else None)
            ev_scatter.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_scatter)
                vocab_2(res)
            main.wait_event(ev_tables)
            patient_2(rels)
            main.wait_stream(side)
        elif self.overlap:
            # sharded: the two collectives of this layer (BN statistics, scattered partial sums) stay on the main stream
            # and every fork is joined before the next one
            main, side = torch.cuda.current_stream(), self.side
            sums = patient_sums()                        # collective
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rels = vocab_1()
            res = patient_1(sums, reduce=False)
            main.wait_stream(side)
            if res is not None:
                self.allreduce(res[3])                   # collective
            side.wait_stream(main)
            with torch.cuda.stream(side):
                vocab_2(res)
            patient_2(rels)
            main.wait_stream(side)
        else:
            rels = vocab_1()
            res = patient_1(patient_sums() if g_out.get(BN_SUMS) is not None else None)
            vocab_2(res)
            patient_2(rels)
        return g_in

    # ======================================================================== heads
    def head_weight_halves(self):
        """W1[:, :D] | W1[:, D:] of both heads' first Linear(2D, H) as contiguous matrices: ONE launch for the four."""
        D, out, jobs = self.D, {}, []
        for which in ("edge_predictor", "tabular_mlp"):
            w1 = getattr(self.m, which).mlp[0].weight.detach()
            halves = torch.empty(2, w1.shape[0], D, device=self.dev)
            jobs += [(halves[0], [w1[:, :D]]), (halves[1], [w1[:, D:]])]
            out[which] = (halves[0], halves[1])
        ops.vec_sums(jobs)
        return out

    def head_tensors(self, which, xP, xlab, w1a, w1b):
        """xP: the patient rows this head can see (all of them, or the compacted low-degree rows)."""
        mod = getattr(self.m, which)
        if isinstance(xP, _LazyAct):
            A = ops.linear_fwd(xP.y, w1a, pro=xP.pro)
        else:
            A = ops.linear_fwd(xP, w1a) if xP.shape[0] else ops.zeros(1, w1a.shape[0], device=self.dev)
        B = ops.linear_fwd(xlab, w1b, mod.mlp[0].bias.detach())
        head = ops.Head(A, B, mod.mlp[3].weight.detach(), mod.mlp[3].bias.detach(),
                        mod.mlp[6].weight.detach().reshape(-1).contiguous(), mod.mlp[6].bias.detach())
        return head, w1a, w1b

    def heads_fwd(self, init, fin):
        plan = self.plan
        if LAB_EDGE not in plan.rels:
            raise KeyError(f"graph has no {LAB_EDGE} relation (model.py:297)")
        pi, li, perm, ids, (sel_low, sel_high, counts, n_low, n_high, low_rows, pi_low, deg_low, _) = self.pairs
        thr = int(self.m.degree_threshold)
        if self.lists_ready is not None:             # drawn / selected on the side stream beside the encoder pass
            torch.cuda.current_stream().wait_event(self.lists_ready)
        if self.forward_select is not None:
            pred = ops.zeros(pi.numel(), device=self.dev)        # pairs outside the lists: 0 (never read by the loss)
            sel_low, sel_high, counts = self.forward_select      # (subsets of the static lists: n_low / n_high still bound them)
        else:
            pred = torch.empty(pi.numel(), device=self.dev)      # every pair belongs to exactly one head list
        rec = dict(init=init, fin=fin)
        halves = self.head_weight_halves()
        for which, src, want_low in (("edge_predictor", fin, False), ("tabular_mlp", init, True)):
            # tabular_mlp: patient rows, ids and gate of the compacted low-degree patients (gate: all of them are low)
            if want_low and self._init_compact:
                xP = src[ROW_TYPE]                                   # the first pass was evaluated on these rows only
            else:
                xP = src[ROW_TYPE].index_select(0, low_rows) if want_low else src[ROW_TYPE]
            head, w1a, w1b = self.head_tensors(which, xP, src["lab"], *halves[which])
            sel, n_sel, nb = (sel_low, counts[0:1], n_low) if want_low else (sel_high, counts[1:2], n_high)
            # training: the forward leaves the first layer's sign bits and the second layer's activations of the pairs it
            # visits (136 B per pair) -- the backward then recomputes neither the dropout masks nor the 64 x 32 product
            # (indexed by list position -- dense -- when the backward is known to run over the same lists)
            by_pos = self.static_select is self.forward_select
            save = ops.pair_saved_alloc(nb if by_pos else pi.numel(), self.dev) + (by_pos,) \
                if self.T and self.save_pair_state and self.forward_select is not None else None
            ops.pair_head_fwd(head, pi_low if want_low else pi, li, deg_low if want_low else plan.lab_deg, thr, want_low,
                              self.p, self.seed, ids, pred, self.seed_dev,
                              sel=sel, n_sel=n_sel, n_bound=nb, io_perm=perm, save=save)   # written in the caller's pair order
            rec[which] = (head, w1a, w1b, xP, save)
        return pred, rec

    def heads_bwd(self, rec, dpred):
        plan, D = self.plan, self.D
        pi, li, perm, ids, (_, _, _, n_low, n_high, low_rows, pi_low, deg_low, _) = self.pairs
        thr = int(self.m.degree_threshold)
        dps = dpred.contiguous()                 # caller's pair order: the kernels read it through perm
        n_lab = plan.num_nodes["lab"]
        gsets = {}
        # pairs with a zero upstream gradient (everything outside the supervision subset, train.py:366-370) add
        # exactly nothing: visit only the others, split by head
        if self.static_select is not None:
            # the supervision subset is known (and a pair of it whose gradient happens to be 0 adds exactly nothing):
            # no selection pass in the step; the kernels read dpred through the pair permutation
            bsel_low, bsel_high, bcounts = self.static_select
            dsrc, dio = dps, perm
        else:
            dsorted = torch.empty_like(dps)      # dpred in sorted pair order: the one random pass, made by the selection
            bsel_low, bsel_high, bcounts = ops.pair_select(pi, plan.lab_deg, thr, dps, io_perm=perm, dpred_sorted=dsorted)
            dsrc, dio = dsorted, None
        order = (("edge_predictor", rec["fin"], False), ("tabular_mlp", rec["init"], True))
        # one zero-fill for the small gradients of BOTH heads; the two lab-side tables dB sit first and adjacent, so that
        # a sharded run sums them over the ranks with ONE all-reduce
        smalls = {w: [rec[w][0].B, rec[w][0].W2, rec[w][0].b2, rec[w][0].W3, rec[w][0].b3] for w, _, _ in order}
        n_b = sum(smalls[w][0].numel() for w, _, _ in order)
        n_small = sum(t.numel() for w, _, _ in order for t in smalls[w])
        n_small_pad = (n_small + 63) & ~63               # (the dA tables behind them start on a 256-byte boundary)
        # ...and the per-patient tables dA of both heads behind them: ONE zero-fill for everything the kernels add into
        flat = ops.zeros(n_small_pad + sum(rec[w][0].A.numel() for w, _, _ in order), device=self.dev)
        views, ob, o, oa, dA = {}, 0, n_b, n_small_pad, {}
        for w, _, _ in order:
            tB = smalls[w][0]
            v = [flat[ob:ob + tB.numel()].view(tB.shape)]
            ob += tB.numel()
            for t in smalls[w][1:]:
                v.append(flat[o:o + t.numel()].view(t.shape))
                o += t.numel()
            views[w] = v
            tA = rec[w][0].A
            dA[w] = flat[oa:oa + tA.numel()].view(tA.shape)
            oa += tA.numel()
        gs, join = {}, []
        for which, src, want_low in order:
            head, w1a, w1b, xP, save = rec[which]
            g = ops.Head(dA[which], *views[which])
            sel, n_sel, nb = (bsel_low, bcounts[0:1], n_low) if want_low else (bsel_high, bcounts[1:2], n_high)
            ops.pair_head_bwd(head, g, pi_low if want_low else pi, li, deg_low if want_low else plan.lab_deg, thr,
                              want_low, n_lab, self.p, self.seed, ids, dsrc,
                              self.seed_dev, sel=sel, n_sel=n_sel, n_bound=nb, io_perm=dio, saved=save)
            gs[which] = g
            self.acc(f"{which}.mlp.3.weight", g.W2, partial=True)
            self.acc(f"{which}.mlp.3.bias", g.b2, partial=True)
            self.acc(f"{which}.mlp.6.weight", g.W3.reshape(1, -1), partial=True)
            self.acc(f"{which}.mlp.6.bias", g.b3, partial=True)
            # first-layer weight / bias gradients from this shard's pairs only (the LOCAL dA and dB): per-shard partial
            # sums like every other parameter gradient, summed by the one bucket at the end of the backward
            if isinstance(xP, _LazyAct):
                dW1a = ops.linear_wgrad(g.A, xP.y, xP.pro)
            elif xP.shape[0]:
                dW1a = ops.linear_wgrad(g.A, xP)
            else:
                dW1a = ops.zeros(*w1a.shape, device=self.dev)
            dW1b, db1 = ops.linear_wgrad(g.B, src["lab"], with_bias=True)
            dW1 = torch.empty(w1a.shape[0], 2 * D, device=self.dev)
            join += [(dW1[:, :D], [dW1a]), (dW1[:, D:], [dW1b])]
            self.acc(f"{which}.mlp.0.weight", dW1, partial=True)
            self.acc(f"{which}.mlp.0.bias", db1, partial=True)
        ops.vec_sums(join)                       # [dW1a | dW1b] of both heads: one launch
        self.allreduce(flat[:n_b])               # lab-side partials dB of both heads from the sharded pairs
        for which, src, want_low in order:
            head, w1a, w1b, xP, _ = rec[which]
            g = gs[which]
            glab = ops.linear_fwd(g.B, w1b, w_kn=True)
            if want_low:                         # gradient rows of the low-degree patients only: (row ids, rows)
                if xP.shape[0]:
                    gP = (low_rows, ops.linear_fwd(g.A, w1a, w_kn=True))
                elif self.comm is not None:
                    # a shard without a low-degree patient still takes part in the backward of the first encoder pass:
                    # its BatchNorm backward needs the GLOBAL sums (rows with a zero upstream gradient get a non-zero
                    # dz2 once any other shard has one), and every rank must issue the same sequence of collectives
                    gP = (low_rows, torch.zeros(0, D, device=self.dev))
                else:
                    gP = None                    # single GPU, no such row anywhere: the whole pass contributes exactly 0
                gsets[which] = {ROW_TYPE: gP, "lab": glab}
            elif isinstance(xP, _LazyAct) and xP.fold is not None and "heads" in self.next_bn_sites:
                # the final patient activations are BatchNorm outputs: the statistics of that BatchNorm's backward are
                # summed in the epilogue of the GEMM that produces its upstream gradient
                gP, bsums = ops.linear_fwd(g.A, w1a, w_kn=True, next_bn=ops.NextBN(xP.y, xP.pro, xP.fold))
                gsets[which] = {ROW_TYPE: gP, "lab": glab, BN_SUMS: bsums}
            else:
                gsets[which] = {ROW_TYPE: ops.linear_fwd(g.A, w1a, w_kn=True), "lab": glab}
        return gsets["tabular_mlp"], gsets["edge_predictor"]


# =============================================================================================
# factory + loss (model.py:523-612)
# =============================================================================================
def build_model(config: Dict, metadata: Tuple, patient_feature_dim: int):
    model_config = config["model"]
    architecture = model_config["architecture"]
    if architecture == "RGCN":
        model = HeteroRGCN(metadata=metadata, hidden_dim=model_config["hidden_dim"],
                           num_layers=model_config["num_layers"], dropout=model_config["dropout"],
                           patient_feature_dim=patient_feature_dim,
                           use_batch_norm=model_config["use_batch_norm"], activation=model_config["activation"])
        logging.info("Built HeteroRGCN model")
    elif architecture == "HGT":
        raise NotImplementedError("HGT is outside the accelerated hot path (SURVEY.md section 2: not the "
                                  "configured architecture, conf/config.yaml:174)")
    else:
        raise ValueError(f"Unknown architecture: {architecture}")
    num_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    logging.info(f"Model has {num_params:,} trainable parameters")
    return model


def compute_regression_loss(predictions: torch.Tensor, targets: torch.Tensor, loss_type: str = "mae") -> torch.Tensor:
    if loss_type == "mae":
        return F.l1_loss(predictions, targets)
    if loss_type == "mse":
        return F.mse_loss(predictions, targets)
    if loss_type == "huber":
        return F.huber_loss(predictions, targets)
    raise ValueError(f"Unknown loss type: {loss_type}")
