#!/usr/bin/env python3
"""Benchmark of the hot path: has_lab edges/s for one full training step
(new 20 % supervision mask + fwd + weighted-MAE loss + bwd + Adam) of ``predict_lab_values`` on a synthetic eICU-shape
hetero-graph.

    python bench.py --gpus N --steps K --warmup W [--scale S] [--dim D] [--strong] [--no-strong-x1000]

One process per GPU (N>1: launched by torch.distributed.run, RCCL).  Workload at N GPUs:
  headline (weak) : every GPU holds an  x<scale>  eICU-shape shard (1,834*scale patients, 61,484*scale
                    has_lab edges, 128-d) of ONE global graph of N shards; vocab nodes shared.
                    N=1, scale=100 is BASELINE.json configs[2].
  --strong        : ONE global x<scale> graph, patient-sharded over the N GPUs (headline becomes the strong run).
  strong_x1000    : in the same invocation, unless --no-strong-x1000: BASELINE.json config 4 -- ONE x1000 graph (1.83 M
                    patients, 61.5 M has_lab edges) at 256-d, patient-sharded over the N GPUs, RCCL all-reduces on the
                    shared vocab-side sums / gradients -- and the same graph at 128-d (north_star's 1 -> 8 curve).
                    Reported as objects of the same JSON line, so the per-N lines of a 1/2/4/8 sweep give the curves.
A step is what one epoch of the full-batch reference is (src/train.py:332-392): the supervision subset is redrawn
(train.py:150-176) INSIDE the timed step (device RNG, mmg_sup_mask_draw), then forward, loss, backward, optimizer.
Prints ONE JSON line (rank 0):
  value / ms_per_step : the timed region (K steps, barrier + synchronize on both sides, max over ranks);
  fixed_supervision_mask : the same replays without the per-step redraw (last round's headline), for comparison;
  trainer_epoch       : N=1: the SAME step reached through the reference's call surface -- mmgnn.train.Trainer
                        (train_epoch / validate / the loop body of Trainer.train with its one host read per epoch);
  chain_overhead_ms   : N=1: the step replayed the way a sharded run launches it (world_size-1 `nccl` group: ONE
                        hipGraph with the RCCL all-reduces recorded inside; `chain.segment_chain` = round 3's chain of
                        segments with the all-reduces issued between them) minus the single-graph step: the fixed cost
                        of the multi-GPU launch scheme, measurable on one GPU;
  roofline            : the kernel (instantiated symbol) of the family with the most kernel time per step.  Kernel
                        durations are the HIP start / stop events that hipExtLaunchKernelGGL attaches to each launch
                        (mmg_probe_*: the kernel's own begin / end timestamps on its stream), taken in eager steps of the
                        same kernels right after the timed region (graph replays cannot carry per-kernel events);
  kernels             : the same figures for the kernels north_star grades -- the per-relation gather and scatter
                        (scatter = matrix kernel + its fixed-order slab sum) -- and the other big kernel families;
  cpu_baseline        : the oracle (CPU restatement of the reference) timed on this host's cores, N=1 only, on bounded
                        samples of the same workload; the full x100 step (run once, profiles/) is quoted beside it.
"""
import argparse
import gc
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))


def self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start `torch.distributed.run` with N ranks of THIS
    file as a CHILD process (never a re-exec), relay rank 0's one JSON line and return the child's exit code.  Runs before
    anything of this process has imported torch or touched the GPU.  None = nothing to do (N = 1, or already a rank)."""
    n = 1
    for i, a in enumerate(sys.argv[1:], 1):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ or "--cpu-only" in sys.argv:
        return None
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=REPO)
    line = None
    for raw in proc.stdout:                  # rank 0 prints exactly one JSON line; anything else goes to stderr
        txt = raw.decode(errors="replace")
        if txt.lstrip().startswith("{") and line is None:
            line = txt
        else:
            sys.stderr.write(txt)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line if line.endswith("\n") else line + "\n")
        sys.stdout.flush()
    return rc


if __name__ == "__main__":
    _rc = self_launch()
    if _rc is not None:
        sys.exit(_rc)

import torch  # noqa: E402

sys.path.insert(0, REPO)
import mmgnn  # noqa: E402,F401
from mmgnn import dist as mdist  # noqa: E402
from mmgnn import model as mmodel  # noqa: E402
from mmgnn import ops  # noqa: E402
from mmgnn.data import build_plan  # noqa: E402
from mmgnn.model import build_model  # noqa: E402
from mmgnn.synth import make_graph  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3     # dense fp32 MFMA
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA
MFMA_F16_PEAK_TF = 2500.0    # dense f16 MFMA (same rate: v_mfma_f32_32x32x16_f16 / _bf16 issue once per 32 clocks)
# the dense layers and the aggregates compute fp32 products as exact bf16 pieces on the bf16 matrix cores: 6 matrix
# FLOP per algorithmic fp32 FLOP (linear layers, the pair forward's 64 x 32 layer; 3 f16 products at K = N = 256: op_summary),
# 3 (0/1-indicator aggregates on bf16 pieces; the forward scatter: 2 f16 pieces -- they carry no algorithmic FLOPs here)
MFMA_PEAK_BY_OP = {"linear_fwd": MFMA_BF16_PEAK_TF / 6, "linear_wgrad": MFMA_BF16_PEAK_TF / 6,
                   "gather_rows": MFMA_BF16_PEAK_TF / 3, "scatter_rows": MFMA_BF16_PEAK_TF / 3,
                   "pair_head_fwd": MFMA_BF16_PEAK_TF / 6, "pair_head_bwd": MFMA_F32_PEAK_TF}
LAB = ("patient", "has_lab", "lab")
# kernels that belong to ONE op: the fixed-order slab sums are part of the scatter / weight-gradient they finish
FOLLOWERS = {"scatter_reduce": "scatter_rows", "linear_wgrad_reduce": "linear_wgrad"}
MASK_FRACTION = 0.2          # conf/config.yaml train.mask_fraction


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scale", type=int, default=100, help="eICU-shape multiples per GPU (weak) or in total (--strong)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--shape", choices=["eicu", "mimic"], default="eicu",
                    help="vocabulary of the synthetic graph: eicu = 50 / 114 / 100 (BASELINE configs 1-4), mimic = 50 / 200 / "
                         "100 (config 5: conf/config.yaml's MIMIC vocabulary caps)")
    ap.add_argument("--strong", action="store_true")
    ap.add_argument("--no-strong-x1000", action="store_true", help="skip the x1000 patient-sharded sub-records")
    ap.add_argument("--strong-scale", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scale", type=int, default=10, help="largest CPU sample (x1 is always timed as well)")
    ap.add_argument("--profile-ops", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replays")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel probe pass (profiler runs)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the fixed-mask, Trainer and segment-chain sub-records (profiler runs)")
    ap.add_argument("--next-bn", default=None,
                    help="comma list of BatchNorm-backward statistics taken from producer epilogues (mmgnn.model.set_next_bn; "
                         "'none' = all of them as separate passes); default: the library's own")
    ap.add_argument("--overlap", choices=["auto", "off", "on"], default="auto",
                    help="vocab-side work of a layer on a side stream (mmgnn.model.set_overlap)")
    return ap.parse_args()


def setup_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:       # (python bench.py --gpus N starts its own ranks: self_launch)
            raise SystemExit("--gpus N>1 outside a launcher: run `python bench.py --gpus N` or torch.distributed.run")
        args.gpus = world
    n_dev = max(torch.cuda.device_count(), 1)
    backend = os.environ.get("MMG_DIST_BACKEND", "nccl")     # "gloo": rehearse N ranks on fewer GPUs (tests only)
    if world > n_dev and backend == "nccl":
        raise SystemExit(f"{world} ranks need {world} GPUs with the RCCL backend ({n_dev} visible)")
    local = local % n_dev
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
    return world, rank, torch.device("cuda", local)


def model_config(args, dim):
    return {"model": {"architecture": "RGCN", "hidden_dim": dim, "num_layers": 2, "dropout": args.dropout,
                      "use_batch_norm": True, "activation": "relu"}}


def build_workload(args, world, rank, dev, scale, strong, dim, force_comm=False):
    cfg = model_config(args, dim)
    comm = mdist.ShardComm() if (world > 1 or force_comm) else None
    from mmgnn.synth import EICU, MIMIC_LIKE
    shape = MIMIC_LIKE if args.shape == "mimic" else EICU
    if strong and world > 1:
        g_all = make_graph(scale, seed=0, device=dev, shape=shape)
        w = mdist.patient_weights(g_all)
        b = mdist.partition_rows(w, world)
        lo, hi = b[rank], b[rank + 1]
        g = mdist.shard_graph(g_all, lo, hi)
        n_global = int(g_all["patient"].num_nodes)
        del g_all, w
    else:
        g = make_graph(scale, seed=(0 if strong else 1000 * rank), device=dev, shape=shape)   # this rank's shard
        P = int(g["patient"].num_nodes)
        lo, hi, n_global = rank * P, (rank + 1) * P, world * P
    torch.manual_seed(42)
    model = build_model(cfg, (g.node_types, g.edge_types), None).to(dev)
    model._init_embeddings(g)
    plan = build_plan(g, dev, use_cache=False)
    if comm is not None:
        mdist.shard_plan(plan, comm, lo, n_global)
        mdist.shard_model(model, comm)
        for name, p in model.named_parameters():   # replicated weights: identical on every rank
            if name != "embeddings.patient.weight":  # (the patient table is sharded, not replicated)
                torch.distributed.broadcast(p.data, 0)
    # EdgeMasker rule (train.py:98-129) on this shard's has_lab edges, device RNG for speed at scale
    ei = g[LAB].edge_index
    E = ei.shape[1]
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    perm = torch.randperm(E, generator=gen, device=dev)
    tr = perm[: int(0.7 * E)].sort().values
    del perm
    pi, li = ei[0][tr].contiguous(), ei[1][tr].contiguous()
    y = g[LAB].edge_attr[tr].squeeze(-1).contiguous()
    sup = torch.rand(tr.numel(), generator=torch.Generator(device=dev).manual_seed(1234 + rank), device=dev) < MASK_FRACTION
    wlab = torch.ones(int(g["lab"].num_nodes), device=dev)
    # F5: embeddings are not in the optimizer.  mmgnn.optim.Adam = torch.optim.Adam's arithmetic as ONE launch of
    # mmg_adam_step_dev over the flat parameter bucket
    from mmgnn.optim import Adam
    opt = Adam([p for n, p in model.named_parameters() if not n.startswith("embeddings.")], lr=1e-3, weight_decay=1e-5)
    n_sup_global = torch.tensor([float(sup.sum())], device=dev)
    if comm is not None:
        torch.distributed.all_reduce(n_sup_global)
    return dict(model=model, plan=plan, g=g, pi=pi, li=li, y=y, sup=sup, supf=sup.float(), wpair=wlab[li].contiguous(),
                wlab=wlab, opt=opt, E=E, n_sup=float(n_sup_global), comm=comm)


def train_step(w):
    model, opt = w["model"], w["opt"]
    model.train()
    model.zero_grad(set_to_none=True)    # all parameters (the frozen embedding tables too): .grad adopted, not accumulated
    pred = model.predict_lab_values(w["plan"], w["pi"], w["li"])
    # weighted MAE over the supervision subset (train.py:366-386), global mean over all shards: one fused pass
    loss = ops.weighted_pair_loss(pred, w["y"], w["wpair"], w["supf"], 1.0 / w["n_sup"], "mae")
    loss.backward()
    opt.step()
    return loss


def host_cores():
    """CPU share of this process: affinity mask, cgroup quota, and the one-GPU box share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_sample(args, s, n_warm, n_timed, cores):
    """The oracle's training step (fwd + weighted-MAE loss + bwd; no optimizer) on the x<s> eICU-shape graph."""
    from oracle import model as om, train as ot
    g = make_graph(s, seed=0, device="cpu")
    gv = om.GraphView(g)
    sd = om.init_state(gv.num_nodes, gv.edge_types, args.dim, 2, seed=42)
    ei, ea = g[LAB].edge_index, g[LAB].edge_attr
    E = ei.shape[1]
    tr = torch.randperm(E, generator=torch.Generator().manual_seed(42))[: int(0.7 * E)].sort().values
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    sup = torch.rand(tr.numel(), generator=torch.Generator().manual_seed(1234)) < MASK_FRACTION
    wl = torch.ones(gv.num_nodes["lab"])
    torch.set_num_threads(cores)
    times = []
    for i in range(n_warm + n_timed):
        t0 = time.perf_counter()
        ot.train_step_grads(sd, gv, pi, li, y, wl, sup, p=args.dropout)
        times.append(time.perf_counter() - t0)
    t = sorted(times[n_warm:])[n_timed // 2]
    return dict(scale=s, has_lab_edges=E, s_per_step=t, edges_per_s=E / t, warmup=n_warm, timed=n_timed)


def cpu_baseline(args):
    """the CPU restatement beside the GPU figure -- x1 (the reference's own CPU-runnable case,
    2 warm-up + 5 timed steps) and a bounded sample of the x100 workload (x<cpu-scale>).  The full x100 step (~1 min and
    tens of GB on 16 cores) is not part of the default run: it was timed once with `--cpu-only --cpu-scale 100` and its
    record (profiles/r3_cpu_x100.json) is quoted as `x100`."""
    cores = host_cores()
    s_big = max(1, min(args.cpu_scale, args.scale))
    x1 = cpu_sample(args, 1, 2, 5, cores)
    big = cpu_sample(args, s_big, 1 if s_big > 1 else 2, 3 if s_big > 1 else 5, cores) if s_big > 1 else x1
    out = {"value": big["edges_per_s"], "unit": "has_lab edges/s", "cores": cores, "kind": "port",
           "cpu_model": cpu_model_name(),
           "sample": f"oracle (pure PyTorch CPU restatement of the reference) fwd + weighted-MAE loss + bwd, no optimizer "
                     f"step (the GPU step includes Adam), on the x{big['scale']} eICU-shape graph "
                     f"({big['has_lab_edges']} has_lab edges, {args.dim}-d, dropout {args.dropout}): median of "
                     f"{big['timed']} steps after {big['warmup']} warm-up, {cores} threads",
           "x1": {"value": x1["edges_per_s"], "s_per_step": x1["s_per_step"], "has_lab_edges": x1["has_lab_edges"],
                  "warmup": 2, "timed": 5},
           "s_per_step": big["s_per_step"]}
    path = os.path.join(REPO, "profiles", "r3_cpu_x100.json")
    if os.path.exists(path):
        try:
            rec = json.load(open(path))
            rec["source"] = "profiles/r3_cpu_x100.json (one untimed-warm-up-free step, run once on a GPU box's host cores)"
            out["x100"] = rec
        except Exception:
            pass
    return out


# ------------------------------------------------------------------------------------------------ kernel table
def alg_of(tag, M, N, K, fl, extra):
    """Algorithmic (compulsory) bytes and flops of one launch of a probed kernel family."""
    if tag == "linear_fwd":
        if fl & (16 | 64):
            # BatchNorm / L2-norm backward inside the data-gradient GEMM: reads G and Y (and a second G: flag 128; no G at
            # all but a row list: flag 256), writes dZ and dX
            n_in = 4 if fl & 128 else (2 if fl & 256 else 3)
            return 4 * (n_in * M * K + N * K + M * N), 2 * M * N * K
        return 4 * (M * K + N * K + M * N * (2 if fl & 1 else 1)), 2 * M * N * K
    if tag == "linear_wgrad":
        return 4 * (M * N + M * K + N * K), 2 * M * N * K
    if tag in ("gather_rows", "scatter_rows"):
        # every index once, every distinct feature row once: 4 E + 4 (P + 1) per relation + the tables + the patient
        # tensor (read and written when the gather accumulates); E comes from the workload (extra)
        e = extra.get((tag, K, fl & (1 if tag == "gather_rows" else 8)), None)
        return (e if e is not None else 0), 0
    if tag == "pair_head_fwd":
        # per pair: 12 B of indices / output + its two 256-B first-layer rows are table reads (A: one row per patient,
        # B: per lab) -> 12 n + 256 (P + L) compulsory; 2 * (64 * 32 + 32) FLOP per pair
        return 12 * M + 256 * extra.get("pair_rows", 0), 4160 * M
    if tag == "pair_head_bwd":
        # M = the launch bound (all pairs of the head); the kernel visits the supervised ones only: 4 chained products
        # of 2 * 64 * 32 FLOP per visited pair
        n_vis = extra.get("pair_visited", M)
        return 12 * n_vis + 2 * 256 * extra.get("pair_rows", 0), 16384 * n_vis
    return 0, 0


def kernel_table(rows, n_steps, extra):
    """probe rows -> list of per-(family, shape, flags) stats; followers (slab sums) are folded in by op_summary."""
    groups = {}
    for ms, tag, M, N, K, fl, sym in rows:
        g = groups.setdefault((tag, M, N, K, fl), [0, 0.0, sym])
        g[0] += 1
        g[1] += ms
    table = []
    for (tag, M, N, K, fl), (cnt, tot, sym) in groups.items():
        b, f = alg_of(tag, M, N, K, fl, extra)
        table.append(dict(kernel=tag, symbol=sym, M=M, N=N, K=K, flags=fl, launches_per_step=cnt / n_steps, avg_ms=tot / cnt,
                          ms_per_step=tot / n_steps, alg_bytes=b, alg_flops=f))
    return table


def op_summary(table, op, flags_mask=None, flags_val=None):
    """The biggest launch shape of `op` (by kernel time), with its follower kernel's time added per launch."""
    cand = [t for t in table if t["kernel"] == op and (flags_mask is None or (t["flags"] & flags_mask) == flags_val)]
    if not cand:
        return None
    big = max(cand, key=lambda t: t["ms_per_step"])
    out = dict(big)
    fol = [t for t in table if FOLLOWERS.get(t["kernel"]) == op and t["M"] == big["M"] and t["N"] == big["N"]]
    if fol:
        # the follower serves every launch of the op with this (M, N), whatever its flags: average per launch
        n_op = sum(t["launches_per_step"] for t in table if t["kernel"] == op and t["M"] == big["M"] and t["N"] == big["N"])
        f_ms = sum(t["ms_per_step"] for t in fol) / max(n_op, 1e-9)
        out["follower_avg_ms"] = f_ms
        out["avg_ms_with_follower"] = big["avg_ms"] + f_ms
    t_ms = out.get("avg_ms_with_follower", out["avg_ms"])
    out["hbm_frac"] = (big["alg_bytes"] / (t_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if big["alg_bytes"] else None
    mf = MFMA_PEAK_BY_OP.get(op)
    if op == "linear_fwd" and big["K"] == 256 and big["N"] % 256 == 0:
        mf = MFMA_F16_PEAK_TF / 3            # k_linear_fwd_h3_k256: three f16 products per fp32 product
    out["mfma_frac"] = (big["alg_flops"] / (big["avg_ms"] * 1e-3) / 1e12) / mf if (mf and big["alg_flops"]) else None
    out["mfma_peak_tf"] = mf
    return out


def agg_extra(w, D):
    """Algorithmic bytes of the fused aggregate launches of this workload (ops._agg_bytes: every index once, every distinct
    feature row once, the patient tensor ONCE for the fused relations), keyed like the probe rows: (family, K, flags) with
    K = total (gather) / total 32-padded (scatter) vocab rows of the fused relations; flags: 1 accumulate, 8 rowscale."""
    plan = w["plan"]
    P = plan.n_rows
    extra = {"pair_rows": P + int(plan.num_nodes.get("lab", 0)), "pair_visited": int(w["sup"].sum())}
    rin = plan.rels_into_patient()

    def rel_bytes(rels, rowscale, colscale):
        return sum(4 * r.n_edges + 4 * (P + 1) + 4 * D * r.n_cols + (4 * P if rowscale else 0) +
                   (4 * r.n_cols if colscale else 0) for r in rels)

    if rin:
        sets = [rin] + [[r] for r in rin]
        for rels in sets:
            kg, ks = sum(r.n_cols for r in rels), sum((r.n_cols + 31) // 32 * 32 for r in rels)
            for acc in (0, 1):
                extra[("gather_rows", kg, acc)] = rel_bytes(rels, True, False) + 4 * D * P * (2 if acc else 1)
            # (the strip kernel packs the relations' items back to back: its padded row count is the instance's)
            tiles = (kg + 31) // 32
            k_strip = next((i * 32 for i in (4, 8, 9, 10, 11) if tiles <= i), ks)
            extra[("scatter_rows", k_strip, 0)] = rel_bytes(rels, False, True) + 4 * D * P
            extra[("scatter_rows", ks, 0)] = rel_bytes(rels, False, True) + 4 * D * P
            extra[("scatter_rows", ks, 8)] = rel_bytes(rels, True, False) + 4 * D * P
    return extra


def load_profile(args, scale, strong, what):
    """Committed rocprofv3 PMC records of THIS workload -- keyed on (vocabulary, scale, width), never borrowed from another
    one -- or ({}, None).  what = 'traffic' (FETCH_SIZE / WRITE_SIZE passes -> HBM bytes per launch, by kernel symbol) or
    'pmc' (matrix-core / VALU / LDS counters, by kernel symbol)."""
    if strong:
        return {}, None
    for rnd in ("r4", "r3"):               # this round's record of the workload, else the last one's
        name = f"{rnd}_{what}_{args.shape}_x{scale}_d{args.dim}.json"
        path = os.path.join(REPO, "profiles", name)
        if os.path.exists(path):
            break
    else:
        return {}, None
    try:
        d = json.load(open(path))
        return (d.get("all_kernels", d) if what == "traffic" else d), f"profiles/{name}"
    except Exception:
        return {}, None


N_SE, N_SIMD = 32, 1024      # MI355X: 8 XCDs x 4 shader engines; 256 CUs x 4 SIMDs


def pmc_fractions(pc, src, name):
    """Issue fractions of a kernel from its rocprofv3 --pmc record, PER SIMD: SQ_BUSY_CYCLES is summed over the 32 shader
    engines (busy cycles / 32 = the kernel's duration in shader clocks: 40-70 us kernels come out at 1.7-1.95 GHz),
    SQ_VALU_MFMA_BUSY_CYCLES and SQ_ACTIVE_INST_VALU (one count per vector instruction = 4 issue cycles of a 64-lane wave on
    a 16-lane SIMD) are summed over the 1,024 SIMDs.  Both fractions are asserted to be in (0, 1]."""
    if not pc or not pc.get("SQ_BUSY_CYCLES"):
        return {}
    kernel_cycles = pc["SQ_BUSY_CYCLES"] / N_SE
    out = {"pmc_source": src, "kernel_cycles": kernel_cycles}
    if pc.get("SQ_ACTIVE_INST_VALU") is not None:
        out["valu_issue_frac"] = 4.0 * pc["SQ_ACTIVE_INST_VALU"] / (N_SIMD * kernel_cycles)
    if pc.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        out["mfma_busy_frac"] = pc["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * kernel_cycles)
    for k in ("valu_issue_frac", "mfma_busy_frac"):
        if k in out and out[k] > 0:
            assert 0.0 < out[k] <= 1.0, (name, k, out[k])
    return out


def sym_key(sym):
    return sym.replace(" ", "")


def lookup(profile, sym):
    want = sym_key(sym)
    for k, v in profile.items():
        if sym_key(k) == want:
            return v
    return None


def timed_steps(step_fn, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step_fn()
    barrier()
    return time.perf_counter() - t0, out


def make_graphed_step(w, world, warmup_capture, mask_fraction, capture_collectives=None):
    from mmgnn.train import PiecewiseGraphedTrainStep
    return PiecewiseGraphedTrainStep(w["model"], w["plan"], w["pi"], w["li"], w["y"], w["wlab"], w["opt"],
                                     None if mask_fraction is not None else w["sup"], w["comm"],
                                     n_sup_global=None if mask_fraction is not None else w["n_sup"],
                                     warmup=warmup_capture, mask_fraction=mask_fraction,
                                     capture_collectives=capture_collectives)


def measure(args, world, rank, dev, scale, strong, dim, steps, warmup, want_kernels, want_fixed=False, force_comm=False,
            capture_collectives=None):
    w = build_workload(args, world, rank, dev, scale, strong, dim, force_comm=force_comm)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    train_step(w)                      # cold: code objects load, plans and pair lists are built
    train_step(w)
    gstep = None
    if not args.no_graph:
        # one hipGraph on a single GPU; sharded: a chain of hipGraph segments with the all-reduces between them.  The
        # step draws its own supervision subset (a new one per replay, as every epoch of the reference does)
        ok = torch.ones(1, device=dev)
        try:
            gstep = make_graphed_step(w, world, 2 if world == 1 else 1, MASK_FRACTION, capture_collectives)
        except Exception as e:   # capture is an optimisation, never a requirement
            print(f"[bench] rank {rank}: hipGraph capture unavailable ({type(e).__name__}: {e}); timing eager launches",
                  file=sys.stderr)
            ok.zero_()
            gstep = None
        if world > 1:
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)     # all ranks or none
        if float(ok) == 0.0:
            w["model"]._seed_dev = None
            gstep = None

    def eager_step():                  # (eager fallback: the subset is redrawn on the device here as well)
        w["sup"] = torch.rand(w["pi"].numel(), device=dev) < MASK_FRACTION
        w["supf"] = w["sup"].float()
        n = w["supf"].sum(dtype=torch.float64).reshape(1)
        if w["comm"] is not None:
            torch.distributed.all_reduce(n)
        w["n_sup"] = max(float(n), 1.0)
        return train_step(w)

    step_fn = (lambda: gstep.step()) if gstep is not None else eager_step
    for _ in range(max(warmup, 1)):
        step_fn()

    # ---- timed region: exactly K steps
    dt, loss = timed_steps(step_fn, steps, barrier)
    loss_value = float(loss.detach())
    n_items = (sum(1 for k, _ in gstep.items if k == "graph"), sum(1 for k, _ in gstep.items if k == "all_reduce"),
               int(gstep.n_collectives)) if gstep is not None else (0, 0, 0)

    # ---- the same K steps with ONE fixed supervision mask (what round 2 reported as its headline): the per-step draw,
    # count and pair-list selection are then outside the step
    dt_fixed = None
    if want_fixed and gstep is not None and world == 1:
        del gstep
        gstep = make_graphed_step(w, world, 1, None)
        for _ in range(3):
            gstep.step()
        dt_fixed, _ = timed_steps(lambda: gstep.step(), steps, barrier)

    # ---- per-kernel durations: eager steps of the same kernels, every big launch with its own HIP event pair.  Each
    # step is queued behind a ~10 ms spin on the device, so that the host (~25 us of Python per launch) runs ahead and the
    # kernels execute back to back, as they do inside the graph.
    rows, rows_shared, n_probe = [], [], 0
    if want_kernels:
        w["model"]._seed_dev = None
        n_probe = max(1, min(steps, 5))
        # one stream for this pass: a kernel that shares the GPU with the vocab-side chain of the side stream (model.py
        # overlaps them in the timed region) would report the time it spent sharing, not its own
        prev = mmodel.OVERLAP_MODE
        mmodel.set_overlap("off")
        try:
            ops.probe_arm(1 << 14)
            for _ in range(n_probe):
                torch.cuda._sleep(20_000_000)
                train_step(w)
            torch.cuda.synchronize()
            rows = ops.probe_read()
            # ... and the same steps the way the timed region runs them (vocab-side chain on the side stream): a kernel that
            # shares the chip with the other stream's kernels reports the time it spent sharing -- `in_step_us`
            mmodel.set_overlap(prev)
            if prev == "on" or (prev == "auto" and w["plan"].n_rows >= 16384):      # (model._Run's own rule)
                ops.probe_arm(1 << 14)
                for _ in range(n_probe):
                    torch.cuda._sleep(20_000_000)
                    train_step(w)
                torch.cuda.synchronize()
                rows_shared = ops.probe_read()
        finally:
            mmodel.set_overlap(prev)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    edges = torch.tensor([float(w["E"])], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(edges)
    launch = ("eager" if step_fn is eager_step else
              "hipGraph replay" if n_items[2] == 0 else
              f"ONE hipGraph per step with its {n_items[2]} all-reduces recorded inside" if n_items[1] == 0 else
              f"{n_items[0]} hipGraph segments + {n_items[1]} all-reduces per step")
    rec = dict(dt=float(tmax), edges=float(edges), loss=loss_value, rows=rows, rows_shared=rows_shared, n_probe=n_probe,
               launch=launch,
               dt_fixed=dt_fixed, P_loc=int(w["plan"].n_rows), pairs=int(w["pi"].numel()), extra=agg_extra(w, dim))
    if w["comm"] is not None:          # the captured step goes before anything else of this group does (ShardComm.close)
        w["comm"].close(*([gstep] if gstep is not None else []), destroy=False)
    del w, gstep, step_fn
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def measure_trainer(args, dev, scale, dim, steps):
    """The step through the reference's call surface: mmgnn.train.EdgeMasker + Trainer (src/train.py:37-176, 183-561) on the
    same synthetic graph -- Trainer.train_epoch's captured device step alone, the captured validation pass, and the loop
    body of Trainer.train (train epoch + validation + ONE host read of both losses)."""
    from mmgnn.train import EdgeMasker, Trainer
    from mmgnn.synth import EICU, MIMIC_LIKE
    g = make_graph(scale, seed=0, device=dev, shape=MIMIC_LIKE if args.shape == "mimic" else EICU)
    cfg = dict(model_config(args, dim))
    cfg["train"] = {"optimizer": {"type": "adam", "lr": 1e-3, "weight_decay": 1e-5},
                    "lr_scheduler": {"enabled": True, "type": "reduce_on_plateau", "factor": 0.5, "patience": 10},
                    "loss": "mae", "epochs": steps, "early_stopping_patience": 10 ** 9, "train_split": 0.7, "val_split": 0.15,
                    "test_split": 0.15, "mask_fraction": MASK_FRACTION, "seed": 42}
    cfg["logging"] = {"save_checkpoints": False, "log_interval": 0}
    masker = EdgeMasker(g, 0.7, 0.15, 0.15, MASK_FRACTION, 42)          # no generator: the reference's per-epoch redraw
    torch.manual_seed(42)
    model = build_model(cfg, (g.node_types, g.edge_types), None)
    trainer = Trainer(model, g, masker, cfg, dev)                       # moves model and graph (train.py:605-622)
    for _ in range(3):
        trainer._train_epoch_device()
        trainer._validate_device("val")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer._train_epoch_device()
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer._validate_device("val")
    torch.cuda.synchronize()
    t_val = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):                 # the loop body of Trainer.train: two replays, one host read
        trainer._train_epoch_device()
        trainer._validate_device("val")
        tl, vl = trainer._losses.tolist()
    torch.cuda.synchronize()
    t_epoch = (time.perf_counter() - t0) / steps
    rec = {"train_epoch_ms": 1e3 * t_train, "validate_ms": 1e3 * t_val, "epoch_ms": 1e3 * t_epoch, "epochs_timed": steps,
           "train_loss": tl, "val_loss": vl, "train_pairs": int(trainer._dstep.pi.numel()),
           "val_pairs": int(trainer._deval["val"].pi.numel()),
           "path": "mmgnn.train.Trainer._train_epoch_device / _validate_device: what Trainer.train_epoch, Trainer.validate and "
                   "Trainer.train run; the supervision mask is drawn inside the captured step (masker without a generator)"}
    del trainer, model, masker, g
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def measure_chain(args, dev, scale, dim, steps):
    """The step as a sharded run launches it -- hipGraph segments cut at every collective, RCCL all-reduces between them --
    on ONE GPU (world_size-1 `nccl` group): its time minus the single graph's is the fixed cost of the launch scheme."""
    import socket
    if not torch.distributed.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # an own store, not init_method="tcp://...": under torch.distributed.run with one rank (TORCHELASTIC_USE_AGENT_STORE
        # in the environment) the tcp:// handler connects as a CLIENT of the agent's store and waits on this port forever
        store = torch.distributed.TCPStore("127.0.0.1", port, 1, True)
        torch.distributed.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=dev)
    try:
        rec = measure(args, 1, 0, dev, scale, False, dim, steps, 3, False, force_comm=True)     # the default scheme
        rec["segments"] = measure(args, 1, 0, dev, scale, False, dim, steps, 3, False, force_comm=True,
                                  capture_collectives=False)                              # round 3's: cut at every collective
    finally:
        mdist.ShardComm().close()          # collect -> synchronise -> (barrier) -> destroy: the one teardown order
    return rec


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON record: libraries that chat on fd 1 (RCCL prints a version banner when a
    # communicator is created) are sent to stderr for the whole run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    mmodel.set_overlap(args.overlap)
    if args.next_bn is not None:
        mmodel.set_next_bn([] if args.next_bn == "none" else [t for t in args.next_bn.split(",") if t])
    world, rank, dev = setup_dist(args)
    head = measure(args, world, rank, dev, args.scale, args.strong, args.dim, args.steps, args.warmup, not args.no_kernels,
                   want_fixed=not args.no_extras)
    strong_recs = {}
    if not args.no_strong_x1000 and not (args.strong and args.scale == args.strong_scale):
        s_steps = max(3, min(args.steps, 5))
        for d in sorted({256, args.dim}, reverse=True):          # config 4 (256-d) and north_star's 128-d curve
            r = measure(args, world, rank, dev, args.strong_scale, True, d, s_steps, 2, False)
            r["steps"] = s_steps
            strong_recs[d] = r
    trainer_rec = chain_rec = None
    if world == 1 and not args.no_extras and not args.no_graph:
        try:
            trainer_rec = measure_trainer(args, dev, args.scale, args.dim, max(5, min(args.steps, 30)))
        except Exception as e:
            trainer_rec = {"error": f"{type(e).__name__}: {e}"}
        try:
            chain_rec = measure_chain(args, dev, args.scale, args.dim, max(5, min(args.steps, 30)))
        except Exception as e:
            chain_rec = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        dt, total_edges = head["dt"], head["edges"]
        ms_per_step = 1e3 * dt / args.steps
        out = {
            "metric": "has_lab edges/s, one full training step (new 20 % supervision mask + fwd + weighted-MAE loss + bwd + "
                      "Adam) of predict_lab_values on the full hetero-graph; message passing over every edge, the two edge "
                      "heads evaluated on the supervised pairs (the only predictions the loss reads, train.py:366-370)",
            "value": total_edges * args.steps / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (("eICU-shape" if args.shape == "eicu" else "MIMIC-vocabulary (50 / 200 / 100)")
                                    + f" synthetic hetero-graph x{args.scale}"
                                    + (" total, patient-sharded" if args.strong else " per GPU")
                                    + f", {args.dim}-d, 2 SAGE layers x 6 relations, dropout {args.dropout}, supervision "
                                      f"mask ({MASK_FRACTION:.0%}) redrawn inside every step"),
                       "patients_per_gpu": head["P_loc"], "has_lab_edges_total": int(total_edges),
                       "train_pairs_rank0": head["pairs"], "hidden_dim": args.dim,
                       "parallelism": f"patient-shard x{world}" if world > 1 else "single GPU",
                       "ranks": world, "collective_backend": (os.environ.get("MMG_DIST_BACKEND", "nccl") + (
                           " (RCCL)" if os.environ.get("MMG_DIST_BACKEND", "nccl") == "nccl" else "")) if world > 1 else None,
                       "launch": head["launch"]},
            "loss": head["loss"],
        }
        if head.get("dt_fixed") is not None:
            out["fixed_supervision_mask"] = {
                "ms_per_step": 1e3 * head["dt_fixed"] / args.steps,
                "note": "the same replays with ONE resident supervision mask (round 2's headline): the draw, its count and "
                        "the backward's pair-list selection are outside the step"}
        if trainer_rec is not None:
            if "train_epoch_ms" in trainer_rec:
                trainer_rec["train_epoch_vs_headline"] = trainer_rec["train_epoch_ms"] / ms_per_step
            out["trainer_epoch"] = trainer_rec
        if chain_rec is not None:
            if "dt" in chain_rec:
                cms = 1e3 * chain_rec["dt"] / max(5, min(args.steps, 30))
                out["chain_overhead_ms"] = cms - ms_per_step
                out["chain"] = {"ms_per_step": cms, "launch": chain_rec["launch"],
                                "note": "world_size-1 RCCL group on this GPU: the sharded step as a multi-GPU run launches it "
                                        "(all-reduces recorded inside the hipGraph when RCCL allows, else segment replays + "
                                        "all-reduces issued from Python), without any shard imbalance or wire time"}
                seg = chain_rec.get("segments")
                if seg and "dt" in seg:
                    sms = 1e3 * seg["dt"] / max(5, min(args.steps, 30))
                    out["chain"]["segment_chain"] = {"ms_per_step": sms, "overhead_ms": sms - ms_per_step,
                                                     "launch": seg["launch"]}
            else:
                out["chain"] = chain_rec
        if head["rows"]:
            table = kernel_table(head["rows"], head["n_probe"], head["extra"])
            traffic, tsrc = load_profile(args, args.scale, args.strong, "traffic")
            pmc, psrc = load_profile(args, args.scale, args.strong, "pmc")
            fam = {}
            for t in table:
                fam[FOLLOWERS.get(t["kernel"], t["kernel"])] = fam.get(FOLLOWERS.get(t["kernel"], t["kernel"]), 0.0) + t["ms_per_step"]
            dominant = max(fam.items(), key=lambda kv: kv[1])[0]
            if args.profile_ops:
                for t in sorted(table, key=lambda t: -t["ms_per_step"]):
                    print(f"  {t['kernel']:20s} M={t['M']:8d} N={t['N']:4d} K={t['K']:4d} fl={t['flags']:3d} "
                          f"x{t['launches_per_step']:5.1f}/step  {1e3 * t['avg_ms']:8.1f} us  {t['ms_per_step']:7.3f} ms/step  "
                          f"{t['symbol']}", file=sys.stderr)
            d = op_summary(table, dominant)
            t_ms = d.get("avg_ms_with_follower", d["avg_ms"])
            hbm_frac = d["hbm_frac"] or 0.0
            mfma_frac = d["mfma_frac"] or 0.0
            if mfma_frac > hbm_frac:
                roof = {"bound": "mfma", "achieved": d["alg_flops"] / (d["avg_ms"] * 1e-3) / 1e12,
                        "peak": d.get("mfma_peak_tf") or MFMA_PEAK_BY_OP[dominant], "unit": "TFLOP/s", "frac": mfma_frac,
                        "traffic": None}
            else:
                roof = {"bound": "hbm", "achieved": d["alg_bytes"] / (t_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": hbm_frac, "traffic": None}
            tr = lookup(traffic, d["symbol"])
            if tr and d["alg_bytes"] and 0.5 < tr["hbm_bytes_per_launch"] / d["alg_bytes"] < 3.0:
                roof["traffic"] = tr["hbm_bytes_per_launch"]
                roof["traffic_source"] = tsrc + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, 2*FETCH+WRITE)"
            roof.update({"kernel": d["symbol"], "family": dominant, "shape": [d["M"], d["N"], d["K"]], "flags": d["flags"],
                         "family_ms_per_step": fam[dominant], "avg_launch_ms": t_ms,
                         "launches_per_step": d["launches_per_step"],
                         "algorithmic_bytes_per_launch": d["alg_bytes"], "algorithmic_flops_per_launch": d["alg_flops"],
                         "timing": "HIP start/stop events attached to each kernel launch (hipExtLaunchKernelGGL), "
                                   f"{head['n_probe']} eager single-stream steps of the same kernels right after the timed "
                                   "region; slab-sum kernels are added to the op they finish"})
            if d["alg_flops"] and d["mfma_frac"] is not None:     # north_star: matrix-core utilisation of the dense transform
                roof["mfma_frac"] = d["mfma_frac"]
                roof["mfma_frac_note"] = ("algorithmic fp32 FLOPs / time / (2.5 PFLOP/s dense bf16 or f16 / the matrix products "
                                          "per fp32 product: 6 in the exact six-term bf16 split, 3 in the f16 form of K = 256)")
            pf = pmc_fractions(lookup(pmc, d["symbol"]), psrc, "roofline")
            if "mfma_busy_frac" in pf:
                roof["mfma_busy_frac"] = pf["mfma_busy_frac"]
                roof["valu_issue_frac"] = pf.get("valu_issue_frac")
                roof["pmc_source"] = psrc
            # self-checks: a kernel cannot take longer than the step that contains it, and a roofline fraction is in (0, 1]
            assert roof["family_ms_per_step"] <= ms_per_step, (roof, ms_per_step)
            assert 0.0 < roof["frac"] <= 1.0, roof
            out["roofline"] = roof
            shared = {}
            if head.get("rows_shared"):
                for t in kernel_table(head["rows_shared"], head["n_probe"], head["extra"]):
                    shared[(t["kernel"], t["M"], t["N"], t["K"], t["flags"])] = t["avg_ms"]
            ks = {}
            for name, op, fm, fv in (("gather", "gather_rows", None, None), ("scatter", "scatter_rows", 8, 0),
                                     ("scatter_rowscale", "scatter_rows", 8, 8), ("linear_fwd", "linear_fwd", 16 | 64, 0),
                                     ("linear_bnbwd", "linear_fwd", 16, 16),
                                     ("linear_wgrad", "linear_wgrad", None, None), ("pair_head_fwd", "pair_head_fwd", None, None),
                                     ("pair_head_bwd", "pair_head_bwd", None, None)):
                s_ = op_summary(table, op, fm, fv)
                if s_ is None:
                    continue
                e = {"kernel": s_["symbol"], "avg_launch_us": 1e3 * s_.get("avg_ms_with_follower", s_["avg_ms"]),
                     "launches_per_step": s_["launches_per_step"], "ms_per_step": fam.get(op) if fm is None else s_["ms_per_step"],
                     "shape": [s_["M"], s_["N"], s_["K"]]}
                if "follower_avg_ms" in s_:
                    e["slab_sum_us"] = 1e3 * s_["follower_avg_ms"]
                sh = shared.get((s_["kernel"], s_["M"], s_["N"], s_["K"], s_["flags"]))
                if sh is not None:       # the same launch while the side stream's kernels run beside it (as in the timed step)
                    e["in_step_us"] = 1e3 * (sh + s_.get("follower_avg_ms", 0.0))
                    if s_["alg_bytes"]:
                        e["in_step_hbm_frac"] = (s_["alg_bytes"] / (e["in_step_us"] * 1e-6) / 1e9) / HBM_PEAK_GBS
                if s_["alg_bytes"]:
                    e["algorithmic_bytes"] = s_["alg_bytes"]
                    e["hbm_frac"] = s_["hbm_frac"]
                    assert 0.0 < e["hbm_frac"] <= 1.0, (name, e)
                if s_["alg_flops"]:
                    e["algorithmic_flops"] = s_["alg_flops"]
                    if s_["mfma_frac"] is not None:
                        e["mfma_frac"] = s_["mfma_frac"]
                tr = lookup(traffic, s_["symbol"])
                if tr and s_["alg_bytes"] and 0.5 < tr["hbm_bytes_per_launch"] / s_["alg_bytes"] < 3.0:
                    e["traffic"] = tr["hbm_bytes_per_launch"]
                    folk = {"scatter_rows": "mmg_k_reduce_slabs<EpiScatter>", "linear_wgrad": "mmg_k_reduce_slabs<EpiStore>"}.get(op)
                    fol = lookup(traffic, folk) if folk else None
                    if fol and "slab_sum_us" in e:
                        e["traffic"] += fol["hbm_bytes_per_launch"]       # the slab sum that finishes the op
                    e["traffic_source"] = tsrc
                e.update(pmc_fractions(lookup(pmc, s_["symbol"]), psrc, name))
                ks[name] = e
            out["kernels"] = ks
            out["kernel_ms_per_step_all_probed"] = sum(fam.values())
        for d_, r in strong_recs.items():
            key = "strong_x1000" if d_ == 256 else f"strong_x1000_d{d_}"
            out[key] = {
                "value": r["edges"] * r["steps"] / r["dt"], "unit": "edges/s",
                "ms_per_step": 1e3 * r["dt"] / r["steps"], "steps": r["steps"],
                "scaling": "strong", "n_gpus": world, "patients_per_gpu": r["P_loc"], "hidden_dim": d_,
                "has_lab_edges_total": int(r["edges"]), "launch": r["launch"],
                "workload": f"ONE eICU-shape x{args.strong_scale} graph, patient-sharded over {world} GPU(s), {d_}-d"
                            + (" (BASELINE.json config 4)" if d_ == 256 and args.strong_scale == 1000 else "")}
            # the same two figures inside `config`, where a parser that keeps only the contract's keys still finds the
            # STRONG-scaling curve of north_star (one x1000 graph over the N ranks) next to the weak headline
            ck = f"strong_x{args.strong_scale}" + ("" if d_ == 256 else f"_d{d_}")
            out["config"][ck + "_ms_per_step"] = out[key]["ms_per_step"]
            out["config"][ck + "_edges_per_s"] = out[key]["value"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        mdist.ShardComm().close()          # every rank idle and at the same point before the group goes


def cpu_only():
    """`python bench.py --cpu-only --cpu-scale S`: one oracle step (no warm-up) on the x<S> graph -> one JSON line; the
    record kept as profiles/r3_cpu_x100.json comes from this (S = 100)."""
    args = parse()
    cores = host_cores()
    r = cpu_sample(args, args.cpu_scale, 0, 1, cores)
    print(json.dumps({"value": r["edges_per_s"], "unit": "has_lab edges/s", "s_per_step": r["s_per_step"], "cores": cores,
                      "cpu_model": cpu_model_name(), "scale": r["scale"], "has_lab_edges": r["has_lab_edges"],
                      "warmup": 0, "timed": 1, "dim": args.dim, "dropout": args.dropout}))


if __name__ == "__main__":
    if "--cpu-only" in sys.argv:
        sys.argv.remove("--cpu-only")
        cpu_only()
    else:
        main()
