#!/usr/bin/env python3
"""Benchmark of the hot path: has_lab edges/s for one full training step
(fwd + weighted-MAE loss + bwd + Adam) of ``predict_lab_values`` on a synthetic eICU-shape hetero-graph.

    python bench.py --gpus N --steps K --warmup W [--scale S] [--dim D] [--strong]

One process per GPU (N>1: launched by torch.distributed.run, RCCL).  Workload at N GPUs:
  default (weak): every GPU holds an  x<scale>  eICU-shape shard (1,834*scale patients, 61,484*scale
                  has_lab edges, 128-d) of ONE global graph of N shards; vocab nodes shared.
                  N=1, scale=100 is BASELINE.json configs[2].
  --strong      : ONE global x<scale> graph, patient-sharded over the N GPUs.
Prints ONE JSON line (rank 0) with the whole-job rate, the roofline of the dominant kernel (HIP events
on the launch stream: the start / stop events hipExtLaunchKernelGGL attaches to each launch of that
kernel, read back through mmg_probe_*; an eager re-run of the same kernels when the timed region was a
hipGraph replay) and a CPU baseline (the oracle restatement, timed on this host's cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import mmgnn  # noqa: E402,F401
from mmgnn import dist as mdist  # noqa: E402
from mmgnn import ops  # noqa: E402
from mmgnn.data import build_plan  # noqa: E402
from mmgnn.model import build_model  # noqa: E402
from mmgnn.synth import make_graph  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3     # dense fp32 MFMA (the pair-head kernels)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA
# the dense layers and the aggregates compute fp32 products as exact bf16 pieces on the bf16 matrix cores: 6 matrix
# FLOP per algorithmic fp32 FLOP (linear layers), 3 (0/1-indicator aggregates)
MFMA_PEAK_BY_OP = {"linear_fwd": MFMA_BF16_PEAK_TF / 6, "linear_wgrad": MFMA_BF16_PEAK_TF / 6,
                   "gather_rows": MFMA_BF16_PEAK_TF / 3, "scatter_rows": MFMA_BF16_PEAK_TF / 3}
LAB = ("patient", "has_lab", "lab")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=100, help="eICU-shape multiples per GPU (weak) or in total (--strong)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--strong", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scale", type=int, default=10)
    ap.add_argument("--profile-ops", action="store_true", help="print the per-op time table to stderr")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replays")
    return ap.parse_args()


def setup_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
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


def build_workload(args, world, rank, dev):
    cfg = {"model": {"architecture": "RGCN", "hidden_dim": args.dim, "num_layers": 2, "dropout": args.dropout,
                     "use_batch_norm": True, "activation": "relu"}}
    comm = mdist.ShardComm() if world > 1 else None
    if args.strong and world > 1:
        g_all = make_graph(args.scale, seed=0, device=dev)
        w = mdist.patient_weights(g_all)
        b = mdist.partition_rows(w, world)
        lo, hi = b[rank], b[rank + 1]
        g = mdist.shard_graph(g_all, lo, hi)
        n_global = int(g_all["patient"].num_nodes)
        del g_all
    else:
        g = make_graph(args.scale, seed=1000 * rank, device=dev)     # this rank's shard of the global graph
        P = int(g["patient"].num_nodes)
        lo, hi, n_global = rank * P, (rank + 1) * P, world * P
    torch.manual_seed(42)
    model = build_model(cfg, (g.node_types, g.edge_types), None).to(dev)
    model._init_embeddings(g)
    plan = build_plan(g, dev)
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
    pi, li = ei[0][tr].contiguous(), ei[1][tr].contiguous()
    y = g[LAB].edge_attr[tr].squeeze(-1).contiguous()
    sup = torch.rand(tr.numel(), generator=torch.Generator(device=dev).manual_seed(1234 + rank), device=dev) < 0.2
    wlab = torch.ones(int(g["lab"].num_nodes), device=dev)
    # F5: embeddings are not in the optimizer.  fused = one multi-tensor kernel for all ~60 parameter tensors
    opt = torch.optim.Adam([p for n, p in model.named_parameters() if not n.startswith("embeddings.")],
                           lr=1e-3, weight_decay=1e-5, capturable=True, fused=True)
    n_sup_global = torch.tensor([float(sup.sum())], device=dev)
    if comm is not None:
        torch.distributed.all_reduce(n_sup_global)
    return dict(model=model, plan=plan, g=g, pi=pi, li=li, y=y, sup=sup, supf=sup.float(), wpair=wlab[li].contiguous(),
                wlab=wlab, opt=opt, E=E,
                n_sup=float(n_sup_global), comm=comm)


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


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference) on a bounded sample of the same workload."""
    from oracle import model as om, train as ot
    s = max(1, min(args.cpu_scale, args.scale))
    g = make_graph(s, seed=0, device="cpu")
    gv = om.GraphView(g)
    sd = om.init_state(gv.num_nodes, gv.edge_types, args.dim, 2, seed=42)
    ei, ea = g[LAB].edge_index, g[LAB].edge_attr
    E = ei.shape[1]
    tr = torch.randperm(E, generator=torch.Generator().manual_seed(42))[: int(0.7 * E)].sort().values
    pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
    sup = torch.rand(tr.numel(), generator=torch.Generator().manual_seed(1234)) < 0.2
    wl = torch.ones(gv.num_nodes["lab"])
    cores = host_cores()
    torch.set_num_threads(cores)
    times = []
    n_steps = 4 if s <= 1 else 3
    for i in range(n_steps):
        t0 = time.perf_counter()
        ot.train_step_grads(sd, gv, pi, li, y, wl, sup, p=args.dropout)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": E / t, "unit": "has_lab edges/s", "cores": cores, "kind": "port",
            "sample": f"oracle (pure PyTorch CPU restatement) fwd+loss+bwd on the x{s} eICU-shape graph "
                      f"({E} has_lab edges, {args.dim}-d, dropout {args.dropout}), median of {len(times) - 1} steps "
                      f"after 1 warm-up, {cores} threads"}


def main():
    args = parse()
    world, rank, dev = setup_dist(args)
    w = build_workload(args, world, rank, dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- one eager step, profiled per op, finds the dominant kernel (and warms every cache)
    train_step(w)                      # cold: code objects load, plans and pair lists are built
    prof = ops.OpProfiler()
    ops.set_profiler(prof)
    train_step(w)
    table = prof.summary()
    ops.set_profiler(None)
    dominant = max(table.items(), key=lambda kv: kv[1]["ms"])[0]
    if args.profile_ops and rank == 0:
        tot = sum(v["ms"] for v in table.values())
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
            print(f"  {k:18s} calls {v['calls']:3d}  {v['ms']:9.3f} ms ({100 * v['ms'] / tot:5.1f}%)  "
                  f"{gbs:8.1f} GB/s alg  {tf:6.2f} TFLOP/s", file=sys.stderr)

    # ---- the whole step as ONE hipGraph (single GPU): launch overhead leaves the timed region
    gstep = None
    if world == 1 and not args.no_graph:
        try:
            from mmgnn.train import PiecewiseGraphedTrainStep
            gstep = PiecewiseGraphedTrainStep(w["model"], w["plan"], w["pi"], w["li"], w["y"], w["wlab"], w["opt"], w["sup"],
                                              None, n_sup_global=w["n_sup"], warmup=2)
        except Exception as e:   # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            w["model"]._seed_dev = None
            gstep = None
    elif world > 1 and not args.no_graph:
        # sharded: a chain of hipGraph segments with the all-reduces between them (nothing RCCL-specific is captured)
        ok = torch.ones(1, device=dev)
        try:
            from mmgnn.train import PiecewiseGraphedTrainStep
            gstep = PiecewiseGraphedTrainStep(w["model"], w["plan"], w["pi"], w["li"], w["y"], w["wlab"], w["opt"], w["sup"],
                                              w["comm"], n_sup_global=w["n_sup"], warmup=1)
        except Exception as e:
            print(f"[bench] rank {rank}: piecewise capture unavailable ({type(e).__name__}: {e})", file=sys.stderr)
            ok.zero_()
            gstep = None
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)     # all ranks or none
        if float(ok) == 0.0:
            w["model"]._seed_dev = None
            gstep = None
    step_fn = (lambda: gstep.step()) if gstep is not None else (lambda: train_step(w))
    for _ in range(max(args.warmup, 1)):
        step_fn()
    if os.environ.get("MMG_BENCH_DEBUG"):
        torch.cuda.synchronize()
        print("[debug] loss after warm-up", float(step_fn().detach()) if gstep is None else float(gstep.loss), file=sys.stderr)

    # ---- timed region: exactly K steps
    prof = ops.OpProfiler(only=[dominant])
    if gstep is None:
        ops.set_profiler(prof)           # eager: the dominant op carries HIP events inside the timed region
        ops.probe_arm(1 << 16)           # ... and its kernel launches their own start / stop events
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step_fn()
        if os.environ.get("MMG_BENCH_DEBUG"):
            torch.cuda.synchronize()
            print("[debug] step loss", float(loss.detach()), "seed", int(w["model"]._seed_dev) if w["model"]._seed_dev is not None else None, file=sys.stderr)
    barrier()
    dt = time.perf_counter() - t0
    ops.set_profiler(None)
    loss_value = float(loss.detach())
    if gstep is not None:
        # graph replays cannot carry per-kernel events: time the SAME kernels in K eager steps right after.  Each step is
        # queued behind a ~10 ms spin on the device, so that the host (≈25 us of Python per launch) runs ahead and the
        # kernels execute back to back: an event pair then brackets the kernel, not the host's gap before its launch.
        w["model"]._seed_dev = None
        ops.set_profiler(prof)
        ops.probe_arm(1 << 16)
        for _ in range(args.steps):
            torch.cuda._sleep(20_000_000)
            train_step(w)
        torch.cuda.synchronize()
        ops.set_profiler(None)
    probed = ops.probe_read()            # (ms, M, N, K, flags) per launch of the bf16-split dense forward
    dom = prof.summary()[dominant]
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    edges = torch.tensor([float(w["E"])], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(edges)
    dt = float(tmax)
    total_edges = float(edges)

    if rank == 0:
        value = total_edges * args.steps / dt
        # the dominant KERNEL = the launch shape of the dominant op that takes the most time (an op such as
        # linear_fwd also serves the tiny vocab-side tables with a different kernel: those do not dilute the figure)
        shape = max(dom["shapes"].values(), key=lambda d: d["ms"])
        launches = shape["calls"]
        avg_ms = bracket_ms = shape["ms"] / launches
        bytes_per_launch = shape["bytes"]
        flops_per_launch = shape["flops"]
        timing = ("HIP events on the launch stream, eager re-run of the same kernels right after the graph-replay timed "
                  "region" if gstep is not None else "HIP events on the launch stream inside the timed region")
        if dominant == "linear_fwd" and probed:
            # the same launches timed by the event pair that hipExtLaunchKernelGGL attaches to the kernel itself (its
            # begin / end timestamps, what rocprofv3 reports); the bracketing event pair above also contains the two
            # extra queue entries and the dispatch latency of the kernel
            groups = {}
            for ms_, M_, N_, K_, fl_ in probed:
                b_ = 4 * (M_ * K_ + N_ * K_ + M_ * N_ * (2 if fl_ & 1 else 1))
                g_ = groups.setdefault((b_, 2 * M_ * N_ * K_), [0, 0.0])
                g_[0] += 1
                g_[1] += ms_
            key = (bytes_per_launch, flops_per_launch)
            if key in groups:
                launches, avg_ms = groups[key][0], groups[key][1] / groups[key][0]
                timing = "HIP start/stop events attached to each kernel launch (hipExtLaunchKernelGGL) on its stream, " + \
                         ("eager re-run of the same kernels right after the graph-replay timed region"
                          if gstep is not None else "inside the timed region")
        mfma_peak = MFMA_PEAK_BY_OP.get(dominant, MFMA_F32_PEAK_TF)
        hbm_frac = (bytes_per_launch / (avg_ms * 1e-3) / 1e9) / HBM_PEAK_GBS
        mfma_frac = (flops_per_launch / (avg_ms * 1e-3) / 1e12) / mfma_peak
        if mfma_frac > hbm_frac:
            roof = {"bound": "mfma", "achieved": flops_per_launch / (avg_ms * 1e-3) / 1e12, "peak": mfma_peak,
                    "unit": "TFLOP/s", "frac": mfma_frac, "traffic": None}
        else:
            roof = {"bound": "hbm", "achieved": bytes_per_launch / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hbm_frac, "traffic": None}
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this exact workload (never guessed)
        try:
            if args.scale == 100 and args.dim == 128 and not args.strong:
                tr_ = json.load(open(os.path.join(REPO, "profiles", "r1_traffic_x100.json")))
                roof["traffic"] = tr_["per_kernel"][dominant]["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/r1_traffic_x100.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2*FETCH+WRITE)"
        except Exception:
            roof["traffic"] = None
        roof.update({"kernel": dominant, "op_ms_per_step": dom["ms"] / max(args.steps, 1),
                     "avg_launch_ms": avg_ms, "launches_timed": launches, "timing": timing,
                     "avg_launch_ms_bracketing_events": bracket_ms,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_flops_per_launch": flops_per_launch})
        P_loc = int(w["plan"].n_rows)
        out = {
            "metric": "has_lab edges/s, one full training step (fwd + weighted-MAE loss + bwd + Adam) of "
                      "predict_lab_values on the full hetero-graph",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"eICU-shape synthetic hetero-graph x{args.scale}"
                                    + (" total, patient-sharded" if args.strong else " per GPU")
                                    + f", {args.dim}-d, 2 SAGE layers x 6 relations, dropout {args.dropout}"),
                       "patients_per_gpu": P_loc, "has_lab_edges_total": int(total_edges),
                       "train_pairs_rank0": int(w["pi"].numel()), "hidden_dim": args.dim,
                       "parallelism": f"patient-shard x{world}" if world > 1 else "single GPU",
                       "launch": ("eager" if gstep is None else "hipGraph replay" if world == 1 else
                                  f"{sum(1 for k, _ in gstep.items if k == 'graph')} hipGraph segments + "
                                  f"{sum(1 for k, _ in gstep.items if k == 'all_reduce')} all-reduces per step")},
            "roofline": roof,
            "loss": loss_value,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
