"""N>1 path on CPU: world_size-2 `gloo` runs (one process per shard) of
  * the product's sharding helpers and collectives (mmgnn.dist: partition, shard_graph/pairs/state, ShardComm),
  * the patient-sharded decomposition itself (oracle/sharded.py) against the single-process oracle:
    per-shard predictions, the weighted-MAE loss, every parameter gradient and the Sync-BN running buffers.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fixtures as fx
from oracle import model as om
from oracle import sharded as osh
from oracle import train as ot

SHAPE = (150, 10, 12, 9)
HIDDEN = 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import mmgnn  # noqa: F401
        from mmgnn import dist as md

        g = fx.graph_from_frames(fx.det_frames(*SHAPE))
        gv = om.GraphView(g)
        sd = fx.det_state(gv.num_nodes, HIDDEN)
        ei, ea = g["patient", "has_lab", "lab"].edge_index, g["patient", "has_lab", "lab"].edge_attr
        tr, _, _ = ot.edge_splits(ei.shape[1])
        pi, li, y = ei[0][tr], ei[1][tr], ea[tr].squeeze(-1)
        w = ot.lab_weights(li, y, gv.num_nodes["lab"])
        sup = ot.supervision_mask(int(tr.sum()), 0.2, torch.Generator().manual_seed(3))

        # ---- product helpers: nnz-balanced contiguous partition, shard extraction
        wts = md.patient_weights(g)
        b = md.partition_rows(wts, world)
        assert b[0] == 0 and b[-1] == SHAPE[0] and all(b[i] < b[i + 1] for i in range(world))
        lo, hi = b[rank], b[rank + 1]
        gs = md.shard_graph(g, lo, hi)
        assert gs["patient"].num_nodes == hi - lo
        for et in g.edge_types:
            prow = 0 if et[0] == "patient" else 1
            keep = (g[et].edge_index[prow] >= lo) & (g[et].edge_index[prow] < hi)
            ref = g[et].edge_index[:, keep].clone()
            ref[prow] -= lo
            assert torch.equal(gs[et].edge_index, ref)
        pil, lil, ids = md.shard_pairs(pi, li, lo, hi)
        sds = md.shard_state(sd, lo, hi)
        assert sds["embeddings.patient.weight"].shape[0] == hi - lo
        comm = md.ShardComm(pair_ids=ids)
        n_pairs = torch.tensor([pil.numel()])
        comm.all_reduce(n_pairs)
        assert int(n_pairs) == pi.numel()

        # ---- sharded step
        leaf = {k: (v.detach().clone().requires_grad_(True)
                    if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
                for k, v in sds.items()}
        pred, bufs = osh.predict_sharded(leaf, om.GraphView(gs), pil, lil, SHAPE[0], training=True)
        sup_l, y_l = sup[ids], y[ids]
        n_sup = float(sup.sum())
        loss_part = ((pred[sup_l] - y_l[sup_l]).abs() * w[lil[sup_l]]).sum() / n_sup
        loss_part.backward()
        names = [k for k, v in leaf.items() if torch.is_tensor(v) and v.requires_grad and k != "embeddings.patient.weight"]
        grads = [leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(leaf[k]) for k in names]
        calls0 = comm.n_calls
        views = comm.all_reduce_bucket([g_.clone() for g_ in grads])      # the product's end-of-backward bucket: views of ONE tensor
        assert comm.n_calls == calls0 + 1
        assert [v.shape for v in views] == [g_.shape for g_ in grads]
        assert len({v.untyped_storage().data_ptr() for v in views}) == 1 and all(v.is_contiguous() for v in views)
        comm.all_reduce_list(grads)               # ... and the in-place form (the same sums copied back)
        assert comm.n_calls == calls0 + 2
        for v, g_ in zip(views, grads):
            assert torch.equal(v, g_)
        loss = loss_part.detach().clone()
        dist.all_reduce(loss)

        # ---- single-process oracle (every rank computes it; cheap)
        oloss, opred, ograds, obufs = ot.train_step_grads(sd, gv, pi, li, y, w, sup, p=0.0)
        err = {}
        err["pred"] = float((pred.detach() - opred[ids]).abs().max() / opred.abs().max())
        err["loss"] = abs(float(loss) - float(oloss)) / abs(float(oloss))
        gmax = max(float(v.abs().max()) for v in ograds.values())
        worst = 0.0
        for k, gsum in zip(names, grads):
            worst = max(worst, float((gsum - ograds[k]).abs().max()) / (float(ograds[k].abs().max()) + 1e-3 * gmax))
        gE = leaf["embeddings.patient.weight"].grad
        worst = max(worst, float((gE - ograds["embeddings.patient.weight"][lo:hi]).abs().max())
                    / (float(ograds["embeddings.patient.weight"].abs().max()) + 1e-3 * gmax))
        err["grad"] = worst
        bw = 0.0
        for k, v in bufs.items():
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(obufs[k]), k
            else:
                bw = max(bw, float((v - obufs[k]).abs().max() / obufs[k].abs().max()))
        err["bufs"] = bw
        q.put((rank, err, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_step_matches_single_process_oracle():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err, tb in res:
        assert tb is None, f"rank {rank} failed:\n{tb}"
        assert err["pred"] <= 1e-5, err
        assert err["loss"] <= 1e-5, err
        assert err["grad"] <= 2e-4, err
        assert err["bufs"] <= 1e-5, err


def test_partition_rows_properties():
    import mmgnn  # noqa: F401
    from mmgnn import dist as md
    w = torch.tensor([5, 0, 0, 9, 1, 1, 1, 30, 2, 2], dtype=torch.long)
    for world in (1, 2, 3, 4, 8):
        b = md.partition_rows(w, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == len(w)
        assert all(b[i] <= b[i + 1] for i in range(world))
        if world <= len(w):
            assert all(b[i] < b[i + 1] for i in range(world))       # nobody is left without rows
    b = md.partition_rows(torch.ones(1000, dtype=torch.long), 8)
    sizes = [b[i + 1] - b[i] for i in range(8)]
    assert max(sizes) - min(sizes) <= 1
    assert md.partition_rows(torch.zeros(0, dtype=torch.long), 2) == [0, 0, 0]


def test_edge_masker_shards_keep_the_global_split_and_supervision_draws():
    """EdgeMasker.shard (what a patient-sharded Trainer is given): every has_lab edge stays in the split the global
    permutation gave it, the shards' train-pair ids tile the unsharded train-pair list, and the per-epoch supervision
    subsets of the shards are the global draw (train.py:150-166) cut at the shard boundaries."""
    import mmgnn  # noqa: F401
    from mmgnn import dist as md
    from mmgnn.train import EdgeMasker
    g = fx.graph_from_frames(fx.det_frames(*SHAPE))
    lab = ("patient", "has_lab", "lab")
    src = g[lab].edge_index[0]
    whole = EdgeMasker(g, 0.7, 0.15, 0.15, 0.2, 42, mask_generator=torch.Generator().manual_seed(5))
    b = md.partition_rows(md.patient_weights(g), 3)
    parts = []
    for r in range(3):
        lo, hi = b[r], b[r + 1]
        keep = (src >= lo) & (src < hi)
        m = EdgeMasker(g, 0.7, 0.15, 0.15, 0.2, 42, mask_generator=torch.Generator().manual_seed(5)).shard(
            md.shard_graph(g, lo, hi), keep)
        assert torch.equal(m.train_mask, whole.train_mask[keep]) and torch.equal(m.val_mask, whole.val_mask[keep])
        assert m.num_edges == int(keep.sum()) == m.edge_index.shape[1]
        ei, ev, _, sup = m.get_masked_data("train")
        assert ei.shape[1] == m.train_pair_ids.numel() == sup.numel()
        assert torch.equal(ei[0] + lo, g[lab].edge_index[0][keep & whole.train_mask])
        parts.append((m, sup))
    ids = torch.cat([m.train_pair_ids for m, _ in parts]).sort().values
    assert torch.equal(ids, torch.arange(int(whole.train_mask.sum())))
    _, _, _, sup_all = whole.get_masked_data("train")
    for m, sup in parts:                                     # first epoch's draw; the generators advance alike afterwards
        assert torch.equal(sup, sup_all[m.train_pair_ids])
    _, _, _, sup_all2 = whole.get_masked_data("train")
    for m, _ in parts:
        assert torch.equal(m.get_masked_data("train")[3], sup_all2[m.train_pair_ids])
    assert not torch.equal(sup_all, sup_all2)
