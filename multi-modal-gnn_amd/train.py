"""Caller side of the hot path: the counterparts of the reference's ``src/train.py`` pieces that drive
``predict_lab_values`` (EdgeMasker :37-176, Trainer :183-561).  Same class / method names, argument
meaning and error behaviour; the graph and the model live on the GPU.

Differences that are switches, not silent fixes:
  * ``train_embeddings=False`` (default) reproduces the reference: the optimizer is built from
    ``model.parameters()`` BEFORE the lazy embeddings exist, so they are never updated;
  * the per-epoch supervision mask draws from ``mask_generator`` when given (a CPU generator reproduces the
    reference's ``torch.rand(n) < fraction`` bit for bit; a device generator draws on the device); without one the
    reference seeds from the wall clock (train.py:156) -- the device step then draws the subset inside the captured
    step from the device RNG stream (``mmg_sup_mask_draw``), the eager path from the wall-clock seed as the reference;
  * ``Trainer.train()`` runs every epoch as replays of two hipGraphs (training step, validation forward) with ONE host
    read of the two loss scalars per epoch -- what the scheduler, early stopping and the history need
    (``device_step=False`` keeps the eager autograd path);
  * ``ReduceLROnPlateau`` is built without the ``verbose`` kwarg torch >= 2.7 removed.
"""
from __future__ import annotations

import json
import logging
import time
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .model import build_model, compute_regression_loss

LAB_EDGE = ("patient", "has_lab", "lab")
# Stream capture checks "unsafe" HIP calls of the capturing THREAD only.  With the default ("global") any thread counts, and
# the watchdog thread of a torch.distributed NCCL / RCCL process group polls its work events (hipEventQuery) on its own
# schedule.  (The abort this was introduced for turned out to be something else -- the watchdog querying an event whose
# stream is capturing, which no error mode allows: mmgnn.dist.eager_collective_stream -- but a poll of an unrelated event
# from that thread is still a call "global" mode counts.)
CAPTURE_ERROR_MODE = "thread_local"


def capture_error_mode() -> str:
    """Mode of the next stream capture.  `thread_local` only while a torch.distributed process group is alive -- its
    watchdog thread is the one known source of HIP calls from another thread during a capture window; without a group
    the default ("global") stays, so that a capture-invalidating call from ANY thread is still reported instead of
    silently corrupting the recording."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return CAPTURE_ERROR_MODE
    except Exception:
        pass
    return "global"


class EdgeMasker:
    """train.py:37-176 -- edge-level 70/15/15 split (seed 42) + per-epoch supervision mask."""

    def __init__(self, data, train_split: float = 0.7, val_split: float = 0.15, test_split: float = 0.15,
                 mask_fraction: float = 0.2, seed: int = 42, mask_generator: Optional[torch.Generator] = None):
        self.data = data
        self.train_split, self.val_split, self.test_split = train_split, val_split, test_split
        self.mask_fraction = mask_fraction
        self.seed = seed
        self.mask_generator = mask_generator
        assert abs(train_split + val_split + test_split - 1.0) < 1e-6, "Splits must sum to 1.0"
        self.edge_type = LAB_EDGE
        self.edge_index = data[self.edge_type].edge_index
        self.edge_attr = data[self.edge_type].edge_attr
        self.num_edges = self.edge_index.shape[1]
        self.train_mask, self.val_mask, self.test_mask = self._create_splits()
        self._cache = {}
        logging.info("Edge splits created:")
        logging.info(f"  Train: {self.train_mask.sum()} edges ({100 * train_split:.1f}%)")
        logging.info(f"  Val: {self.val_mask.sum()} edges ({100 * val_split:.1f}%)")
        logging.info(f"  Test: {self.test_mask.sum()} edges ({100 * test_split:.1f}%)")

    def _create_splits(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        # CPU generator on purpose: bit-identical to train.py:110-127 for the same seed
        torch.manual_seed(self.seed)
        np.random.seed(self.seed)
        perm = torch.randperm(self.num_edges)
        n_train = int(self.train_split * self.num_edges)
        n_val = int(self.val_split * self.num_edges)
        masks = [torch.zeros(self.num_edges, dtype=torch.bool) for _ in range(3)]
        masks[0][perm[:n_train]] = True
        masks[1][perm[n_train:n_train + n_val]] = True
        masks[2][perm[n_train + n_val:]] = True
        dev = self.edge_index.device
        return tuple(m.to(dev) for m in masks)

    def to(self, data, device):
        """Follow the graph to `device` (Trainer.__init__ moves it after the masker was built, train.py:605-622)."""
        self.data = data
        self.edge_index = data[self.edge_type].edge_index
        self.edge_attr = data[self.edge_type].edge_attr
        self.train_mask, self.val_mask, self.test_mask = (m.to(device) for m in (self.train_mask, self.val_mask,
                                                                                self.test_mask))
        self._cache = {}
        return self

    def shard(self, shard_data, keep: torch.Tensor) -> "EdgeMasker":
        """This masker restricted to ONE patient shard of the graph it was built on (dist.shard_graph): the same 70/15/15
        split -- every has_lab edge stays in the split the global permutation gave it -- and the same per-epoch
        supervision subsets (the shard draws the GLOBAL mask from its equally seeded generator and keeps its own
        positions), so that N sharded Trainers train exactly what one unsharded Trainer trains.
        keep: bool [num_edges], the has_lab edges of the shard in their original order.  `train_pair_ids` = positions of
        the shard's train pairs in the unsharded train-pair list (the keys of the per-pair dropout / in-step draw)."""
        m = object.__new__(EdgeMasker)
        m.__dict__.update(self.__dict__)
        keep = keep.to(self.train_mask.device)
        m.data = shard_data
        m.edge_index = shard_data[self.edge_type].edge_index
        m.edge_attr = shard_data[self.edge_type].edge_attr
        m.num_edges = int(keep.sum())
        if m.edge_index.shape[1] != m.num_edges:
            raise ValueError("EdgeMasker.shard: `keep` does not select the shard's has_lab edges")
        m.train_mask, m.val_mask, m.test_mask = self.train_mask[keep], self.val_mask[keep], self.test_mask[keep]
        pos = torch.cumsum(self.train_mask.to(torch.int64), 0) - 1
        m.train_pair_ids = pos[keep & self.train_mask].contiguous()
        m._draw_from = (int(self.train_mask.sum()), m.train_pair_ids)
        m._cache = {}
        return m

    def draw_supervision_mask(self, n: int, device=None) -> torch.Tensor:
        """The per-epoch supervision subset of the n train edges (train.py:150-166): torch.rand(n) < mask_fraction.
        mask_generator on the CPU (or none: wall-clock seed, train.py:156) draws on the host exactly like the reference;
        a device generator draws on its device, nothing crosses PCIe."""
        if self.mask_fraction <= 0:
            return torch.ones(n, dtype=torch.bool, device=device)
        frm = getattr(self, "_draw_from", None)
        if frm is not None and n == frm[1].numel():        # a shard: the global draw, this shard's positions of it
            n_all, pos = frm
            self._draw_from = None
            try:
                m = self.draw_supervision_mask(n_all, device)
            finally:
                self._draw_from = frm
            return m[pos.to(m.device)]
        gen = self.mask_generator
        if gen is None:
            torch.manual_seed(int(time.time()))                              # train.py:156
            m = torch.rand(n) < self.mask_fraction
        elif gen.device.type == "cpu":
            m = torch.rand(n, generator=gen) < self.mask_fraction
        else:
            return torch.rand(n, generator=gen, device=gen.device) < self.mask_fraction
        return m if device is None else m.to(device)

    def get_masked_data(self, split: str = "train", want_mask: bool = True):
        """want_mask = False: the fourth value is None (the device step draws / receives its mask separately)."""
        if split == "train":
            mask = self.train_mask
        elif split in ("val", "test"):
            mask = self.val_mask if split == "val" else self.test_mask
        else:
            raise ValueError(f"Unknown split: {split}")
        # the split is static: keep ONE tensor object per split so the model's pair cache (sorted pairs) hits
        if split not in self._cache:
            self._cache[split] = (self.edge_index[:, mask].contiguous(), self.edge_attr[mask].squeeze(-1).contiguous())
        edge_indices, edge_values = self._cache[split]
        supervision_mask = None
        if want_mask:
            n = edge_indices.shape[1]
            if split == "train":
                supervision_mask = self.draw_supervision_mask(n, edge_indices.device)
            else:
                supervision_mask = torch.ones(n, dtype=torch.bool, device=edge_indices.device)
        return edge_indices, edge_values, mask, supervision_mask


class Trainer:
    """train.py:183-561."""

    def __init__(self, model: nn.Module, data, masker: EdgeMasker, config: Dict, device: torch.device,
                 train_embeddings: bool = False, device_step: bool = True):
        self.model = model.to(device)
        self.data = data.to(device)
        self.masker = masker
        # The reference builds the masker from the CPU graph BEFORE Trainer moves the graph (train.py:605-622):
        # re-point it at the moved tensors, or its pairs / masks / lab weights would stay on the host.
        if hasattr(masker, "to"):
            masker.to(self.data, device)
        self.config = config
        self.device = device
        tc = config["train"]
        if train_embeddings and len(self.model.embeddings) == 0:
            self.model._init_embeddings(self.data)          # switch: make the tables visible to the optimizer
        self.optimizer = self._build_optimizer(tc["optimizer"])
        self.scheduler = self._build_scheduler(tc.get("lr_scheduler", {}))
        self.loss_fn = tc["loss"]
        self.epochs = tc["epochs"]
        self.early_stopping_patience = tc["early_stopping_patience"]
        self.best_val_loss = float("inf")
        self.patience_counter = 0
        self.train_losses, self.val_losses = [], []
        # patient-sharded model (dist.shard_model): this Trainer holds ONE shard of the graph; losses, lab weights and the
        # supervision normaliser are global (summed over the group), everything else is local
        self.comm = getattr(self.model, "_comm", None)
        if self.comm is not None and getattr(self.comm, "pair_ids", None) is None and hasattr(masker, "train_pair_ids"):
            self.comm.pair_ids = masker.train_pair_ids.to(device)     # per-pair dropout / in-step draw: global pair ids
        self.lab_weights = self._compute_lab_weights()
        self._pairs = {}
        # the captured device step (built at the first epoch) and the captured validation passes
        self.device_step = bool(device_step)
        self._dstep = None
        self._deval = {}
        self._losses = None          # fp64 [2] on the device: (train loss, validation loss) of the current epoch

    def _build_optimizer(self, oc: Dict) -> optim.Optimizer:
        kind = oc.get("type", "adam").lower()
        if kind == "adam":
            params = list(self.model.parameters())
            if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
                from .optim import Adam                                        # one launch of mmg_adam_step per step
                return Adam(params, lr=oc["lr"], weight_decay=oc["weight_decay"])
            return optim.Adam(params, lr=oc["lr"], weight_decay=oc["weight_decay"])
        if kind == "sgd":
            return optim.SGD(self.model.parameters(), lr=oc["lr"], weight_decay=oc["weight_decay"],
                             momentum=oc.get("momentum", 0.9))
        raise ValueError(f"Unknown optimizer: {kind}")

    def _build_scheduler(self, sc: Dict):
        if not sc.get("enabled", False):
            return None
        kind = sc.get("type", "reduce_on_plateau")
        if kind == "reduce_on_plateau":
            return optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=sc.get("factor", 0.5),
                                                        patience=sc.get("patience", 10))
        if kind == "step":
            return optim.lr_scheduler.StepLR(self.optimizer, step_size=sc.get("step_size", 30), gamma=sc.get("gamma", 0.1))
        raise ValueError(f"Unknown scheduler: {kind}")

    def _compute_lab_weights(self) -> torch.Tensor:
        """train.py:295-330: w_l = L * (1/(var_l + 1e-6)) / sum, unbiased var over TRAIN edges (1.0 if <= 1 sample)."""
        edge_indices, edge_values, _, _ = self.masker.get_masked_data("train")
        lab = edge_indices[1]
        L = int(self.data["lab"].num_nodes)
        v = edge_values.double()
        cnt = torch.zeros(L, dtype=torch.float64, device=v.device).index_add_(0, lab, torch.ones_like(v))
        s1 = torch.zeros(L, dtype=torch.float64, device=v.device).index_add_(0, lab, v)
        if self.comm is not None:              # the statistics are over ALL shards' train edges
            self.comm.raw_all_reduce(cnt)
            self.comm.raw_all_reduce(s1)
        mean = s1 / cnt.clamp(min=1)
        s2 = torch.zeros(L, dtype=torch.float64, device=v.device).index_add_(0, lab, (v - mean[lab]) ** 2)
        if self.comm is not None:
            self.comm.raw_all_reduce(s2)
        var = torch.where(cnt > 1, s2 / (cnt - 1).clamp(min=1), torch.ones_like(s2))
        w = 1.0 / (var + 1e-6)
        w = w * L / w.sum()
        return w.float()

    def _split_pairs(self, split, want_mask: bool = True):
        edge_indices, edge_values, _, sup = self.masker.get_masked_data(split, want_mask=want_mask)
        if split not in self._pairs:          # keep the SAME tensor objects across epochs (pair-cache key)
            self._pairs[split] = (edge_indices[0].contiguous(), edge_indices[1].contiguous())
        pi, li = self._pairs[split]
        return pi, li, edge_values, sup

    # ------------------------------------------------------------------ device step (hipGraph replays)
    def _device_step_ok(self) -> bool:
        from .optim import Adam
        return (self.device_step and torch.device(self.device).type == "cuda" and isinstance(self.optimizer, Adam)
                and self.loss_fn in ("mae", "mse", "huber"))

    def _graph_plan(self):
        """ONE plan (CSR, bit planes, degrees) of the static graph for the captured training step and every captured
        validation pass (held here: the module-level plan cache is bounded and may decline a very large graph)."""
        if getattr(self, "_plan", None) is None:
            from .data import build_plan
            if self.comm is None:
                self._plan = build_plan(self.data, self.device)
            else:
                # this shard of the global graph: contiguous patient ranges in rank order (dist.shard_graph records its
                # range; otherwise the shards are taken back to back), vocab in-degrees summed over the group
                from . import dist as mdist
                plan = build_plan(self.data, self.device, use_cache=False)
                sizes = torch.zeros(int(self.comm.world), dtype=torch.float64, device=self.device)
                sizes[int(self.comm.rank)] = float(plan.n_rows)
                self.comm.raw_all_reduce(sizes)
                rr = getattr(self.data, "row_range", None)
                lo = int(rr[0]) if rr is not None else int(sizes[:int(self.comm.rank)].sum().item())
                self._plan = mdist.shard_plan(plan, self.comm, lo, int(sizes.sum().item()))
        return self._plan

    def _loss_slots(self):
        if self._losses is None:
            self._losses = torch.zeros(2, dtype=torch.float64, device=self.device)
        return self._losses

    def _build_device_step(self):
        """The training step of train_epoch as ONE captured hipGraph (PiecewiseGraphedTrainStep, comm = None).  The
        supervision subset is drawn inside the step when the masker has no generator (the reference's wall-clock seed,
        train.py:156), injected through set_mask otherwise."""
        from .data import build_plan
        pi, li, y, _ = self._split_pairs("train", want_mask=False)
        m = self.masker
        frac = float(m.mask_fraction)
        in_graph = frac > 0 and m.mask_generator is None
        sup0 = None if in_graph else torch.ones(pi.numel(), dtype=torch.bool, device=pi.device)
        plan = self._graph_plan()
        if len(self.model.embeddings) == 0:
            self.model._init_embeddings(self.data)
        self._dstep = PiecewiseGraphedTrainStep(self.model, plan, pi, li, y, self.lab_weights, self.optimizer, sup0,
                                                self.comm, loss_fn=self.loss_fn,
                                                mask_fraction=frac if in_graph else None, loss_out=self._loss_slots()[0],
                                                reduce_loss=self.comm is not None)
        self._dstep_injects = frac > 0 and not in_graph

    def _train_epoch_device(self) -> torch.Tensor:
        """One epoch = one replay; returns the loss as a device scalar (no host synchronisation)."""
        self.model.train()
        if self._dstep is None:
            self._build_device_step()
        if self._dstep_injects:
            self._dstep.set_mask(self.masker.draw_supervision_mask(self._dstep.pi.numel(), self._dstep.pi.device))
        return self._dstep.step()

    def _validate_device(self, split: str, slot: int = 1) -> torch.Tensor:
        self.model.eval()
        ev = self._deval.get(split)
        if ev is None:
            from .data import build_plan
            pi, li, y, _ = self._split_pairs(split, want_mask=False)
            ev = GraphedEval(self.model, self._graph_plan(), pi, li, y, self.loss_fn, loss_out=self._loss_slots()[slot],
                             comm=self.comm)
            self._deval[split] = ev
        return ev.step()

    # ------------------------------------------------------------------ the reference's call surface
    def train_epoch(self) -> float:
        """train.py:332-392.  Returns the loss as a Python float (one host read), as the reference does; train() uses
        the device-resident form and reads both losses of an epoch at once."""
        if self._device_step_ok():
            return float(self._train_epoch_device())
        return self._train_epoch_eager()

    def _train_epoch_eager(self) -> float:
        self.model.train()
        pi, li, y, sup = self._split_pairs("train")
        self.optimizer.zero_grad()
        if self.comm is not None:
            return self._train_epoch_eager_sharded(pi, li, y, sup)
        pred = self.model.predict_lab_values(self.data, pi, li)
        sp, st, sl = pred[sup], y[sup], li[sup]
        if self.loss_fn == "mae":
            per = torch.abs(sp - st)
        elif self.loss_fn == "mse":
            per = (sp - st) ** 2
        else:
            loss = compute_regression_loss(sp, st, loss_type=self.loss_fn)
            loss.backward()
            self.optimizer.step()
            return loss.item()
        loss = (per * self.lab_weights[sl]).mean()
        loss.backward()
        self.optimizer.step()
        return loss.item()

    def _train_epoch_eager_sharded(self, pi, li, y, sup) -> float:
        """The eager epoch of ONE shard: the mean over predictions[supervision_mask] (train.py:366-386) is a global mean --
        the local weighted sum over the global subset size; gradients are exchanged inside the model's backward."""
        from . import ops
        if self.loss_fn not in ("mae", "mse", "huber"):
            raise ValueError(f"Unknown loss type: {self.loss_fn}")
        n = sup.sum(dtype=torch.float64).reshape(1)
        self.comm.raw_all_reduce(n)
        pred = self.model.predict_lab_values(self._graph_plan(), pi, li)
        wl = self.lab_weights[li].contiguous() if self.loss_fn in ("mae", "mse") else None
        loss = ops.weighted_pair_loss(pred, y, wl, sup.float(), 1.0 / max(float(n), 1.0), self.loss_fn)
        loss.backward()
        self.optimizer.step()
        tot = loss.detach().double().reshape(1).clone()
        self.comm.raw_all_reduce(tot)
        return float(tot)

    @torch.no_grad()
    def validate(self, split: str = "val") -> float:
        """train.py:394-431."""
        if split not in ("val", "test"):
            raise ValueError(f"Unknown split: {split}")
        if self._device_step_ok():
            return float(self._validate_device(split))
        return self._validate_eager(split)

    @torch.no_grad()
    def _validate_eager(self, split: str = "val") -> float:
        self.model.eval()
        pi, li, y, _ = self._split_pairs(split)
        if self.comm is not None:              # global mean: local sums over the global pair count
            from . import ops
            pred = self.model.predict_lab_values(self._graph_plan(), pi, li)
            n = torch.tensor([float(pi.numel())], dtype=torch.float64, device=pi.device)
            self.comm.raw_all_reduce(n)
            loss, _ = ops.pair_loss(pred, y, None, None, 1.0 / max(float(n), 1.0), self.loss_fn, want_dpred=False)
            tot = loss.detach().double().reshape(1).clone()
            self.comm.raw_all_reduce(tot)
            return float(tot)
        pred = self.model.predict_lab_values(self.data, pi, li)
        return compute_regression_loss(pred, y, loss_type=self.loss_fn).item()

    def train(self, output_dir: Path) -> Dict:
        """train.py:433-544 (same files: best_model.pt, checkpoint_epoch_N.pt, training_history.json)."""
        output_dir = Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        lc = self.config.get("logging", {})
        history = {"train_loss": [], "val_loss": [], "learning_rates": [], "epoch_times": []}
        on_device = self._device_step_ok()
        for epoch in range(1, self.epochs + 1):
            t0 = time.time()
            if on_device:
                # two graph replays, then ONE host read of both losses: the scheduler, early stopping and the history
                # below are host decisions on these two numbers (train.py:459-495)
                self._train_epoch_device()
                self._validate_device("val")
                tl, vl = self._losses.tolist()
            else:
                tl = self._train_epoch_eager()
                vl = self._validate_eager("val")
            history["train_loss"].append(tl)
            history["val_loss"].append(vl)
            history["learning_rates"].append(self.optimizer.param_groups[0]["lr"])
            history["epoch_times"].append(time.time() - t0)
            self.train_losses.append(tl)
            self.val_losses.append(vl)
            li_ = lc.get("log_interval", 0)
            if li_ and epoch % li_ == 0:
                logging.info(f"Epoch {epoch}/{self.epochs} | Train Loss: {tl:.4f} | Val Loss: {vl:.4f} | "
                             f"Time: {history['epoch_times'][-1]:.2f}s")
            if self.scheduler is not None:
                if isinstance(self.scheduler, optim.lr_scheduler.ReduceLROnPlateau):
                    self.scheduler.step(vl)
                else:
                    self.scheduler.step()
            if vl < self.best_val_loss:
                self.best_val_loss = vl
                self.patience_counter = 0
                if lc.get("save_checkpoints", True):
                    self._save(output_dir / "best_model.pt", epoch, vl)
            else:
                self.patience_counter += 1
            if lc.get("save_checkpoints", True) and epoch % lc.get("checkpoint_interval", 10) == 0:
                self._save(output_dir / f"checkpoint_epoch_{epoch}.pt", epoch, vl)
            if self.patience_counter >= self.early_stopping_patience:
                logging.info(f"Early stopping at epoch {epoch}")
                break
        with open(output_dir / "training_history.json", "w") as f:
            json.dump(history, f, indent=2)
        return history

    def _save(self, path, epoch, val_loss):
        torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": self.optimizer.state_dict(), "val_loss": val_loss,
                    "config": self.config}, path)                                   # train.py:501-509

    def load_best_model(self, output_dir: Path):
        ck = torch.load(Path(output_dir) / "best_model.pt", map_location=self.device, weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"])
        logging.info(f"Loaded best model from epoch {ck['epoch']} (val_loss: {ck['val_loss']:.4f})")


def _snapshot_training_state(model, optimizer):
    """Copies of everything a training step mutates: parameters, BatchNorm buffers, optimizer state tensors."""
    snap = {"model": {k: v.detach().clone() for k, v in model.state_dict().items()}, "opt": {}}
    for p_, st in optimizer.state.items():
        snap["opt"][p_] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
    return snap


def _restore_training_state(model, optimizer, snap):
    """Undo the warm-up / capture steps IN PLACE (the captured graphs hold the addresses): recording a step must not
    train the model.  Optimizer state that did not exist before (created lazily by the first step) is zeroed."""
    with torch.no_grad():
        cur = model.state_dict()
        for k, v in snap["model"].items():
            cur[k].copy_(v)
        for p_, st in optimizer.state.items():
            old = snap["opt"].get(p_)
            for k, v in st.items():
                if torch.is_tensor(v):
                    if old is not None and torch.is_tensor(old.get(k)):
                        v.copy_(old[k])
                    else:
                        v.zero_()
                elif old is not None and k in old:
                    st[k] = old[k]
                elif isinstance(v, (int, float)):
                    st[k] = type(v)(0)


class _SupervisionState:
    """Supervision subset of a captured step: the float mask and the normaliser 1 / n_sup both live on the device and
    are updated in place, so a replay with a new per-epoch mask (train.py:150-176 of the reference) divides by the size
    of THAT subset, exactly like the reference's .mean() over pred[mask] (train.py:366-386)."""

    def __init__(self, sup_mask, comm, n_sup_global, n=None, device=None):
        self.comm = comm
        if sup_mask is None:                     # drawn inside the step (draw()): nothing supervised until then
            self.sup = torch.zeros(int(n), dtype=torch.float32, device=device)
        else:
            self.sup = sup_mask.to(torch.float32).contiguous()
        self.inv_den = torch.ones(1, dtype=torch.float64, device=self.sup.device)
        self.count = torch.zeros(1, dtype=torch.float64, device=self.sup.device)
        if sup_mask is not None:
            self._set_den(n_sup_global)

    def _set_den(self, n_sup_global=None):
        if n_sup_global is not None:
            self.inv_den.fill_(1.0 / max(float(n_sup_global), 1.0))
            return
        n = self.sup.sum(dtype=torch.float64).reshape(1)
        if self.comm is not None:
            self.comm.raw_all_reduce(n)          # outside any capture: set_mask runs between replays
        torch.reciprocal(n.clamp_(min=1.0), out=self.inv_den)

    def set_mask(self, sup_mask, n_sup_global=None):
        self.sup.copy_(sup_mask)                 # (bool -> float inside the copy)
        self._set_den(n_sup_global)

    def draw(self, fraction, seed_dev, ids=None, n_global=None):
        """New subset drawn on the device from the step's seed stream (mmg_sup_mask_draw); capturable.  Sharded: the draw
        is a pure function of (seed, global pair id), so every rank counts the subset of ALL n_global pairs itself (a
        hash-only pass, no memory traffic) instead of summing the shard sizes with a collective."""
        from . import ops
        if self.comm is None:
            ops.sup_mask_draw(self.sup.numel(), fraction, self.sup.device, seed_dev=seed_dev, ids=ids, sup=self.sup,
                              count=self.count, inv_den=self.inv_den)
            return
        if n_global is None:
            raise ValueError("a sharded draw needs the global number of pairs")
        local = torch.empty(1, dtype=torch.float64, device=self.sup.device)
        ops.sup_mask_draw(self.sup.numel(), fraction, self.sup.device, seed_dev=seed_dev, ids=ids, sup=self.sup,
                          count=local, inv_den=None)
        ops.sup_mask_draw(int(n_global), fraction, self.sup.device, seed_dev=seed_dev, count=self.count,
                          inv_den=self.inv_den, count_only=True)


def _new_seed_state(dev) -> torch.Tensor:
    """int64 [2] on the device: (dropout seed the kernels read, position of its SplitMix64 stream).  The stream starts
    from the host generator (torch.manual_seed), so equally seeded ranks draw the same masks; mmg_seed_advance moves it
    on inside the captured step -- nothing runs between two replays."""
    from . import ops
    st = torch.tensor([0, int(torch.randint(0, 2 ** 62, (1,)).item())], dtype=torch.int64, device=dev)
    ops.seed_advance(st)
    return st


class _Recording:
    """Device work recorded into hipGraphs and replayed: ONE graph, or -- for a patient-sharded model whose collectives
    cannot be recorded (ShardComm.capturable() is false: gloo, or an RCCL that refuses) -- a CHAIN of graphs cut at every
    ``ShardComm.all_reduce`` the body meets, with the all-reduces issued eagerly between the replays.  All segments share one
    memory pool (static addresses), are recorded on one thread on the current stream, and replay in recording order.
    ``capture_collectives``: None = ask the communicator; chosen once, never by retrying a failed recording."""

    def __init__(self, comm, capture_collectives: Optional[bool] = None):
        self.comm = comm
        if comm is None:
            self.capture_collectives = False
        elif capture_collectives is None:
            cap = getattr(comm, "capturable", None)
            self.capture_collectives = bool(cap()) if cap is not None else False
        else:
            self.capture_collectives = bool(capture_collectives)
        self.items = []
        self.n_collectives = 0
        self._cur = None
        self._pool = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._pool, capture_error_mode=capture_error_mode())
        self._cur = g

    def _end(self):
        self._cur.capture_end()
        self.items.append(("graph", self._cur))
        self._cur = None

    def _collective(self, t):
        self.n_collectives += 1
        if self.capture_collectives:       # recorded like any kernel of the body (RCCL on the capturing stream)
            self.comm.raw_all_reduce(t)
            return
        self._end()
        self.comm.raw_all_reduce(t)
        self.items.append(("all_reduce", t))
        self._begin()

    def record(self, body):
        """Run `body` once under stream capture on the CURRENT stream (the caller has warmed it up there)."""
        self._pool = torch.cuda.graph_pool_handle()
        self._begin()
        if self.comm is not None:
            self.comm.on_collective = self._collective
        try:
            body()
        finally:
            if self.comm is not None:
                self.comm.on_collective = None
            self._end()

    def replay(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                self.comm.raw_all_reduce(x)

    def release(self):
        """Drop the hipGraphs (with recorded collectives their nodes ARE RCCL kernels) and the all-reduce tensors of the
        chain: what a recording keeps alive that references the communicator."""
        self.items = []
        self._cur = None
        self._pool = None
        self.comm = None


class GraphedTrainStep:
    """One whole training step -- zero_grad, predict_lab_values, weighted loss, backward, optimizer.step --
    captured ONCE into a hipGraph and replayed per epoch (the eICU-scale graph is launch-bound: ~250 kernel
    launches of a few microseconds each).  Shapes are static: the per-epoch supervision mask is a float
    vector updated in place, the dropout seed lives in device memory and is redrawn before every replay.

    Same arithmetic as ``Trainer.train_epoch`` (train.py:332-392 of the reference) with the boolean indexing
    ``pred[mask]`` replaced by a multiplication with the mask.
    """

    def __init__(self, model, data, pi, li, y, lab_weights, optimizer, sup_mask, loss_fn: str = "mae",
                 n_sup_global: Optional[float] = None, warmup: int = 3):
        self.model, self.data, self.opt = model, data, optimizer
        dev = pi.device
        self.pi, self.li, self.y = pi, li, y
        if loss_fn not in ("mae", "mse", "huber"):
            raise ValueError(f"Unknown loss type: {loss_fn}")
        # the reference weights mae / mse by lab (train.py:366-386) and falls back to the unweighted loss otherwise
        self.wl = lab_weights[li].contiguous() if loss_fn in ("mae", "mse") else None
        self._sv = _SupervisionState(sup_mask, None, n_sup_global)   # mask + 1/n_sup on the device: set_mask()
        self.sup = self._sv.sup
        self.loss_fn = loss_fn
        model._seed_dev = _new_seed_state(dev)
        self.loss = torch.zeros((), device=dev)
        model.train()
        if len(model.embeddings) == 0:
            model._init_embeddings(data)
        snap = _snapshot_training_state(model, optimizer)           # warm-up and capture run REAL steps: undone below
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode()):
            self._body()
        torch.cuda.synchronize()
        _restore_training_state(model, optimizer, snap)

    def _body(self):
        from . import ops
        # every parameter, the (frozen, F5) embedding tables included: their .grad is then adopted, not accumulated
        self.model.zero_grad(set_to_none=True)
        pred = self.model.predict_lab_values(self.data, self.pi, self.li)
        loss = ops.weighted_pair_loss(pred, self.y, self.wl, self.sup, 1.0, self.loss_fn, self._sv.inv_den)
        loss.backward()
        self.opt.step()
        ops.seed_advance(self.model._seed_dev)       # fresh dropout masks for the next replay, drawn on the device
        self.loss = loss.detach()                    # (lives in the graph's pool: the same address at every replay)

    def set_mask(self, sup_mask, n_sup_global=None):
        """New supervision subset; the normaliser 1 / n_sup follows it (device scalar read by the captured loss)."""
        self._sv.set_mask(sup_mask, n_sup_global)

    def step(self) -> torch.Tensor:
        if hasattr(self.opt, "sync_hyper"):
            self.opt.sync_hyper()                    # a scheduler may have changed lr since the last replay
        self.graph.replay()
        return self.loss


class PiecewiseGraphedTrainStep:
    """The training step with forward, loss and backward driven by hand (no autograd engine: gradients are assigned to
    ``.grad`` directly, nothing is cloned) and recorded into hipGraphs: ONE graph on a single GPU (``comm=None``), a
    CHAIN of graphs with the RCCL all-reduces between them when the patients are sharded.

    A patient-sharded step has ~14 tiny all-reduces (Sync-BN statistics, vocab-side partial sums, the gradient bucket;
    dist.py) between its ~330 kernel launches.  Collectives are not captured: every ``ShardComm.all_reduce`` met while
    the step is recorded closes the current graph segment, runs eagerly, and opens the next segment, so a replay is
    ``segment, all_reduce, segment, ...`` -- ~15 graph launches and ~14 collectives of host work instead of one Python
    dispatch per kernel, and nothing that a stock RCCL cannot do.  All segments share one memory pool (static
    addresses), are recorded on one thread -- forward, loss and backward are driven by hand (``_Run.run_forward`` /
    ``run_backward``), not through the autograd engine -- and replay in recording order.
    """

    def __init__(self, model, plan, pi, li, y, lab_weights, optimizer, sup_mask, comm, loss_fn: str = "mae",
                 n_sup_global: Optional[float] = None, warmup: int = 2, mask_fraction: Optional[float] = None,
                 loss_out: Optional[torch.Tensor] = None, supervised_heads_only: bool = True,
                 capture_collectives: Optional[bool] = None, reduce_loss: bool = False):
        """capture_collectives: record the all-reduces of a sharded step INTO the hipGraph (one launch per step) instead
        of cutting the recording at each of them; None = ask the communicator (`ShardComm.capturable()`: RCCL backend and
        a probe capture that went through).  Chosen here, once -- never by retrying a failed recording.
        reduce_loss: a sharded step's `loss` is this shard's share of the global mean; True sums it over the group (one
        more fp64 all-reduce of 8 bytes per step) so that every rank reads the reference's number (Trainer's history).
        mask_fraction: draw a NEW supervision subset of that fraction inside every step (device RNG; the reference
        redraws it every epoch, train.py:150-176) -- `sup_mask` may then be None; otherwise the subset is `sup_mask`
        until set_mask.  loss_out: fp64 device scalar the step writes its loss to.
        supervised_heads_only: the two edge heads are evaluated on the supervised pairs alone (the loss of train.py:366-386
        reads predictions[supervision_mask] and nothing else; message passing still covers the whole graph): loss,
        gradients and parameter updates are bit for bit those of the full sweep, `self.pred` is 0 elsewhere."""
        from . import ops
        from .model import _Run
        if loss_fn not in ("mae", "mse", "huber"):
            raise ValueError(f"Unknown loss type: {loss_fn}")
        if sup_mask is None and mask_fraction is None:
            raise ValueError("PiecewiseGraphedTrainStep: a supervision mask or a mask_fraction")
        self.model, self.plan, self.opt, self.comm = model, plan, optimizer, comm
        self.pi, self.li, self.y = pi, li, y
        # the reference weights mae / mse by lab (train.py:366-386) and falls back to the unweighted loss otherwise
        self.wl = lab_weights[li].contiguous() if loss_fn in ("mae", "mse") else None
        self.mask_fraction = None if mask_fraction is None else float(mask_fraction)
        self.supervised_heads_only = bool(supervised_heads_only)
        self.reduce_loss = bool(reduce_loss) and comm is not None
        self.pred = None
        self._sel_ready = None
        dev = pi.device
        self.n_pairs_global, self._draw_ids = None, None
        if comm is not None and self.mask_fraction is not None:
            # the draw is keyed on GLOBAL pair ids 0 .. n_global-1: the positions in the unsharded pair list when the
            # caller gave them (dist.shard_pairs), the shards' pairs numbered back to back otherwise (one all-reduce of
            # the shard sizes, here, once)
            ids = getattr(comm, "pair_ids", None)
            if ids is not None:
                hi = torch.tensor([float(int(ids.max()) + 1 if ids.numel() else 0)], dtype=torch.float64, device=dev)
                sizes = torch.zeros(int(comm.world), dtype=torch.float64, device=dev)
                sizes[int(comm.rank)] = hi[0]
                comm.raw_all_reduce(sizes)
                self.n_pairs_global = int(sizes.max().item())
                self._draw_ids = ids.to(torch.int64).contiguous()
            else:
                sizes = torch.zeros(int(comm.world), dtype=torch.float64, device=dev)
                sizes[int(comm.rank)] = float(pi.numel())
                comm.raw_all_reduce(sizes)
                off = int(sizes[:int(comm.rank)].sum().item())
                self.n_pairs_global = int(sizes.sum().item())
                self._draw_ids = torch.arange(off, off + pi.numel(), dtype=torch.int64, device=dev)
        self._sv = _SupervisionState(sup_mask, comm, n_sup_global, n=pi.numel(), device=dev)
        self.sup = self._sv.sup
        self.loss_fn = loss_fn
        self._loss_out = loss_out
        self._ops, self._Run = ops, _Run
        self._sel = None
        model._seed_dev = _new_seed_state(dev)
        self._select()                 # the pair lists of the backward (rebuilt inside the step when it draws its mask)
        self.loss = torch.zeros((), device=dev)
        self.params = [p for p in model.parameters()]
        model.train()
        self._rec = _Recording(comm, capture_collectives)
        snap = _snapshot_training_state(model, optimizer)           # warm-up and capture run REAL steps: undone below
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._body()
            torch.cuda.synchronize()
            self._rec.record(self._body)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _restore_training_state(model, optimizer, snap)

    # ---- what was recorded (tests and bench read these)
    @property
    def items(self):
        return self._rec.items

    @property
    def n_collectives(self) -> int:
        return self._rec.n_collectives

    @property
    def capture_collectives(self) -> bool:
        return self._rec.capture_collectives

    def release(self):
        """Drop what this step keeps alive that references the communicator or its streams -- the recording (hipGraphs,
        all-reduce tensors of a chain), the side-stream event -- so that `ShardComm.close` tears the group down with
        nothing of the step left.  The step cannot run afterwards."""
        self._rec.release()
        self._sel_ready = None
        self.comm = None
        self._sv.comm = None

    def _body(self):
        ops, model = self._ops, self.model
        with torch.no_grad():
            for p in self.params:
                p.grad = None
            if self.mask_fraction is not None:       # this step's supervision subset, its size and its pair lists
                self._draw_and_select()
            run = self._Run(model, self.plan)
            run.pairs = model._pairs(self.pi, self.li, self.plan.n_rows,
                                     getattr(self.comm, "pair_ids", None) if self.comm else None,
                                     self.plan.lab_deg, int(model.degree_threshold))
            run.n_pairs = self.pi.numel()
            run.need_grad = True
            run.static_select = self._sel
            run.lists_ready = self._sel_ready
            if self.supervised_heads_only:
                run.forward_select = self._sel
            (pred,) = run.run_forward("predict")
            self.pred = pred
            loss, dpred = ops.pair_loss(pred, self.y, self.wl, self.sup, 1.0, self.loss_fn, self._sv.inv_den,
                                        loss_out=self._loss_out)
            if self.reduce_loss:
                self.comm.all_reduce(loss)           # reporting only: the gradient came out of the same pass already
            grads = run.run_backward((dpred,))
            for p, g in zip(self.params, grads):
                if g is not None:
                    p.grad = g
            self.opt.step()
            ops.seed_advance(model._seed_dev)        # fresh dropout masks for the next replay, drawn on the device
            self.loss = loss                         # fp64 scalar in the graph's pool: the same address at every replay

    def _draw_and_select(self):
        """The step's supervision subset, its size and the pair lists derived from it.  Single GPU: on the side stream,
        beside the encoder pass (nothing reads them before the heads); sharded: on the main stream (the subset size is a
        collective, and collectives cut the graph segments there)."""
        model = self.model
        ids = self._draw_ids
        side = getattr(model, "_side_stream", None)
        if self.comm is not None or side is None:
            self._sv.draw(self.mask_fraction, model._seed_dev, ids, self.n_pairs_global)
            self._select()
            return
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._sv.draw(self.mask_fraction, model._seed_dev, ids)
            self._select()
            self._sel_ready = torch.cuda.Event()
            self._sel_ready.record(side)      # the heads wait for this (_Run.heads_fwd); the encoder pass does not

    # ---- replay
    def _select(self):
        """Pair lists of the backward (per head: the positions of the supervised pairs in patient order) from the
        supervision mask: the loss gradient is exactly 0 outside it (train.py:366-370), so the lists only change with
        set_mask -- the captured step holds their addresses, they are refreshed in place."""
        ops, model = self._ops, self.model
        pairs = model._pairs(self.pi, self.li, self.plan.n_rows,
                             getattr(self.comm, "pair_ids", None) if self.comm else None,
                             self.plan.lab_deg, int(model.degree_threshold))
        self._sel = ops.pair_select(pairs[0], self.plan.lab_deg, int(model.degree_threshold), self.sup, io_perm=pairs[2],
                                    out=self._sel)

    def set_mask(self, sup_mask, n_sup_global=None):
        """New supervision subset; 1 / n_sup (summed over the shards) and the backward's pair lists follow it on the device."""
        if self.mask_fraction is not None:
            raise ValueError("this step draws its supervision subset itself (mask_fraction)")
        self._sv.set_mask(sup_mask, n_sup_global)
        self._select()

    def step(self) -> torch.Tensor:
        if hasattr(self.opt, "sync_hyper"):
            self.opt.sync_hyper()                    # a scheduler may have changed lr since the last replay
        self._rec.replay()
        return self.loss


class GraphedEval:
    """``Trainer.validate`` (train.py:394-431 of the reference) as ONE captured hipGraph: the eval-mode
    ``predict_lab_values`` on the split's pairs and the unweighted ``compute_regression_loss`` over all of them.  The
    loss stays on the device (``loss_out``, or a scalar of the graph's pool); the predictions are ``self.pred``."""

    def __init__(self, model, plan, pi, li, y, loss_fn: str = "mae", loss_out: Optional[torch.Tensor] = None,
                 warmup: int = 1, comm=None):
        """comm: the model is patient-sharded -- pi / li / y are this shard's pairs, the loss is the mean over ALL shards'
        pairs (local sum over the global count, then one fp64 all-reduce of the scalar: recorded inside the graph when the
        communicator allows, issued right after the replay otherwise)."""
        from . import ops
        from .model import _Run
        if loss_fn not in ops.LOSS_TYPES:
            raise ValueError(f"Unknown loss type: {loss_fn}")
        self.model, self.plan = model, plan
        self.pi, self.li, self.y = pi, li, y
        self.loss_fn, self._loss_out = loss_fn, loss_out
        self._ops, self._Run = ops, _Run
        self.loss = self.pred = None
        self.comm = comm
        self._n_global = max(int(pi.numel()), 1)
        if comm is not None:
            n = torch.tensor([float(pi.numel())], dtype=torch.float64, device=pi.device)
            comm.raw_all_reduce(n)
            self._n_global = max(int(n.item()), 1)
            if self._loss_out is None:         # the all-reduce acts in place on a tensor the caller can read
                self._loss_out = torch.zeros((), dtype=torch.float64, device=pi.device)
        was_training = model.training
        model.eval()
        # a sharded eval forward meets collectives too (the vocab-side partial sums of every layer, the loss): recorded
        # like the training step -- inside the one graph, or as a chain of segments (_Recording)
        self._rec = _Recording(comm)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._body()
            torch.cuda.synchronize()
            self._rec.record(self._body)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        model.train(was_training)

    def _body(self):
        ops, model = self._ops, self.model
        with torch.no_grad():
            run = self._Run(model, self.plan)
            run.pairs = model._pairs(self.pi, self.li, self.plan.n_rows, None, self.plan.lab_deg,
                                     int(model.degree_threshold))
            run.n_pairs = self.pi.numel()
            run.need_grad = False
            (pred,) = run.run_forward("predict")
            self.loss, _ = ops.pair_loss(pred, self.y, None, None, 1.0 / self._n_global, self.loss_fn,
                                         loss_out=self._loss_out, want_dpred=False)
            if self.comm is not None:
                self.comm.all_reduce(self.loss)          # (through on_collective: recorded, or a cut of the chain)
            self.pred = pred

    @property
    def graph(self):
        """The one hipGraph of an unsharded (or fully recorded) pass."""
        return self._rec.items[0][1] if len(self._rec.items) == 1 else None

    def release(self):
        self._rec.release()
        self.comm = None

    def step(self) -> torch.Tensor:
        self._rec.replay()
        return self.loss
