"""Optimizer of the training step on the device: ``Adam`` = ``torch.optim.Adam`` (the reference's optimizer,
src/train.py:216-229) with ``step()`` replaced by ONE launch of ``mmg_adam_step`` (include/mmgnn.h).

The parameters of a group are moved into one flat fp32 bucket (each ``p.data`` becomes a view of it), the moments live
in two more; gradients stay where the backward kernels wrote them and are found through a pointer table.  The
hyper-parameters (lr, betas, eps, weight_decay) are read by the kernel from a small device buffer that follows
``param_groups`` (``sync_hyper``), so a learning-rate scheduler keeps working when the step is a replayed hipGraph.  Same
arithmetic, same ``state_dict()`` layout as ``torch.optim.Adam`` ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}},
'param_groups': [...]}), so the reference's checkpoint dict (train.py:501-509) round-trips.  The step counter is a device
scalar advanced by the kernel, so a captured hipGraph keeps counting across replays.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib
from .ops import _stream


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self._flat = None
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._flat = [self._flatten_group(group) for group in self.param_groups]

    def add_param_group(self, param_group):
        """torch.optim.Optimizer.add_param_group; a group added after construction gets its own flat buckets."""
        super().add_param_group(param_group)
        if self._flat is not None:                 # (Optimizer.__init__ adds the initial groups before _flat exists)
            self._flat.append(self._flatten_group(self.param_groups[-1]))

    def _flatten_group(self, group):
        ps: List[torch.nn.Parameter] = [p for p in group["params"]]
        if not ps:
            return None
        dev = ps[0].device
        if dev.type != "cuda" or any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise _lib.MmgError("mmgnn.optim.Adam: fp32 parameters on one HIP device (move the model first)")
        sizes = [p.numel() for p in ps]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        if offs[-1] >= 2 ** 31:
            raise ValueError("parameter bucket exceeds int32 offsets")
        flat_p = torch.empty(offs[-1], device=dev)
        flat_m = torch.zeros(offs[-1], device=dev)
        flat_v = torch.zeros(offs[-1], device=dev)
        step = torch.zeros((), device=dev)
        with torch.no_grad():
            for p, o, n in zip(ps, offs, sizes):
                flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = flat_p[o:o + n].view(p.shape)            # the Parameter object (and its identity) stays
                self.state[p] = {"step": step, "exp_avg": flat_m[o:o + n].view(p.shape),
                                 "exp_avg_sq": flat_v[o:o + n].view(p.shape)}
        fl = dict(p=flat_p, m=flat_m, v=flat_v, step=step, ps=ps,
                  ticket=torch.zeros(1, dtype=torch.int32, device=dev),
                  offs=(C.c_int32 * len(offs))(*offs),
                  # (lr, beta1, beta2, eps, weight_decay) on the device: what the kernel reads -- a captured step keeps its
                  # launch arguments, a scheduler (train.py:271-291) changes param_groups between replays
                  hyper=torch.zeros(5, device=dev), hyper_host=None)
        self._upload_hyper(group, fl)
        return fl

    @staticmethod
    def _hyper_of(group):
        b1, b2 = group["betas"]
        return (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))

    def _upload_hyper(self, group, fl):
        h = self._hyper_of(group)
        if h != fl["hyper_host"]:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.MmgError("mmgnn.optim.Adam: hyper-parameters changed inside a stream capture; call sync_hyper() "
                                    "between replays")
            fl["hyper"].copy_(torch.tensor(h, dtype=torch.float32))
            fl["hyper_host"] = h

    def sync_hyper(self):
        """Copy param_groups' (lr, betas, eps, weight_decay) to the device if they changed.  Eager steps do it themselves;
        a captured step calls this before every replay (one tuple comparison when nothing changed)."""
        for group, fl in zip(self.param_groups, self._flat):
            if fl is not None:
                self._upload_hyper(group, fl)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        if len(self._flat) != len(self.param_groups):
            raise _lib.MmgError("mmgnn.optim.Adam: param_groups were edited behind add_param_group")
        for group, fl in zip(self.param_groups, self._flat):
            if fl is None:
                continue
            ps = fl["ps"]
            gp = (C.c_void_p * len(ps))()
            for i, p in enumerate(ps):
                g = p.grad
                if g is None:
                    gp[i] = None
                    continue
                if not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != p.numel():
                    raise _lib.MmgError("mmgnn.optim.Adam: gradients must be contiguous fp32 device tensors")
                gp[i] = g.data_ptr()
            self._upload_hyper(group, fl)
            _lib.check(lib.mmg_adam_step_dev(C.c_void_p(fl["p"].data_ptr()), C.c_void_p(fl["m"].data_ptr()),
                                             C.c_void_p(fl["v"].data_ptr()), gp, fl["offs"], len(ps),
                                             C.c_void_p(fl["hyper"].data_ptr()), C.c_void_p(fl["step"].data_ptr()),
                                             C.c_void_p(fl["ticket"].data_ptr()), _stream()), "mmg_adam_step_dev")
        return loss

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors: copy what it loaded back into the flat buckets and restore the views."""
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for fl in self._flat:
                if fl is None:
                    continue
                off = 0
                for p in fl["ps"]:
                    n = p.numel()
                    st = self.state.get(p, {})
                    if "exp_avg" in st:
                        fl["m"][off:off + n].copy_(st["exp_avg"].reshape(-1).to(fl["m"].device))
                        fl["v"][off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(fl["v"].device))
                        fl["step"].fill_(float(st["step"]))
                    self.state[p] = {"step": fl["step"], "exp_avg": fl["m"][off:off + n].view(p.shape),
                                     "exp_avg_sq": fl["v"][off:off + n].view(p.shape)}
                    off += n
