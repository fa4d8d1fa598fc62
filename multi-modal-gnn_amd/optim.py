"""Optimizer of the training step on the device: ``Adam`` = ``torch.optim.Adam`` (the reference's optimizer,
src/train.py:216-229) with ``step()`` replaced by ONE launch of ``mmg_adam_step`` (include/mmgnn.h).

The parameters of a group are moved into one flat fp32 bucket (each ``p.data`` becomes a view of it), the moments live
in two more; gradients stay where the backward kernels wrote them and are found through a pointer table.  Same
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
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._flat = []
        for group in self.param_groups:
            ps: List[torch.nn.Parameter] = [p for p in group["params"]]
            if not ps:
                self._flat.append(None)
                continue
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
            self._flat.append(dict(p=flat_p, m=flat_m, v=flat_v, step=step, ps=ps,
                                   ticket=torch.zeros(1, dtype=torch.int32, device=dev),
                                   offs=(C.c_int32 * len(offs))(*offs)))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
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
            b1, b2 = group["betas"]
            lr = group["lr"]
            _lib.check(lib.mmg_adam_step(C.c_void_p(fl["p"].data_ptr()), C.c_void_p(fl["m"].data_ptr()),
                                         C.c_void_p(fl["v"].data_ptr()), gp, fl["offs"], len(ps), float(lr), float(b1),
                                         float(b2), float(group["eps"]), float(group["weight_decay"]),
                                         C.c_void_p(fl["step"].data_ptr()), C.c_void_p(fl["ticket"].data_ptr()), _stream()),
                       "mmg_adam_step")
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
