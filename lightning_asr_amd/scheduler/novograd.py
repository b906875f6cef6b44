"""``Novograd`` with the reference's constructor (scheduler/novograd.py:49-73) as a torch Optimizer
whose step() is ONE multi-tensor HIP call over the model's flat parameter / gradient buffers
(csrc/optim.hip), instead of ~10 tiny kernels per tensor."""
from __future__ import annotations

import torch
from torch.optim.optimizer import Optimizer

from .. import ops


class Novograd(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.95, 0.98), eps=1e-8, weight_decay=0, grad_averaging=False,
                 amsgrad=False, luc=False, luc_trust=1e-3, luc_eps=1e-8):
        if lr < 0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"Betas have to be between 0 and 1: {betas}")
        if grad_averaging or amsgrad or luc:
            raise NotImplementedError("grad_averaging / amsgrad / luc are not used by the reference path (train.py:46) "
                                      "and are not implemented in the HIP kernel")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_averaging=False, amsgrad=False)
        super().__init__(params, defaults)
        owners = {id(getattr(p, "_lasr_owner", None)): getattr(p, "_lasr_owner", None)
                  for g in self.param_groups for p in g["params"]}
        if len(owners) != 1 or None in owners.values():
            raise ValueError("Novograd (HIP) optimises the parameters of exactly one MyModel2 (its flat GPU buffer)")
        self.owner = next(iter(owners.values()))
        n = self.owner.native
        if sum(len(g["params"]) for g in self.param_groups) != len(n.param_infos()):
            raise ValueError("Novograd (HIP) needs all parameters of the model in its param groups")
        self.exp_avg = torch.zeros_like(n.params)
        self.exp_avg_sq = torch.zeros(len(n.param_infos()), dtype=torch.float32, device=n.device)
        self.offsets = n.param_offsets()
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=n.device)
        self.grad_scale = 1.0          # set to 1/world by the data-parallel wrapper after a SUM all-reduce

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        n = self.owner.native
        self.lr_dev.fill_(float(g["lr"]))
        ops.novograd_step(n.params, n.grads, self.exp_avg, self.exp_avg_sq, self.offsets, self.lr_dev, g["betas"][0],
                          g["betas"][1], g["eps"], g["weight_decay"], self.grad_scale)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        # gradients live in the flat buffer and are fully overwritten by every backward
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    def state_dict(self):
        sd = super().state_dict()
        sd["lasr"] = {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone()}
        return sd

    def load_state_dict(self, sd):
        extra = sd.pop("lasr", None)
        super().load_state_dict(sd)
        if extra is not None:
            self.exp_avg.copy_(extra["exp_avg"])
            self.exp_avg_sq.copy_(extra["exp_avg_sq"])
