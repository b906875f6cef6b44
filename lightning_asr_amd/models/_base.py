"""MyModel2: the reference's nn.Module surface (models/QuartNet.py:264-291) over the native plan.

``MyModel2(labels, drop_rate=0., mask=False[, in_c=64])``; ``forward(input (B,1,F,T) f32, percents
(B,)) -> log-probs (B, T', len(labels)+1)``.  ``state_dict()`` has exactly the reference's keys
(184 entries for asr13x1) so PL-style checkpoints round-trip.  Parameters are views of ONE flat f32
buffer on the GPU (and ``.grad`` views of one flat gradient buffer), which is what the C ABI, the
RCCL all-reduce and the multi-tensor NovoGrad work on.  No math happens in this file."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .. import ops
from ..engine import NativeModel


class _PlanFn(torch.autograd.Function):
    """Whole-model autograd node: forward and backward are single C-ABI calls."""

    @staticmethod
    def forward(ctx, module, feats_btc, pct, *params):
        logp, am = module.native.forward(feats_btc, pct, training=module.training)
        module.last_argmax = am
        ctx.module = module
        return logp

    @staticmethod
    def backward(ctx, grad_logp):
        m = ctx.module
        m.native.backward(grad_logp)
        # hand every parameter its view of the flat gradient buffer (no copies, optimiser stays flat)
        for p, t in zip(m._plist, m.native.param_infos()):
            p.grad = m.native.view(t, m.native.grads)
        return (None, None, None) + (None,) * len(m._plist)


class MyModel2Base(nn.Module):
    variant = "plain"

    def __init__(self, labels: Sequence[str], drop_rate: float = 0.0, mask: bool = False, in_c: int = 64,
                 act: str = "relu", dtype=torch.float32, device="cuda", seed: Optional[int] = None):
        super().__init__()
        self.labels = labels
        self.native = NativeModel(self.variant, len(labels) + 1, mask=bool(mask), act=act, dtype=dtype, in_c=in_c, device=device)
        self.native.init_parameters(int(torch.initial_seed() & 0x7fffffff) if seed is None else seed)
        if drop_rate:   # nn.Dropout(p=drop_rate) in every SeprationConv and in last_cnn2 (models/QuartNet.py:26,38,149)
            self.native.set_dropout(float(drop_rate), int(torch.initial_seed()) if seed is None else seed)
        self.last_argmax = None
        self._plist: List[nn.Parameter] = []
        self._counters = {}
        for t in self.native.tensors:
            *path, leaf = t.name.split(".")
            mod = self
            for name in path:
                if name not in mod._modules:
                    mod.add_module(name, nn.Module())
                mod = mod._modules[name]
            if t.kind == 0:
                p = nn.Parameter(self.native.view(t), requires_grad=True)
                p._lasr_owner = self
                mod.register_parameter(leaf, p)
                self._plist.append(p)
            elif t.kind == 1:
                mod.register_buffer(leaf, self.native.view(t))
            else:
                mod.register_buffer(leaf, torch.zeros((), dtype=torch.int64))    # host scalar: bumping it launches nothing
                self._counters[t.name] = (mod, leaf)
        # runs whenever this module is loaded, also as a child of LightingModule.load_state_dict / load_from_checkpoint / resume
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._sync_counters())

    # nn.Module.to()/cuda() would re-allocate the parameters and break the flat views
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, device=self.native.device))
        if probe.device != self.native.device or probe.dtype != torch.float32:
            raise RuntimeError("MyModel2 lives in one flat f32 GPU buffer; construct it with device=/dtype= instead of .to()")
        return self

    def forward(self, input: torch.Tensor, percents: torch.Tensor) -> torch.Tensor:
        """input (B,1,F,T) f32 in the reference layout -> (B, T', C) f32 log-probs."""
        dev = self.native.device
        x = input.to(dev, torch.float32)
        if x.dim() == 4:
            x = x.squeeze(1)                                   # models/QuartNet.py:154
        feats = ops.bct_to_btc(x.contiguous(), self.native.act_dtype)
        return self.forward_features(feats, percents)

    def forward_features(self, feats_btc: torch.Tensor, percents: torch.Tensor) -> torch.Tensor:
        """Same, from channels-last [B][T][F] features already on the GPU (the on-device mel output)."""
        pct = percents.to(self.native.device, torch.float32).contiguous()
        out = _PlanFn.apply(self, feats_btc, pct, *self._plist)
        if self.training:
            for mod, leaf in self._counters.values():
                getattr(mod, leaf).add_(1)
        return out

    def _sync_counters(self) -> None:
        """num_batches_tracked: the registered buffers are the source of truth; NativeModel.counters mirrors them"""
        for name, (mod, leaf) in self._counters.items():
            self.native.counters[name] = getattr(mod, leaf).detach().to("cpu", torch.int64).clone()
