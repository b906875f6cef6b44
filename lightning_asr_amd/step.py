"""One data-parallel training step of the hot path, entirely on the GPU:

    wave (HBM) -> log-mel features -> QuartzNet forward -> mean CTC -> backward
               -> gradient all-reduce (RCCL over xGMI, world > 1) -> NovoGrad -> LR schedule

This is what ``LightingModule.training_step`` + Lightning's optimiser loop do in the reference
(train.py:64-86, scheduler/novograd.py:75-145); here each stage is one or a few C-ABI calls."""
from __future__ import annotations

from typing import Optional

import torch

import os

from . import ops
from .comm import Communicator
from .engine import NativeModel
from .schedule import CosineAnnealingWarmupRestarts


class TrainStep:
    def __init__(self, model: NativeModel, learning_rate: float = 1e-2, weight_decay: float = 1e-3,
                 betas=(0.8, 0.5), eps: float = 1e-8, schedule: Optional[CosineAnnealingWarmupRestarts] = None,
                 process_group=None, comm: Optional[Communicator] = None):
        """comm: the library-owned RCCL communicator (lasr_comm_*).  Default for world > 1 on the nccl backend: one is built
        from the torch.distributed group (which then only bootstraps the unique id); ``LASR_COMM=torch`` keeps the gradient
        exchange on torch.distributed (the only choice for gloo groups: CPU rehearsal, two ranks sharing a GPU in the tests)."""
        self.model = model
        self.wd, self.betas, self.eps = weight_decay, betas, eps
        dev = model.device
        self.exp_avg = torch.zeros_like(model.params)
        self.exp_avg_sq = torch.zeros(len(model.param_infos()), dtype=torch.float32, device=dev)
        self.offsets = model.param_offsets()
        self.schedule = schedule
        self.lr = schedule.lr if schedule is not None else learning_rate
        self.lr_dev = torch.tensor([self.lr], dtype=torch.float32, device=dev)
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.global_step = 0
        self.overlap = True        # overlap the gradient all-reduce with backward (world > 1)
        self._reduced = False
        self.comm = comm
        if comm is not None:
            self.world = comm.world
        elif self.world > 1 and os.environ.get("LASR_COMM", "rccl") == "rccl" and \
                torch.distributed.get_backend(process_group) == "nccl":
            self.comm = Communicator.from_torch_distributed(dev, process_group)
        # exercise the staged + async all-reduce path on a 1-rank group too (validation on one GPU)
        self.force_staged = bool(int(os.environ.get("LASR_FORCE_OVERLAP", "0"))) and \
            (self.comm is not None or torch.distributed.is_initialized())
        self._prefetched = None    # (key, feats, pct) of the batch announced by the previous step(prefetch_wave=...)

    def broadcast_parameters(self, src: int = 0) -> None:
        """DDP wrap-time broadcast of parameters and buffers from rank 0."""
        if self.comm is not None:
            self.comm.broadcast(self.model.params, src)
            self.comm.broadcast(self.model.buffers, src)
            self.comm.wait()
        elif self.world > 1:
            torch.distributed.broadcast(self.model.params, src, group=self.pg)
            torch.distributed.broadcast(self.model.buffers, src, group=self.pg)

    def features(self, wave, sample_lens=None, dither=None, aug=None):
        """wave (B, L) f32 on the GPU -> ([B][T][64] features in the activation dtype, pct (B))."""
        _, btf, _, pct = ops.mel(wave, sample_lens, dither, aug, True, self.model.act_dtype, want_bft=False, want_btf=True)
        return btf, pct

    def optimizer_step(self) -> None:
        m = self.model
        if self.world > 1 and not getattr(self, "_reduced", False):
            # one flat 20 MB SUM all-reduce; the 1/world average is folded into the optimiser's grad scale
            if self.comm is not None:
                self.comm.all_reduce(m.grads)
                self.comm.wait()
            else:
                torch.distributed.all_reduce(m.grads, group=self.pg)
        self._reduced = False
        ops.novograd_step(m.params, m.grads, self.exp_avg, self.exp_avg_sq, self.offsets, self.lr_dev, self.betas[0],
                          self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world)
        if self.schedule is not None:
            self.lr = self.schedule.step()
            self.lr_dev.fill_(self.lr)
        self.global_step += 1

    def step_features(self, feats, pct, targets, tgt_lens, want_logp: bool = True):
        m = self.model
        if self.overlap and (self.world > 1 or self.force_staged):
            # bucketed SUM all-reduce, launched bucket by bucket in reverse layer order while the units below
            # are still being differentiated: RCCL runs on its own stream and waits (event) only for the
            # kernels enqueued so far; the optimiser waits for all buckets.
            if self.comm is not None:       # library-owned communicator and side stream (lasr_comm_*)
                loss, nll, logp, am = m.loss_backward_staged(feats, pct, targets, tgt_lens,
                                                             lambda ranges: self.comm.all_reduce_ranges(m.grads, ranges), want_logp=want_logp)
                self.comm.wait()
            else:
                works = []
                loss, nll, logp, am = m.loss_backward_staged(
                    feats, pct, targets, tgt_lens,
                    lambda ranges: works.extend(torch.distributed.all_reduce(m.grads[lo:hi], group=self.pg, async_op=True)
                                                for lo, hi in ranges), want_logp=want_logp)
                for w in works:
                    w.wait()
            self._reduced = True
        else:
            loss, nll, logp, am = m.loss_backward(feats, pct, targets, tgt_lens, want_logp=want_logp)
            self._reduced = False
        self.optimizer_step()
        return loss, nll, logp, am

    @staticmethod
    def _prefetch_key(wave, sample_lens, dither, aug):
        return tuple((t.data_ptr(), tuple(t.shape), t._version) if t is not None else None for t in (wave, sample_lens, dither, aug))

    def step(self, wave, targets, tgt_lens, sample_lens=None, dither=None, aug=None, prefetch_wave=None, prefetch_lens=None,
             prefetch_dither=None, prefetch_aug=None, want_logp: bool = True):
        """One training step on `wave`.  prefetch_wave: the NEXT step's waveforms (already in HBM): their log-mel features are
        computed during this step in the grid of the CTC lattice kernel (32 busy workgroups, 224 idle CUs for ~0.1 ms) and
        picked up by the next call if it passes the same tensors - the data-loader prefetch of the reference's workers;
        every step still computes exactly one batch of features."""
        pf, self._prefetched = self._prefetched, None
        if pf is not None and pf[0] == self._prefetch_key(wave, sample_lens, dither, aug):
            feats, pct = pf[1], pf[2]
        else:
            feats, pct = self.features(wave, sample_lens, dither, aug)
        nxt = None
        if prefetch_wave is not None:
            nf, npct = self.model.arm_prefetch(prefetch_wave, prefetch_lens, prefetch_dither, prefetch_aug)
            nxt = (self._prefetch_key(prefetch_wave, prefetch_lens, prefetch_dither, prefetch_aug), nf, npct)
        out = self.step_features(feats, pct, targets, tgt_lens, want_logp=want_logp)
        self._prefetched = nxt
        return out
