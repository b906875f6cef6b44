"""One data-parallel training step of the hot path, entirely on the GPU:

    wave (HBM) -> log-mel features -> QuartzNet forward -> mean CTC -> backward
               -> gradient all-reduce (RCCL over xGMI, world > 1) -> NovoGrad -> LR schedule

This is what ``LightingModule.training_step`` + Lightning's optimiser loop do in the reference
(train.py:64-86, scheduler/novograd.py:75-145); here each stage is one or a few C-ABI calls."""
from __future__ import annotations

from typing import Optional

import torch

import ctypes as C
import os

from . import _lib, ops
from .comm import Communicator
from .engine import NativeModel
from .schedule import CosineAnnealingWarmupRestarts


_ROCTX = None


def _roctx_on() -> bool:
    global _ROCTX
    if _ROCTX is None:
        _ROCTX = os.environ.get("LASR_ROCTX", "0") not in ("", "0") and bool(_lib.load().lasr_roctx_enabled())
    return _ROCTX


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
                (torch.distributed.get_backend(process_group) == "nccl" or os.environ.get("LASR_RCCL_PATH")):   # (gloo + LASR_RCCL_PATH: rehearsal over the test stub)
            self.comm = self._checked_communicator(dev, process_group)
        # exercise the staged + async all-reduce path on a 1-rank group too (validation on one GPU)
        self.force_staged = bool(int(os.environ.get("LASR_FORCE_OVERLAP", "0"))) and \
            (self.comm is not None or torch.distributed.is_initialized())
        self._prefetched = None    # (key, feats, pct) of the batch announced by the previous step(prefetch_wave=...)
        self._lr_state = None      # device image of the schedule (use_device_schedule)
        self._lr_epoch = None      # host schedule position the device image corresponds to

    @staticmethod
    def _checked_communicator(dev, process_group) -> Optional[Communicator]:
        """The library's RCCL communicator, proven on THIS machine before it carries gradients: every rank all-reduces
        (rank + 1, 1) through it on the side stream and must read (world (world + 1) / 2, world); the ranks then agree (one
        torch.distributed MIN) on whether all of them passed.  Any failure - librccl missing, init error, a wrong sum - sends the
        whole group back to torch.distributed's all-reduce (what LASR_COMM=torch selects), loudly."""
        import sys
        import torch.distributed as dist
        comm, ok, why = None, 1, ""
        try:
            comm = Communicator.from_torch_distributed(dev, process_group)
            probe = torch.tensor([float(comm.rank + 1), 1.0], dtype=torch.float32, device=dev)
            comm.all_reduce(probe)
            comm.wait()
            w = comm.world
            got = probe.cpu().tolist()
            if got != [w * (w + 1) / 2.0, float(w)]:
                ok, why = 0, "all-reduce self-check read %s" % (got,)
        except Exception as e:  # noqa: BLE001
            ok, why = 0, "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        if int(flag.item()) == 1:
            return comm
        sys.stderr.write("lasr_comm self-check failed on rank %d (%s): gradient exchange falls back to torch.distributed\n"
                         % (dist.get_rank(process_group), why or "another rank failed"))
        if comm is not None:
            comm.close()
        return None

    @classmethod
    def from_optimizer(cls, model: NativeModel, optimizer, schedule=None, process_group=None, comm: Optional[Communicator] = None):
        """The fused step over the state of a ``scheduler.novograd.Novograd`` (+ its ``CosineAnnealingWarmupRestarts``): the
        moments and the device learning rate ARE the optimizer's tensors, so ``optimizer.state_dict()`` / checkpoints / resume
        keep working while ``Trainer.fit`` drives this step instead of ``loss.backward(); optimizer.step()``."""
        g = optimizer.param_groups[0]
        ts = cls(model, learning_rate=float(g["lr"]), weight_decay=g["weight_decay"], betas=tuple(g["betas"]), eps=g["eps"],
                 schedule=schedule, process_group=process_group, comm=comm)
        ts.exp_avg, ts.exp_avg_sq, ts.lr_dev = optimizer.exp_avg, optimizer.exp_avg_sq, optimizer.lr_dev
        ts.lr = float(g["lr"])
        ts.lr_dev.fill_(ts.lr)
        ts._optimizer = optimizer
        return ts

    def sync_device_schedule(self) -> None:
        """Re-upload the schedule after its host state changed behind the device twin's back (``load_state_dict`` on resume,
        an assignment to ``ts.lr``): the rate the optimiser reads and the device state restart from the host object."""
        if self.schedule is not None:
            self.lr = self.schedule.lr
        self.lr_dev.fill_(float(self.lr))
        if self._lr_state is not None:
            self._lr_state = None
            self.use_device_schedule()

    def broadcast_parameters(self, src: int = 0) -> None:
        """DDP wrap-time broadcast of parameters and buffers from rank 0."""
        if self.comm is not None:
            self.comm.broadcast(self.model.params, src)
            self.comm.broadcast(self.model.buffers, src)
            self.comm.wait()
        elif self.world > 1:
            torch.distributed.broadcast(self.model.params, src, group=self.pg)
            torch.distributed.broadcast(self.model.buffers, src, group=self.pg)

    def features(self, wave, sample_lens=None, dither=None, aug=None, logical_len=None):
        """wave (B, L) f32 on the GPU -> ([B][T][64] features in the activation dtype, pct (B)).  logical_len: the longest utterance
        when the rows are wider than that (ops.mel)."""
        _, btf, _, pct = ops.mel(wave, sample_lens, dither, aug, True, self.model.act_dtype, want_bft=False, want_btf=True,
                                 logical_len=logical_len)
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
        if getattr(self, "_opt_ws", None) is None:       # kept across steps: zeroed once, left zeroed by every step (no memset launch)
            self._opt_ws = ops.novograd_workspace(self.exp_avg_sq.numel(), m.params.numel(), m.device)
        ops.novograd_step(m.params, m.grads, self.exp_avg, self.exp_avg_sq, self.offsets, self.lr_dev, self.betas[0],
                          self.betas[1], self.eps, self.wd, grad_scale=1.0 / self.world, ws=self._opt_ws)
        if self.schedule is not None:
            if self._lr_state is not None and self.schedule.last_epoch != self._lr_epoch:
                self.sync_device_schedule()                # the host schedule was loaded / reset since the last step
            self.lr = self.schedule.step()                 # host copy of the schedule: bookkeeping / logging / checkpoints
            self._lr_epoch = self.schedule.last_epoch
            if self._lr_state is not None:                 # the rate the next step uses is computed on the device
                _lib.call("lasr_lr_schedule_step", self._lr_state.data_ptr(), self.lr_dev.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
            else:
                self.lr_dev.fill_(self.lr)
        self.global_step += 1

    def use_device_schedule(self) -> None:
        """Move the LR schedule's state onto the GPU (csrc/sched.hip): from now on every optimizer_step advances it with a
        one-thread kernel instead of a host scalar + fill, so the whole step is capturable.  The host object keeps stepping
        alongside (same values to f32 rounding) for logging and checkpoints."""
        if self.schedule is None or self._lr_state is not None:
            return
        sc = self.schedule
        nb = int(_lib.load().lasr_lr_schedule_state_bytes())
        host = C.create_string_buffer(nb)
        _lib.call("lasr_lr_schedule_init", host, nb, int(sc.first_cycle_steps), float(sc.cycle_mult), float(sc.base_max_lr),
                  float(sc.min_lr), int(sc.warmup_steps), float(sc.gamma), int(sc.cycle), int(sc.step_in_cycle),
                  int(sc.cur_cycle_steps), int(sc.last_epoch))
        self._lr_state = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.model.device)
        self._lr_epoch = int(sc.last_epoch)

    def step_features(self, feats, pct, targets, tgt_lens, want_logp: bool = True):
        if _roctx_on():                      # LASR_ROCTX=1: "lasr:step" around the whole step (rocprofv3 --marker-trace)
            _lib.load().lasr_roctx_range_push(b"lasr:step")
            try:
                return self._step_features(feats, pct, targets, tgt_lens, want_logp)
            finally:
                _lib.load().lasr_roctx_range_pop()
        return self._step_features(feats, pct, targets, tgt_lens, want_logp)

    def _step_features(self, feats, pct, targets, tgt_lens, want_logp: bool = True):
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
        return tuple(None if t is None else ((t.data_ptr(), tuple(t.shape), t._version) if torch.is_tensor(t) else ("obj", id(t)))
                     for t in (wave, sample_lens, dither, aug))

    def step(self, wave, targets, tgt_lens, sample_lens=None, dither=None, aug=None, prefetch_wave=None, prefetch_lens=None,
             prefetch_dither=None, prefetch_aug=None, want_logp: bool = True, logical_len=None, prefetch_logical_len=None):
        """One training step on `wave`.  prefetch_wave: the NEXT step's waveforms (already in HBM): their log-mel features are
        computed during this step in the grid of the CTC lattice kernel (32 busy workgroups, 224 idle CUs for ~0.1 ms) and
        picked up by the next call if it passes the same tensors - the data-loader prefetch of the reference's workers;
        every step still computes exactly one batch of features."""
        pf, self._prefetched = self._prefetched, None
        if pf is not None and pf[0] == self._prefetch_key(wave, sample_lens, dither, aug):
            feats, pct = pf[1], pf[2]
        else:
            feats, pct = self.features(wave, sample_lens, dither, aug, logical_len=logical_len)
        nxt = None
        if prefetch_wave is not None:
            nf, npct = self.model.arm_prefetch(prefetch_wave, prefetch_lens, prefetch_dither, prefetch_aug, logical_len=prefetch_logical_len)
            nxt = (self._prefetch_key(prefetch_wave, prefetch_lens, prefetch_dither, prefetch_aug), nf, npct)
        out = self.step_features(feats, pct, targets, tgt_lens, want_logp=want_logp)
        self._prefetched = nxt
        return out


class GraphedTrainStep:
    """One static-shape training step captured into a hipGraph (torch.cuda.CUDAGraph over the stream the C ABI launches on) and
    replayed: ~200 kernel launches per step become one graph launch, so the step no longer depends on how fast the host can
    enqueue (measured on a box whose CPUs were shared with three other jobs: 2.9 ms/step host-bound against 2.35 ms of GPU work).

    ``step(next_wave, targets, tgt_lens)``: like the prefetching ``TrainStep.step`` - the replay trains on the features the
    PREVIOUS replay computed and computes the features of ``next_wave`` inside its CTC launch; ``targets`` belong to the batch
    being trained on.  prefetch=False: features and training step of the same ``wave`` in one replay.
    Shapes are fixed at construction; one instance per (B, L, S)."""

    def __init__(self, ts: TrainStep, B: int, L: int, S: int, ragged: bool = False, prefetch: bool = True, want_logp: bool = False,
                 inputs=None, feats_in=None, feats_out=None, wave_dtype=torch.float32, with_aug: bool = False, dither=None,
                 logical_len: Optional[int] = None):
        """inputs = (wave, sample_lens | None, targets, tgt_lens[, aug]): tensors already resident in HBM that the graph reads IN
        PLACE (``replay()`` then takes no data and nothing is copied); otherwise static buffers are allocated and ``step()``
        copies into them.  feats_in / feats_out = (feats, pct) pairs: the features this graph trains on and where it writes the
        prefetched ones - two graphs with the pairs swapped ping-pong without the end-of-step copy.
        wave_dtype: torch.float32 or torch.int16 (PCM); with_aug: a static (B, 4) SpecAugment block; dither: None, a static
        (B, L) noise tensor or an ``ops.DeviceDither`` (fresh noise per replay)."""
        self.ts, self.prefetch, self.want_logp = ts, prefetch, want_logp
        self.logical_len = logical_len        # rows of L samples whose longest utterance has logical_len <= L (ops.mel): T follows it
        dev = ts.model.device
        self.bound = inputs is not None
        self.dither = dither
        if self.bound:
            self.wave, self.lens, self.targets, self.tgt_lens = inputs[:4]
            self.aug = inputs[4] if len(inputs) > 4 else None
        else:
            self.wave = torch.zeros(B, L, dtype=wave_dtype, device=dev)
            self.lens = torch.full((B,), L, dtype=torch.int32, device=dev) if ragged else None
            self.targets = torch.zeros(B, S, dtype=torch.int64, device=dev)
            self.tgt_lens = torch.ones(B, dtype=torch.int32, device=dev)
            self.aug = torch.zeros(B, 4, dtype=torch.int32, device=dev) if with_aug else None
        self.graph = None
        self.out = None
        self.F_cur, self.pct_cur = feats_in if feats_in is not None else (None, None)
        self.feats_out = feats_out

    def _body(self):
        ts, m = self.ts, self.ts.model
        if self.prefetch:
            nf, npct = m.arm_prefetch(self.wave, self.lens, self.dither, self.aug, out=self.feats_out, logical_len=self.logical_len)
            out = ts.step_features(self.F_cur, self.pct_cur, self.targets, self.tgt_lens, want_logp=self.want_logp)
            if self.feats_out is None:
                self.F_cur.copy_(nf)
                self.pct_cur.copy_(npct)
            return out
        feats, pct = ts.features(self.wave, self.lens, self.dither, self.aug, logical_len=self.logical_len)
        return ts.step_features(feats, pct, self.targets, self.tgt_lens, want_logp=self.want_logp)

    def capture(self, first_wave: Optional[torch.Tensor] = None, first_lens: Optional[torch.Tensor] = None, warmup: int = 2,
                first_aug: Optional[torch.Tensor] = None) -> None:
        """first_wave: the batch the first replay trains on (prefetch mode: its features are computed eagerly here; not needed
        with bound inputs + feats_in, or when ``prime()`` hands the features over).
        The eager warm-up passes and the capture pass leave the training state (parameters, BN buffers, optimiser moments,
        schedule, dither / dropout counters) exactly as it was - also when the capture fails: it is snapshotted before and
        restored in a ``finally``."""
        ts, m = self.ts, self.ts.model
        if ts.world > 1 and not graph_dp_enabled():
            raise RuntimeError("graph capture of the data-parallel step (RCCL inside the graph) is switched off: LASR_GRAPH_DP=0")
        ts.use_device_schedule()
        dev_state = [m.params, m.buffers, ts.exp_avg, ts.exp_avg_sq, ts.lr_dev] + ([ts._lr_state] if ts._lr_state is not None else [])
        if hasattr(self.dither, "step"):
            dev_state.append(self.dither.step)
        if getattr(m, "drop_step", None) is not None:
            dev_state.append(m.drop_step)
        snap = [t.clone() for t in dev_state]
        sched_sd = dict(ts.schedule.state_dict()) if ts.schedule is not None else None
        gstep, bump, lr_epoch = ts.global_step, m._bump, ts._lr_epoch
        counters = {k: v.clone() for k, v in m.counters.items()}
        f = p_ = None
        try:
            if not self.bound and first_wave is not None:
                self.wave.copy_(first_wave)              # warm-up needs real audio (an all-zero wave has zero variance)
                if self.lens is not None and first_lens is not None:
                    self.lens.copy_(first_lens)
                if self.aug is not None and first_aug is not None:
                    self.aug.copy_(first_aug)
            if self.prefetch and self.F_cur is None:
                f, p_ = ts.features(first_wave, first_lens, None, first_aug, logical_len=self.logical_len)
                self.F_cur, self.pct_cur = f.clone(), p_.clone()
            elif self.prefetch and not self.bound and self.feats_out is None:
                f, p_ = self.F_cur.clone(), self.pct_cur.clone()     # (the warm-up passes overwrite the primed features)
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=m.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):            # eager warm-up on a side stream (allocator / lazy-init settle before capture)
                for _ in range(warmup):
                    self._body()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: the ingest threads (pinned allocations, H2D copies on their own stream) and RCCL's own bookkeeping may
            # call the runtime while this thread records
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):            # records the launches; nothing executes
                self.out = self._body()
            self.graph = graph
        finally:
            torch.cuda.synchronize()
            for t, s_ in zip(dev_state, snap):
                t.copy_(s_)
            if f is not None:
                self.F_cur.copy_(f)
                self.pct_cur.copy_(p_)
            if sched_sd is not None:
                ts.schedule.load_state_dict(sched_sd)
                ts.lr = ts.schedule.lr
                ts.schedule._publish()
            ts.global_step, m._bump, ts._lr_epoch = gstep, bump, lr_epoch
            m.counters.update(counters)
            ts._prefetched = None
            m.disarm_prefetch()
            torch.cuda.synchronize()

    def prime(self, feats: torch.Tensor, pct: torch.Tensor) -> None:
        """the features of the batch the next replay trains on (prefetch mode with the graph's own feature buffers)"""
        if self.F_cur is None:
            self.F_cur, self.pct_cur = feats.clone(), pct.clone()
        elif feats.data_ptr() != self.F_cur.data_ptr():
            self.F_cur.copy_(feats)
            self.pct_cur.copy_(pct)

    def step(self, wave: torch.Tensor, targets: torch.Tensor, tgt_lens: torch.Tensor, lens: Optional[torch.Tensor] = None,
             aug: Optional[torch.Tensor] = None):
        """copies the inputs into the graph's static buffers and replays; returns the static (loss, nll, logp, argmax) tensors"""
        if self.bound:
            raise RuntimeError("this graph reads its inputs in place (inputs=...): use replay()")
        self.wave.copy_(wave, non_blocking=True)
        if self.lens is not None and lens is not None:
            self.lens.copy_(lens, non_blocking=True)
        if self.aug is not None:
            if aug is not None:
                self.aug.copy_(aug, non_blocking=True)
            else:
                self.aug.zero_()
        self.targets.copy_(targets, non_blocking=True)
        self.tgt_lens.copy_(tgt_lens, non_blocking=True)
        return self.replay()

    def replay(self):
        """one training step from the captured graph (inputs as they are in the bound / static buffers right now)"""
        self.graph.replay()
        ts = self.ts
        ts.model.bump_counters(1)
        if ts.schedule is not None:
            ts.lr = ts.schedule.step()
            ts._lr_epoch = ts.schedule.last_epoch
        ts.global_step += 1
        return self.out


def graph_dp_enabled() -> bool:
    """hipGraph capture of the data-parallel step (ncclAllReduce on the library's side stream inside the capture): on by default,
    ``LASR_GRAPH_DP=0`` keeps multi-rank steps eager"""
    return os.environ.get("LASR_GRAPH_DP", "1") != "0"


def sync_with_timeout(what: str, timeout_s: float = 180.0, hard_exit: bool = True) -> None:
    """Wait for everything enqueued on the current stream, but not for ever.  The first replay of a hipGraph that holds RCCL
    collectives is the one thing of the data-parallel step that a capture *failure* cannot catch: a replay that hangs would sit in
    ``synchronize()`` until somebody's outer limit.  On timeout the rank reports and leaves with exit code 3 (``hard_exit``; no
    re-exec, no retry - the ranks' collective sequences are unknown by then) or raises ``TimeoutError``."""
    import sys
    import time
    ev = torch.cuda.Event()
    ev.record()
    t0 = time.perf_counter()
    while not ev.query():
        if time.perf_counter() - t0 > timeout_s:
            msg = ("%s did not complete within %.0f s on rank %s - a hang inside the graph-replayed data-parallel step (RCCL collectives "
                   "captured on the library's side stream).  Re-run with LASR_GRAPH_DP=0 (eager launches) or LASR_COMM=torch "
                   "(torch.distributed's all-reduce)." % (what, timeout_s, os.environ.get("RANK", "0")))
            if not hard_exit:
                raise TimeoutError(msg)
            sys.stderr.write("lightning_asr_amd: " + msg + "\n")
            sys.stderr.flush()
            os._exit(3)
        time.sleep(0.002)
