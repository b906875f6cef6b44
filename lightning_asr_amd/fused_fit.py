"""``Trainer.fit``'s hot loop for the stock ``LightingModule``: the reference's ``training_step`` + Lightning's
backward / DDP all-reduce / ``optimizer.step`` / ``scheduler.step`` (train.py:64-86,233-252, conf/conf.yaml:30) executed as the
fused native step - the path ``bench.py`` measures:

    manifest --host threads--> int16 PCM in a pinned ring --copy stream--> HBM
      -> TrainStep.step: features of batch i+1 inside the CTC launch of batch i, lasr_model_loss_backward (lean head for
         large vocabularies: no (B, T', C) f32 log-probs), bucketed RCCL all-reduce on the library's side stream overlapped
         with backward (world > 1), fused NovoGrad, LR schedule on the device; replayed from a hipGraph once a batch shape repeats
      -> greedy decode + edit distance + loss/WER accumulation on the device (train_loss / train_wer as the reference logs them,
         read back every ``log_every_n_steps`` steps instead of every step).

A module that overrides ``training_step`` / ``forward`` / ``configure_optimizers`` with something else keeps the autograd route
of ``Trainer.fit`` (``training_step`` -> ``loss.backward()`` -> ``optimizer.step()``)."""
from __future__ import annotations

import logging
import os
import time
from collections import deque
from typing import Dict, Optional

import torch

from . import _lib, ops
from .ingest import BatchProducer, DevBatch, DeviceFeeder, PinnedRing, SR, fast_ingest_ok
from .step import GraphedTrainStep, TrainStep, graph_dp_enabled

logger = logging.getLogger(__name__)


def fused_eligible(model, optimizer, scheduler) -> bool:
    """the stock module (train.py's training_step / forward untouched) over a native encoder, optimised by the HIP NovoGrad
    with the cosine-warm-up schedule (or none)"""
    if os.environ.get("LASR_TRAINER_FUSED", "1") == "0":
        return False
    from .schedule import CosineAnnealingWarmupRestarts
    from .scheduler.novograd import Novograd
    from .train import LightingModule as LM
    if not isinstance(model, LM):
        return False
    if type(model).training_step is not LM.training_step or type(model).forward is not LM.forward or type(model)._shared is not LM._shared:
        return False
    if getattr(getattr(model, "encoder", None), "native", None) is None:
        return False
    if not isinstance(optimizer, Novograd) or (scheduler is not None and not isinstance(scheduler, CosineAnnealingWarmupRestarts)):
        return False
    return True


class HostWaveSource:
    """Fallback ingest for a custom dataset: the DataLoader's WaveBatch (ragged f32 waves) padded on the host and uploaded as f32"""

    def __init__(self, loader, device, limit: int):
        self.loader, self.device, self.limit = loader, torch.device(device), limit

    def __iter__(self):
        for k, batch in enumerate(self.loader):
            if k >= self.limit:
                return
            waves, targets, sizes, paths, mask = batch
            B, L = len(waves), max(int(w.numel()) for w in waves)
            host = torch.zeros(B, L).pin_memory()
            lens = torch.empty(B, dtype=torch.int32)
            leads = getattr(batch, "leads", None)            # crop_raw's lead-in samples (crop after pre-emphasis, data_module.py:157-159)
            n_real, longest = 0, 1
            for i, w in enumerate(waves):
                host[i, :w.numel()] = w
                ld = int(leads[i]) if leads is not None else 0
                lens[i] = (w.numel() - ld) | (_lib.LEN_LEAD if ld else 0)
                n_real += w.numel() - ld
                longest = max(longest, int(w.numel()) - ld)
            db = DevBatch()
            db.L, db.pitch = longest, L                      # T = the frames of the longest UTTERANCE; rows may be one lead-in sample wider
            db.pcm = host.to(self.device, non_blocking=True)
            db.lens, db.sizes, db.targets = lens.to(self.device), sizes.to(self.device), targets.to(self.device)
            db.aug = None
            db.paths, db.B, db.ld, db.S, db.seconds = paths, B, L, targets.shape[1], float(n_real) / SR
            db.ready, db.dslot, db.index, db.key = None, -1, k, (B, L, targets.shape[1], False)
            db.mask, db.waited = mask, True
            yield db

    def release(self, db) -> None:
        pass

    def close(self) -> None:
        pass


class NativeSource:
    """manifest -> pinned ring -> device ring (ingest.py), ``depth`` batches ahead of the consumer"""

    def __init__(self, dataset, index_batches, audio_parser, device, batch_size: int, max_seconds: float, mask: bool,
                 n_threads: int, limit: int, crop: Optional[bool] = None):
        index_batches = list(index_batches)[:limit]
        cap = int(batch_size * (int(max_seconds * SR) + 64))
        self.ring = PinnedRing(4, cap, 8 * batch_size + 2 * batch_size * 256)
        self.feeder = DeviceFeeder(self.ring, device, n_slots=4)
        self.producer = BatchProducer(dataset, index_batches, self.ring, mask, audio_parser, n_threads=n_threads, crop=crop,
                                      feeder=self.feeder)
        self.mask = mask
        self.n_threads = n_threads
        self.n = len(index_batches)

    def __iter__(self):
        self.producer.start()
        try:
            while True:
                db = self.producer.out.get()          # uploaded by the producer thread; `ready` orders the consumer behind the copies
                if db is None:
                    return
                if isinstance(db, BaseException):
                    raise db
                yield db
        finally:
            self.producer.stop()

    def release(self, db) -> None:
        self.feeder.release(db)

    def close(self) -> None:
        self.producer.stop()
        self.feeder.close()


def ingest_threads_for_rank(requested: int) -> int:
    """host threads of this rank's wav reader: `data.num_worker` (conf/conf.yaml:14: 6 DataLoader workers PER RANK in the reference),
    capped by this rank's share of the cores the process may run on - cores / ranks on this node, minus one for the thread that
    enqueues the step - so that 8 ranks x (1 + num_worker) threads never oversubscribe a small host.  An explicit
    LASR_INGEST_THREADS is taken as given.  LASR_PIN_RANKS=1 additionally pins the rank (all its threads) to its contiguous share of
    the allowed cores - off by default: the right cores are the ones next to the rank's GPU, which only the site's topology knows."""
    if os.environ.get("LASR_INGEST_THREADS") is not None:
        return max(1, int(os.environ["LASR_INGEST_THREADS"]))
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    share = max(1, len(allowed) // local_world)
    if os.environ.get("LASR_PIN_RANKS") == "1" and local_world > 1 and hasattr(os, "sched_setaffinity"):
        lr = int(os.environ.get("LOCAL_RANK", "0")) % local_world
        os.sched_setaffinity(0, allowed[lr * share:(lr + 1) * share])
    return max(1, min(max(1, int(requested or 0)), max(2, share - 1)))


def make_source(datamodule, loader, device, mask: bool, limit: int, max_seconds: float, batch_size: int, crop: Optional[bool] = None):
    ds = loader.dataset
    if fast_ingest_ok(ds) and os.environ.get("LASR_NATIVE_INGEST", "1") != "0":
        n_threads = ingest_threads_for_rank(int(getattr(datamodule, "num_worker", 0) or 0))
        return NativeSource(ds, loader.batch_sampler, datamodule.audio_parser, device, batch_size, max_seconds, mask,
                            n_threads=max(n_threads, 1), limit=limit, crop=crop)
    return HostWaveSource(loader, device, limit)


class FusedLoop:
    def __init__(self, trainer, model, datamodule, optimizer, scheduler):
        self.trainer, self.model, self.dm = trainer, model, datamodule
        self.native = model.encoder.native
        dev = self.native.device
        self.ts = TrainStep.from_optimizer(self.native, optimizer, scheduler)
        if scheduler is not None:
            self.ts.use_device_schedule()
        self.dither = ops.DeviceDither(int(torch.initial_seed()) ^ (0x9E3779B97F4A7C15 * (trainer.rank + 1)), dev)
        self.acc = torch.zeros(8, dtype=torch.float64, device=dev)
        self.use_graph = os.environ.get("LASR_TRAINER_GRAPH", "1") != "0" and (self.ts.world == 1 or graph_dp_enabled())
        self.graphs: Dict[tuple, GraphedTrainStep] = {}
        self.seen: Dict[tuple, int] = {}
        self._pf = None            # (batch index, feats, pct): features of the next batch, computed by the previous step
        self.graph_steps = self.eager_steps = 0
        self.audio_seconds = 0.0
        self.samples_real = self.samples_padded = 0      # padding bookkeeping: sum of valid samples / of B * row pitch
        self.ingest_wait_s = 0.0                         # host time spent waiting for the next batch from the ingest (0 = never starved)
        self.host_step_s = 0.0                           # host WALL time inside step(): includes every moment the runtime made the
                                                         # thread wait for room in the GPU's queue (a GPU-bound loop shows ~the step time here)
        self.host_cpu_s = 0.0                            # CPU time of this thread inside step() (time.thread_time): the work the host did
        self.on_host_batch = getattr(trainer, "_fused_on_batch", None)  # test hook: called with every DevBatch before it is trained on

    # ---- one step ------------------------------------------------------------------------------------------------------
    def _feats_for(self, cur: DevBatch):
        pf, self._pf = self._pf, None
        if pf is not None and pf[0] == cur.index:
            return pf[1], pf[2]
        return self.ts.features(cur.pcm, cur.lens, self.dither, cur.aug, logical_len=cur.L)

    def _graph_for(self, cur: DevBatch, nxt: DevBatch) -> Optional[GraphedTrainStep]:
        key = cur.key
        if not self.use_graph or nxt is None or nxt.key != key or cur.pcm.dtype != nxt.pcm.dtype:
            return None
        g = self.graphs.get(key)
        if g is not None:
            return g if g.graph is not None else None
        self.seen[key] = self.seen.get(key, 0) + 1
        # capture once a shape keeps coming back - and only a shape that is COMMON (>= 8 % of the steps so far): a capture costs
        # ~20 ms with the GPU idle behind it, which a shape met every fiftieth step never earns back.  Under the reference's random
        # crop the batches of a 10 s corpus fall into frame-count classes of 44 / 36 / 14 / 4 % (round 5: one row pitch per class,
        # ingest.py), and a replay needs this batch AND the next in the same class: the two common classes are captured within the
        # first ~25 steps, the rare ones stay eager.
        steps = self.graph_steps + self.eager_steps + 1
        if self.seen[key] < 3 or self.seen[key] < 0.08 * steps or len(self.graphs) >= 16:
            return None
        B, L, S, with_aug = key
        g = GraphedTrainStep(self.ts, B, L, S, ragged=True, prefetch=True, want_logp=False, wave_dtype=cur.pcm.dtype,
                             with_aug=with_aug, dither=self.dither, logical_len=cur.L)
        self.graphs[key] = g
        try:
            g.targets.copy_(cur.targets)
            g.tgt_lens.copy_(cur.sizes)
            # world > 1: NO eager warm-up passes.  They would issue real all-reduces on THIS rank only (each rank decides from its
            # own batch shapes when to capture), pairing with the other ranks' training all-reduces.  The capture pass itself
            # executes nothing, and the replay that follows issues exactly the one set of bucket collectives this step owes the
            # group - the same sequence an eager rank issues - so ranks may capture at different steps, or never.  The warm-ups
            # exist to settle lazy initialisation and the allocator; this shape has already run eagerly >= 2 times here.
            g.capture(first_wave=cur.pcm, first_lens=cur.lens, first_aug=cur.aug, warmup=0 if self.ts.world > 1 else 2)
        except Exception as e:  # noqa: BLE001 - a runtime that refuses the capture: this shape stays eager
            logger.warning("hipGraph capture failed for batch shape %s (%s): eager launches", key, e)
            g.graph = None
            return None
        return g

    def step(self, cur: DevBatch, nxt: Optional[DevBatch], batch_idx: int):
        ts, native = self.ts, self.native
        stream = torch.cuda.current_stream()
        for b in (cur, nxt):          # once per batch: the compute stream is ordered behind the batch's H2D copies (issued two steps ahead)
            if b is not None and not b.waited:
                b.waited = True
                # (no `ready.query()` first: hipEventQuery drains the runtime's pending submissions - measured 1.13 ms of host time
                #  per call here, half of the whole step's enqueue time, profiles/r04_trainer_host_profile.txt; a wait on an event
                #  that has already completed costs the stream nothing)
                if b.ready is not None:
                    stream.wait_event(b.ready)
        if self.on_host_batch is not None:
            self.on_host_batch(cur)
        feats, pct = self._feats_for(cur)
        g = self._graph_for(cur, nxt)
        if g is not None:
            g.prime(feats, pct)
            loss, nll, _, am = g.step(nxt.pcm, cur.targets, cur.sizes, nxt.lens, nxt.aug)
            if ts.world > 1 and not getattr(g, "_replayed_once", False):
                # the first replay of a graph with collectives in it must be seen to complete (a capture failure falls back to
                # eager launches above; a replay hang would otherwise surface at the next metrics read, without a reason)
                from .step import sync_with_timeout
                sync_with_timeout("Trainer.fit: the first replay of the captured data-parallel step", hard_exit=False)
                g._replayed_once = True
            self._pf = (nxt.index, g.F_cur, g.pct_cur)
            self.graph_steps += 1
        else:
            nf = None
            if nxt is not None:
                nf = native.arm_prefetch(nxt.pcm, nxt.lens, self.dither, nxt.aug, logical_len=nxt.L)
            loss, nll, _, am = ts.step_features(feats, pct, cur.targets, cur.sizes, want_logp=False)
            self._pf = (nxt.index, nf[0], nf[1]) if nf is not None else None
            self.eager_steps += 1
        self.audio_seconds += cur.seconds
        self.samples_real += int(round(cur.seconds * SR))
        self.samples_padded += cur.B * cur.ld
        # train.py:79-81: self.log('train_loss'), self.log('train_wer') - decode, distance and accumulation stay on the device
        wer = self.model.wer
        t_lens = native.tap("lens")
        if wer.device_path(am, cur.targets):
            dist, units = wer.device_distances(am, cur.targets, cur.sizes, t_lens)
            ops.step_metrics(loss, dist, units, self.acc)
            host_wer = None
        else:        # labels the device units cannot express: the host path of the metric, once per step like the reference
            host_wer = float(wer(am, cur.targets, cur.sizes, t_lens))
            self.trainer._record("train_wer", host_wer)
            self.trainer._record("train_loss", float(loss.item()))
        if batch_idx % 50 == 0:     # train.py:82-85
            logging.info("pred:" + wer.ctc_decoder_predictions_tensor(am, t_lens)[0])
            logging.info("true:" + wer.decode_reference(cur.targets, cur.sizes)[0])
        return loss

    def read_metrics(self, reset: bool = False) -> Dict[str, float]:
        a = self.acc.cpu()
        out = {}
        if a[2] > 0:
            out = {"train_loss": float(a[0] / a[2]), "train_wer": float(a[1] / a[2]), "train_loss_step": float(a[3]),
                   "train_wer_step": float(a[4])}
        if reset:
            self.acc.zero_()
        return out

    def _reserve_workspace(self) -> None:
        """Size the model workspace ONCE, for the longest clip and transcript of the training set.  The workspace only ever grows,
        but with ragged batches (length buckets, random crops) it grew in as many steps as the batch order happened to offer longer
        clips - each a multi-GB allocation with the GPU idle behind it: the same cfg5 run took 3.2 or 4.2 ms per step depending on
        that order (round 4; the kernels' time was 3.0 ms either way)."""
        if getattr(self, "_ws_reserved", False):
            return
        self._ws_reserved = True
        items = getattr(getattr(self.dm, "train_datasets", None), "datasets", None)
        if not items:
            return
        try:
            max_dur = max(float(d["duration"]) for d in items)
            s_max = max(len(d["text"]) for d in items)
        except (KeyError, TypeError, ValueError):
            return
        cap = getattr(self.dm, "train_max_duration", None)
        if cap:
            max_dur = min(max_dur, float(cap))
        t_in = ops.mel_num_frames(int(max_dur * 16000 + 0.5) + 1)
        self.native.workspace(int(getattr(self.dm, "train_bs", 32)), t_in, max(s_max, 1))

    # ---- one epoch -----------------------------------------------------------------------------------------------------
    def run_epoch(self, loader, n_batches: int, on_step=None) -> None:
        tr, dm = self.trainer, self.dm
        self._reserve_workspace()
        src = make_source(dm, loader, self.native.device, True, n_batches, getattr(dm, "train_max_duration", 16.7) or 16.7,
                          getattr(dm, "train_bs", 32), crop=getattr(dm, "train_crop", True))
        self.source_kind = type(src).__name__
        self.ingest_threads = getattr(src, "n_threads", None)
        self._pf = None
        window = deque()
        it = iter(src)
        try:
            for _ in range(2):                      # two batches in flight ahead of the step: H2D of i+2 rides under step i
                b = next(it, None)
                if b is not None:
                    window.append(b)
            batch_idx = 0
            while window:
                if tr.max_steps and tr.global_step >= tr.max_steps:
                    break
                cur = window.popleft()
                nxt = window[0] if window else None
                t0 = time.perf_counter()
                b = next(it, None)
                t1 = time.perf_counter()
                c1 = time.thread_time()
                if b is not None:
                    window.append(b)
                self.step(cur, nxt, batch_idx)
                self.ingest_wait_s += t1 - t0
                self.host_step_s += time.perf_counter() - t1
                self.host_cpu_s += time.thread_time() - c1
                src.release(cur)
                tr.global_step += 1
                batch_idx += 1
                if tr.log_every_n_steps and tr.global_step % tr.log_every_n_steps == 0:
                    m = self.read_metrics()
                    for k in ("train_loss", "train_wer"):
                        if k + "_step" in m:
                            tr.callback_metrics[k] = m[k + "_step"]
                if on_step is not None:
                    on_step(tr)
        finally:
            it.close()
            src.close()
