"""The slice of pytorch-lightning 1.3 the reference's train.py touches, as a small native host:
LightningModule / LightningDataModule base classes (the real ones when pytorch_lightning is
importable), ``seed_everything`` and a ``Trainer`` that runs fit/validate/test with one process per
GPU over torch.distributed (backend nccl = RCCL on ROCm; gloo on CPU for tests).

Out of scope by SURVEY §2: Comet/TensorBoard loggers, AMP (activations are bf16 in the kernels),
TPUs, profiler="simple".  Metrics go to stdout and a JSONL file."""
from __future__ import annotations

import json
import logging
import os
import random
import time
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as _pl
    HAVE_PL = True
except Exception:  # noqa: BLE001
    _pl = None
    HAVE_PL = False

logger = logging.getLogger(__name__)


def seed_everything(seed: int) -> int:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


class _ModuleBase(nn.Module):
    """Duck-typed pl.LightningModule: what train.py uses (self.log / print / save_hyperparameters /
    current_epoch / train_dataloader / load_from_checkpoint / hooks)."""

    def __init__(self):
        super().__init__()
        self.trainer: Optional["Trainer"] = None
        self.hparams: Dict[str, Any] = {}
        self._logged: Dict[str, float] = {}

    # -- what the reference calls -------------------------------------------------------------
    def save_hyperparameters(self, *args, **kwargs) -> None:
        import inspect
        frame = inspect.currentframe().f_back
        names = inspect.getargvalues(frame)
        self.hparams = {k: names.locals[k] for k in names.args if k != "self"}

    def log(self, name: str, value, on_step=None, on_epoch=None, prog_bar=False, logger=True, **kw) -> None:
        v = float(value.detach().item()) if torch.is_tensor(value) else float(value)
        self._logged[name] = v
        if self.trainer is not None:
            self.trainer._record(name, v)

    def print(self, *args) -> None:
        if self.trainer is None or self.trainer.is_global_zero:
            print(*args)

    @property
    def current_epoch(self) -> int:
        return self.trainer.current_epoch if self.trainer is not None else 0

    @property
    def global_step(self) -> int:
        return self.trainer.global_step if self.trainer is not None else 0

    def train_dataloader(self):
        return self.trainer.datamodule.train_dataloader()

    def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        pass

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, **kwargs):
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(kwargs)
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"])
        return model


class _DataModuleBase:
    def __init__(self):
        self.trainer = None

    def setup(self, stage=None):
        pass


LightningModule = _pl.LightningModule if HAVE_PL else _ModuleBase
LightningDataModule = _pl.LightningDataModule if HAVE_PL else _DataModuleBase


def _to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, (list, tuple)):
        return type(x)(_to_device(v, device) for v in x)
    return x


class GradSync:
    """Data-parallel gradient averaging over a flat buffer: bucketed SUM all-reduces issued in reverse
    layer order (the order backward produces them), 1/world folded into the optimiser's grad scale.
    Works on any backend (RCCL on the GPUs, gloo in the CPU tests)."""

    def __init__(self, bucket_bounds: List[int], process_group=None):
        import torch.distributed as dist
        self.dist = dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bounds = bucket_bounds            # ascending element offsets, first 0, last n

    def all_reduce(self, flat: torch.Tensor, async_op: bool = False):
        if self.world == 1:
            return []
        works = []
        for lo, hi in reversed(list(zip(self.bounds[:-1], self.bounds[1:]))):
            works.append(self.dist.all_reduce(flat[lo:hi], group=self.pg, async_op=async_op))
        return works

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def rank_device_index() -> int:
    """the device this rank trains on: LOCAL_RANK (one GPU per rank over nccl = RCCL).  LASR_DIST_BACKEND=gloo is the rehearsal of
    several ranks on a box with fewer GPUs (RCCL refuses two ranks on one device): the ranks then share the devices round-robin"""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LASR_DIST_BACKEND", "nccl") != "nccl":
        return local_rank % max(1, torch.cuda.device_count())
    return local_rank


class Trainer:
    """fit / test loops for the reference's LightingModule + LibriDataModule pair (train.py:233-253)."""

    def __init__(self, gpus=None, max_epochs: int = 1, resume_from_checkpoint: Optional[str] = None,
                 check_val_every_n_epoch: int = 1, limit_train_batches=1.0, limit_val_batches=1.0,
                 accelerator: Optional[str] = None, num_nodes: int = 1, default_root_dir: str = ".",
                 callbacks=None, logger=None, save_top_k: int = 3, monitor: str = "val_wer", max_steps: Optional[int] = None,
                 device: Optional[str] = None, log_every_n_steps: int = 50, **ignored):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.log_every_n_steps = log_every_n_steps      # how often the fused loop reads its device-side metric accumulators
        self.callbacks = list(callbacks or [])          # objects with on_train_batch_end(trainer) / on_train_epoch_end(trainer)
        self.fused = None                               # the FusedLoop of the last fit (None: autograd route)
        self.resume_from_checkpoint = resume_from_checkpoint
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.root = default_root_dir
        self.save_top_k, self.monitor = save_top_k, monitor
        self.current_epoch = 0
        self.global_step = 0
        self.datamodule = None
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device(device) if device else torch.device("cuda", rank_device_index())
        n_gpus = len(gpus) if isinstance(gpus, (list, tuple)) else (int(gpus) if isinstance(gpus, (int, str)) and str(gpus).lstrip("-").isdigit() else None)
        if n_gpus is not None and n_gpus > 1 and self.world == 1:
            raise RuntimeError("gpus=%s asks for %d data-parallel ranks but WORLD_SIZE is 1: this host runs one process per GPU.  "
                               "`python -m lightning_asr_amd.train train.gpus=%d` starts them itself (lightning_asr_amd/launch.py, before "
                               "anything touches the GPU); a script that builds its own Trainer calls launch.maybe_launch(n, argv) first or "
                               "runs under `python -m torch.distributed.run --nproc-per-node %d`"
                               % (gpus, n_gpus, n_gpus, n_gpus))
        self.history: List[Dict[str, Any]] = []
        self._epoch_metrics: Dict[str, List[float]] = {}
        self._best: List[tuple] = []
        self.callback_metrics: Dict[str, float] = {}

    @property
    def is_global_zero(self) -> bool:
        return self.rank == 0

    def _record(self, name: str, v: float) -> None:
        self._epoch_metrics.setdefault(name, []).append(v)
        self.callback_metrics[name] = v

    def _limit(self, n: int, lim) -> int:
        if isinstance(lim, float):
            return max(1, int(n * lim)) if lim < 1.0 else n
        return min(n, int(lim))

    def _init_dist(self):
        import torch.distributed as dist
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # nccl = RCCL, one GPU per rank.  LASR_DIST_BACKEND=gloo: rehearsal with several ranks sharing one GPU (RCCL refuses
            # two ranks on a device) - the tests' world_size-2 fit
            backend = os.environ.get("LASR_DIST_BACKEND", "nccl" if self.device.type == "cuda" else "gloo")
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=self.device)
            else:
                dist.init_process_group(backend)
        return dist

    def _batch(self, batch, dm, idx=0):
        from .data_module import WaveBatch
        if not isinstance(batch, WaveBatch):     # ragged host waveforms go up in ONE pinned copy inside on_after_batch_transfer
            batch = _to_device(batch, self.device)
        if hasattr(dm, "on_after_batch_transfer"):
            batch = dm.on_after_batch_transfer(batch, idx)
        return batch

    def _eval_batches(self, loader, dm, n: int):
        """the reference's 5-tuples (data_module.py:248) for a validation / test loader: through the native ingest (wav files ->
        pinned ring -> int16 H2D -> device front-end) when the dataset is the stock one, else through the DataLoader"""
        from .ingest import fast_ingest_ok
        if self.device.type == "cuda" and hasattr(dm, "audio_parser") and fast_ingest_ok(getattr(loader, "dataset", None)) \
                and getattr(loader, "batch_sampler", None) is not None and os.environ.get("LASR_NATIVE_INGEST", "1") != "0":
            from .fused_fit import make_source
            src = make_source(dm, loader, self.device, False, n, getattr(dm, "dev_max_duration", 40) or 40, getattr(dm, "dev_bs", 16))
            try:
                for db in src:
                    if db.ready is not None and not db.ready.query():
                        torch.cuda.current_stream().wait_event(db.ready)
                    inputs, pct = dm.audio_parser.features_device(db.pcm, db.lens, None, logical_len=db.L)
                    yield inputs, db.targets, pct, db.sizes, db.paths
                    src.release(db)          # (after the consumer has enqueued everything that reads the slot's targets / sizes)
            finally:
                src.close()
            return
        for k, batch in enumerate(loader):
            if k >= n:
                return
            yield self._batch(batch, dm, k)

    def _save(self, model, opt, sched, name: str) -> str:
        d = os.path.join(self.root, "checkpoints")
        os.makedirs(d, exist_ok=True)
        ckpt = {"epoch": self.current_epoch, "global_step": self.global_step, "state_dict": model.state_dict(),
                "optimizer_states": [opt.state_dict()], "lr_schedulers": [sched.state_dict()] if sched else [],
                "hyper_parameters": dict(getattr(model, "hparams", {})), "pytorch-lightning_version": "lasr-native"}
        model.on_save_checkpoint(ckpt)
        path = os.path.join(d, name)
        torch.save(ckpt, path)
        return path

    def _check_device(self, model):
        native = getattr(getattr(model, "encoder", None), "native", None)
        if native is not None and self.device.type == "cuda" and native.device != self.device:
            raise RuntimeError("model lives on %s but this rank trains on %s: construct it with device=%r (train.main does)"
                               % (native.device, self.device, str(self.device)))

    def fit(self, model, datamodule=None):
        self._check_device(model)
        dist = self._init_dist()
        self.datamodule = datamodule
        model.trainer = self
        datamodule.trainer = self
        datamodule.setup("fit")
        optimizers, schedulers = model.configure_optimizers()
        opt = optimizers[0]
        sched = schedulers[0]["scheduler"] if schedulers else None
        start_epoch = 0
        if self.resume_from_checkpoint:
            ckpt = torch.load(self.resume_from_checkpoint, map_location="cpu", weights_only=False)
            model.load_state_dict(ckpt["state_dict"])
            opt.load_state_dict(ckpt["optimizer_states"][0])
            if sched and ckpt.get("lr_schedulers"):
                sched.load_state_dict(ckpt["lr_schedulers"][0])
            start_epoch, self.global_step = ckpt["epoch"] + 1, ckpt["global_step"]
        native = getattr(getattr(model, "encoder", None), "native", None)
        sync = None
        from .fused_fit import FusedLoop, fused_eligible
        fused = None
        if fused_eligible(model, opt, sched):
            # the stock module: training_step + backward + DDP all-reduce + optimizer.step + scheduler.step run as the fused
            # native step (fused_fit.py); parameters / buffers are broadcast from rank 0 like Lightning's DDP wrap does
            fused = FusedLoop(self, model, datamodule, opt, sched)
            fused.ts.broadcast_parameters()
            if fused.ts.world > 1:
                self._install_metric_reduce(model, fused.ts)
        elif self.world > 1:
            dist.broadcast(native.params, 0)
            dist.broadcast(native.buffers, 0)
            sync = GradSync(native.bucket_bounds(), None)
            opt.grad_scale = sync.grad_scale
            self._install_metric_reduce(model, None)
        self.fused = fused
        loader = datamodule.train_dataloader()
        if self.world > 1:
            loader = datamodule.train_dataloader(distributed=(self.world, self.rank))
        os.makedirs(self.root, exist_ok=True)
        log_path = os.path.join(self.root, "metrics.jsonl")

        def on_step(tr):
            for cb in tr.callbacks:
                if hasattr(cb, "on_train_batch_end"):
                    cb.on_train_batch_end(tr)

        for epoch in range(start_epoch, self.max_epochs):
            self.current_epoch = epoch
            self._epoch_metrics = {}
            model.train()
            for smp in (getattr(loader, "sampler", None), getattr(loader, "batch_sampler", None)):
                if smp is not None and hasattr(smp, "set_epoch"):
                    smp.set_epoch(epoch)
            n = self._limit(len(loader), self.limit_train_batches)
            t0 = time.time()
            if fused is not None:
                fused.run_epoch(loader, n, on_step)
                for k, v in fused.read_metrics(reset=True).items():
                    if not k.endswith("_step"):
                        self._record(k, v)
            else:
                for batch_idx, batch in enumerate(loader):
                    if batch_idx >= n or (self.max_steps and self.global_step >= self.max_steps):
                        break
                    batch = self._batch(batch, datamodule)
                    loss = model.training_step(batch, batch_idx)
                    opt.zero_grad()
                    loss.backward()
                    if sync is not None:
                        sync.all_reduce(native.grads)
                    opt.step()
                    if sched is not None:
                        sched.step()
                    self.global_step += 1
                    on_step(self)
            rec = {"epoch": epoch, "global_step": self.global_step, "train_time_s": time.time() - t0,
                   "lr": opt.param_groups[0]["lr"]}
            rec.update({k: float(np.mean(v)) for k, v in self._epoch_metrics.items()})
            if (epoch + 1) % self.check_val_every_n_epoch == 0:
                rec.update(self.validate(model, datamodule))
            self.history.append(rec)
            if self.is_global_zero:
                with open(log_path, "a") as f:
                    f.write(json.dumps(rec) + "\n")
                self._save(model, opt, sched, "last.ckpt")
                if self.monitor in rec:
                    name = "asr-epoch=%02d-%s=%.2f.ckpt" % (epoch, self.monitor, rec[self.monitor])
                    self._best.append((rec[self.monitor], self._save(model, opt, sched, name)))
                    self._best.sort(key=lambda x: x[0])
                    for _, p in self._best[self.save_top_k:]:
                        if os.path.exists(p):
                            os.remove(p)
                    self._best = self._best[:self.save_top_k]
            if self.max_steps and self.global_step >= self.max_steps:
                break
        return self.history

    @torch.no_grad()
    def validate(self, model, datamodule) -> Dict[str, float]:
        model.eval()
        self._epoch_metrics = {}
        wer = getattr(model, "wer", None)
        if wer is not None and hasattr(wer, "reset"):
            wer.reset()
        loader = datamodule.val_dataloader()
        n = self._limit(len(loader), self.limit_val_batches)
        outs = []
        for batch_idx, batch in enumerate(self._eval_batches(loader, datamodule, n)):
            outs.append(model.validation_step(batch, batch_idx))
        model.validation_epoch_end(outs)
        model.train()
        rec = {k: float(np.mean(v)) for k, v in self._epoch_metrics.items()}
        rec = self._sync_epoch_metrics(rec)
        if wer is not None and hasattr(wer, "compute_total") and outs:
            rec["val_wer_total"] = float(wer.compute_total())     # corpus-level: summed edit distances / summed reference units, all ranks
        return rec

    # ---- data-parallel metric agreement -----------------------------------------------------------------------------------
    def _all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """SUM of a small f32 device vector over the ranks: the library's RCCL communicator when the fused step owns one,
        torch.distributed otherwise (gloo rehearsals, the autograd route)"""
        comm = getattr(getattr(self.fused, "ts", None), "comm", None)
        if comm is not None:
            comm.all_reduce(t)
            comm.wait()
        else:
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size() > 1:
                if dist.get_backend() == "gloo" and t.is_cuda:
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
                else:
                    dist.all_reduce(t)
        return t

    def _install_metric_reduce(self, model, ts) -> None:
        """utils/asr_metrics.py:114-115 dist_reduce_fx='sum' on `scores` / `words`: WER.compute() sums both over the ranks"""
        wer = getattr(model, "wer", None)
        if wer is None or self.world <= 1:
            return

        def reduce(scores, words):
            dev = self.device if self.device.type == "cuda" else scores.device
            v = torch.stack([scores.to(dev).float(), words.to(dev).float()])
            self._all_reduce_sum(v)
            return v[0], v[1]
        wer.world_reduce = reduce               # used by WER.compute_total() at the end of an epoch, not on every step

    def _sync_epoch_metrics(self, rec: Dict[str, float]) -> Dict[str, float]:
        """world > 1: every rank reports the MEAN over the ranks of its epoch metrics, so `val_wer` (what ModelCheckpoint
        monitors, train.py:210-212) and the checkpoints it selects agree across the ranks"""
        if self.world <= 1 or not rec:
            return rec
        keys = sorted(rec)
        dev = self.device if self.device.type == "cuda" else torch.device("cpu")
        v = torch.tensor([rec[k] for k in keys], dtype=torch.float32, device=dev)
        self._all_reduce_sum(v)
        v = (v / self.world).cpu()
        return {k: float(v[i]) for i, k in enumerate(keys)}

    @torch.no_grad()
    def test(self, model, test_dataloaders=None, datamodule=None):
        model.trainer = self
        model.eval()
        dm = datamodule or self.datamodule
        if test_dataloaders is None and dm is not None and dm is not self.datamodule:
            dm.trainer = self
            dm.setup("test")                       # PL calls setup('test') on a datamodule handed to .test()
        loader = test_dataloaders if test_dataloaders is not None else dm.test_dataloader()
        outs = [model.test_step(batch, i) for i, batch in enumerate(self._eval_batches(loader, dm, len(loader)))]
        model.test_epoch_end(outs)
        return outs

    def teardown(self) -> None:
        """end of a multi-rank run (the CLI calls it; Lightning tears its DDP plugin down the same way): every rank arrives, the
        library's communicator and the torch.distributed group are released in that order, so no rank exits under a peer's collective"""
        import torch.distributed as dist
        if self.world > 1 and dist.is_initialized():
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            dist.barrier()
            comm = getattr(getattr(self.fused, "ts", None), "comm", None)
            if comm is not None:
                comm.close()
            dist.destroy_process_group()
