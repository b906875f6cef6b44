"""``LightingModule`` and ``main`` of the reference's train.py, on the HIP hot path.

Same constructor keywords (train.py:186-188), hooks and logged metric names.  ``python -m
lightning_asr_amd.train key=value ...`` loads conf/conf.yaml (Hydra-style overrides) and runs fit+test."""
from __future__ import annotations

import logging
import os
import sys
from typing import Any, Dict, List

import torch

from . import ops
from .config import load_config
from .data_module import LibriDataModule
from .lightning_compat import LightningModule, Trainer, rank_device_index, seed_everything
from .scheduler.cosine_annearing_with_warmup import CosineAnnealingWarmupRestarts
from .scheduler.novograd import Novograd
from .utils.asr_metrics import WER

logger = logging.getLogger(__name__)

MODEL_FILES = {"plain": "QuartNet", "context": "QuartNetContext", "context_se": "QuartNetContextSE"}


def _model_class(variant: str):
    import importlib
    return importlib.import_module(".models." + MODEL_FILES[variant], __package__).MyModel2


class CTCLoss:
    """``torch.nn.CTCLoss(blank, reduction='none')`` call signature (train.py:77-78,196) on the HIP lattice kernels.
    ``loss(log_probs (T,B,C), targets, input_lengths, target_lengths) -> (B,)`` with autograd."""

    def __init__(self, blank: int, reduction: str = "none"):
        if reduction != "none":
            raise NotImplementedError("the reference uses reduction='none' (train.py:196)")
        self.blank = blank

    def __call__(self, log_probs_tbc, targets, input_lengths, target_lengths):
        return _CTCFn.apply(log_probs_tbc, targets, input_lengths, target_lengths, self.blank)


class _CTCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp_tbc, targets, in_lens, tgt_lens, blank):
        logp = lp_tbc.transpose(0, 1).contiguous()               # kernels take (B,T,C)
        dev = logp.device
        ones = torch.ones(logp.shape[0], dtype=torch.float32, device=dev)
        nll, grad = ops.ctc_loss(logp, targets.to(dev).contiguous(), in_lens.to(dev, torch.int32).contiguous(),
                                 tgt_lens.to(dev, torch.int32).contiguous(), blank, True, ones)
        ctx.save_for_backward(grad)
        return nll

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g.view(-1, 1, 1)).transpose(0, 1), None, None, None, None


class LightingModule(LightningModule):
    def __init__(self, learning_rate=5e-3, weight_decay=1e-4, labels=None, total_epoch=50, drop_rate: float = 0.,
                 mask: bool = False, use_cer=False, variant: str = "plain", act: str = "relu", dtype: str = "f32",
                 device: str = "cuda", warmup_steps: int = 1000):
        super().__init__()
        self.learning_rate = learning_rate
        self.weight_decay = weight_decay
        self.labels = labels
        self.total_epoch = total_epoch
        self.warmup_steps = warmup_steps          # the reference hard-codes 1000 (train.py:55)
        self.save_hyperparameters()
        self.wer = WER(vocabulary=self.labels, use_cer=use_cer)
        self.loss = CTCLoss(blank=len(self.labels), reduction="none")          # the last class is the blank
        self.encoder = _model_class(variant)(labels=self.labels, drop_rate=drop_rate, mask=mask, act=act, device=device,
                                             dtype=torch.float32 if dtype == "f32" else torch.bfloat16)

    def forward(self, inputs, percentage):
        return self.encoder(inputs, percentage)  # N*L'*C

    def configure_optimizers(self):
        self.print("设置学习率" + str(self.learning_rate))
        novo_optim = Novograd(self.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay, betas=(0.8, 0.5))
        lr_scheduler = CosineAnnealingWarmupRestarts(novo_optim, first_cycle_steps=self.total_epoch * len(self.train_dataloader()),
                                                     cycle_mult=2, max_lr=self.learning_rate, min_lr=1e-4,
                                                     warmup_steps=self.warmup_steps, gamma=0.5)
        return [novo_optim], [{"scheduler": lr_scheduler, "interval": "step", "monitor": "val_loss"}]

    def _encode(self, inputs, percentage):
        btf = getattr(inputs, "_lasr_btf", None)
        if btf is not None:        # channels-last twin written by the mel kernel: skip the layout pass
            return self.encoder.forward_features(btf, percentage)
        return self.encoder(inputs, percentage)

    def _shared(self, batch):
        inputs, trans, percentage, trans_lengths = batch[0], batch[1], batch[2], batch[3]
        out = self._encode(inputs, percentage)
        t_lengths = ops.mask_lengths(percentage.to(out.device, torch.float32).contiguous(), out.size(1))   # (T'*pct).int()
        loss = torch.mean(self.loss(out.transpose(0, 1), trans, t_lengths, trans_lengths))
        return out, loss, t_lengths, trans, trans_lengths

    def training_step(self, batch, batch_idx):
        out, loss, t_lengths, trans, trans_lengths = self._shared(batch)
        argmax = self.encoder.last_argmax                       # fused with log_softmax on the device
        self.log("train_loss", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        self.log("train_wer", self.wer(argmax, trans, trans_lengths, t_lengths), on_step=True, on_epoch=True, prog_bar=True,
                 logger=True)
        if batch_idx % 50 == 0:
            logging.info("pred:" + self.wer.ctc_decoder_predictions_tensor(argmax, t_lengths)[0])
            logging.info("true:" + self.wer.decode_reference(trans, trans_lengths)[0])
        return loss

    def validation_step(self, batch, batch_idx):
        out, loss, t_lengths, trans, trans_lengths = self._shared(batch)
        argmax = self.encoder.last_argmax
        wer = self.wer(argmax, trans, trans_lengths, t_lengths)
        self.log("val_wer", wer, on_epoch=True, prog_bar=True, logger=True)
        self.log("val_loss", loss, on_epoch=True, prog_bar=True, logger=True)
        return {"val_loss": loss, "input": batch[0], "val_wer": wer,
                "pred": self.wer.ctc_decoder_predictions_tensor(argmax, t_lengths),
                "true": self.wer.decode_reference(trans, trans_lengths), "path": batch[-1]}

    def test_step(self, batch, batch_idx):
        out, loss, t_lengths, trans, trans_lengths = self._shared(batch)
        argmax = self.encoder.last_argmax
        return {"test_loss": loss, "input": batch[0], "test_wer": self.wer(argmax, trans, trans_lengths, t_lengths),
                "pred": self.wer.ctc_decoder_predictions_tensor(argmax, t_lengths),
                "true": self.wer.decode_reference(trans, trans_lengths), "path": batch[-1]}

    def test_epoch_end(self, outputs: List[Any]) -> None:
        total = sum(float(o["test_wer"]) for o in outputs)
        logger.info("测试wer：" + str(total / (len(outputs) + 1e-9)))

    def validation_epoch_end(self, outputs: List[Any]) -> None:
        total = sum(float(o["val_wer"]) for o in outputs)
        logger.info("验证集wer：" + str(total / (len(outputs) + 1e-9)))

    def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        logger.info("保存一个checkpoint epoch={:d}".format(self.current_epoch))


def _n_gpus(gpus) -> int:
    """Lightning's `gpus` argument (int, "N", or a list of device ids) as a rank count"""
    if isinstance(gpus, (list, tuple)):
        return len(gpus)
    try:
        return max(1, int(gpus))
    except (TypeError, ValueError):
        return 1


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    conf_dir = os.environ.get("LASR_CONF_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf"))
    cfg = load_config(conf_dir, "conf", argv)
    seed_everything(0)
    tran_cfg, data_cfg, model_cfg = cfg.get("train"), cfg.get("data"), cfg.get("model")
    labels = data_cfg.get("labels")
    use_cer = False
    if isinstance(labels, str):                    # a vocabulary file => CER mode (train.py:216-219)
        labels = [c.strip() for c in open(labels, "r", encoding="utf-8").readlines()]
        use_cer = True
    dtype = model_cfg.get("dtype", "bf16" if tran_cfg.get("precision") == 16 else "f32")
    # `gpus: N` + `accelerator: ddp` from the plain command (conf/conf.yaml:21,30; Lightning's DDP plugin starts its per-GPU children
    # itself, train.py:233-252): decided HERE, before anything touches the GPU - this process becomes the supervisor of N fresh worker
    # processes (launch.py: RANK / LOCAL_RANK / MASTER_PORT in their environment, a fallback ladder around them) and returns their code
    from . import launch
    rc = launch.maybe_launch(_n_gpus(tran_cfg.get("gpus")), [sys.executable, "-m", "lightning_asr_amd.train"] + argv, hold_json=False)
    if rc is not None:
        if rc != 0:
            raise SystemExit(rc)
        return None
    # one process per GPU (torch.distributed.run sets LOCAL_RANK): select this rank's device BEFORE anything allocates, and
    # hand the indexed device to every component that owns GPU memory (the flat parameter buffers, the mel workspaces)
    local_rank = rank_device_index()
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    data_module = LibriDataModule(data_cfg.get("train_manifest"), data_cfg.get("val_manifest"), labels=labels,
                                  train_bs=tran_cfg.get("train_batch_size"), dev_bs=tran_cfg.get("dev_batch_size"),
                                  test_manifest=data_cfg.get("test_manifest"), num_worker=data_cfg.get("num_worker"),
                                  train_max_duration=data_cfg.get("train_max_duration"),
                                  dev_max_duration=data_cfg.get("dev_max_duration"),
                                  act_dtype=torch.float32 if dtype == "f32" else torch.bfloat16, device=device,
                                  bucket_by_length=bool(data_cfg.get("bucket_by_length", False)),
                                  bucket_batches=int(data_cfg.get("bucket_batches", 50)),
                                  train_crop=bool(data_cfg.get("train_crop", True)))
    model = LightingModule(learning_rate=tran_cfg.get("learning_rate"), weight_decay=tran_cfg.get("weight_decay"), labels=labels,
                           total_epoch=tran_cfg.get("total_epoch"), drop_rate=model_cfg.get("drop_rate"), mask=model_cfg.get("mask"),
                           use_cer=use_cer, variant=model_cfg.get("variant", "plain"), act=model_cfg.get("act", "relu"), dtype=dtype,
                           warmup_steps=tran_cfg.get("warmup_steps", 1000), device=device)
    trainer = Trainer(gpus=tran_cfg.get("gpus"), resume_from_checkpoint=tran_cfg.get("checkpoint"), accelerator=tran_cfg.get("accelerator"),
                      max_epochs=tran_cfg.get("total_epoch"), check_val_every_n_epoch=tran_cfg.get("check_val_every_n_epoch", 1),
                      num_nodes=tran_cfg.get("num_nodes"), default_root_dir=cfg.get("output_dir", "."),
                      max_steps=tran_cfg.get("max_steps"), log_every_n_steps=tran_cfg.get("log_every_n_steps", 50))
    trainer.fit(model, datamodule=data_module)
    trainer.test(model, test_dataloaders=data_module.test_dataloader())
    return trainer


if __name__ == "__main__":
    _tr = main()
    if _tr is not None:
        _tr.teardown()
