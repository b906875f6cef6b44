"""Inference surface of the reference (predict.py:21-74): checkpoint -> ``AsrTranslator.translate`` and
manifest evaluation, on the HIP path (mel front-end, eval-mode model forward, greedy CTC decode).

The checkpoint is the PL-style dict the reference's ``ModelCheckpoint`` writes and ``Trainer`` here
writes too: ``state_dict`` with the reference's key names (``encoder.encoder.block1.seq.0...``) and
``hyper_parameters`` (train.py:194 ``save_hyperparameters``), so reference-trained weights load as-is.
The SSL / LM-beam-search translator (predict.py:76-) belongs to the wav2vec2 branch, out of scope."""
from __future__ import annotations

import time
from typing import List, Optional

import torch

from .data_module import AudioParser, LibriDataModule
from .lightning_compat import Trainer
from .train import LightingModule
from .utils.asr_metrics import WER

EN_LABELS = [" ", "'"] + [chr(ord("a") + i) for i in range(26)]


class AsrTranslator:
    def __init__(self, model_path: str, map_location: str = "cuda", lang: str = "en", labels: Optional[List[str]] = None,
                 verbose: bool = False):
        """model_path: a ``.ckpt`` written by the reference or by ``Trainer``; map_location must name a GPU
        ("cuda" / "cuda:0"): there is no CPU path.  ``labels`` overrides the language's vocabulary."""
        if labels is not None:
            self.labels = list(labels)
        elif lang == "en":
            self.labels = list(EN_LABELS)
        else:
            raise Exception("其他语言未实现")                      # predict.py:36
        if not str(map_location).startswith("cuda"):
            raise ValueError("AsrTranslator runs on the GPU only (map_location=%r)" % (map_location,))
        self.model_path = model_path
        self.map_location = map_location
        self.verbose = verbose
        self.model = LightingModule.load_from_checkpoint(model_path, map_location=map_location, device=str(map_location))
        self.audio_parser = AudioParser(device=str(map_location))
        self.audio_parser.act_dtype = self.model.encoder.native.act_dtype
        self.device = torch.device(map_location)
        self.wer = WER(vocabulary=self.labels)
        self.model.eval()

    @torch.no_grad()
    def translate(self, audio_path) -> str:
        """One local audio file (path or file object) -> text (predict.py:44-63): no dither-free shortcut, the same
        feature chain as training without augmentation, eval-mode BN, argmax, CTC collapse."""
        t0 = time.time()
        inputs = self.audio_parser.parse_audio(audio_path, mask=False)
        pct = torch.ones(inputs.shape[0], dtype=torch.float32, device=self.device)   # torch.FloatTensor([1.])  (:55)
        t1 = time.time()
        out = self.model._encode(inputs, pct)
        ids = torch.argmax(out, dim=-1, keepdim=False)
        t2 = time.time()
        text = self.wer.ctc_decoder_predictions_tensor(ids)[0]
        if self.verbose:
            print("加载音频用时: %.4f  模型计算用时: %.4f  解码用时: %.4f" % (t1 - t0, t2 - t1, time.time() - t2))
        return text

    def evalute_manifest(self, test_manifest: str, batch_size: int = 32, num_workers: int = 0):
        """WER over a manifest (predict.py:65-74; the reference's spelling kept)."""
        data_module = LibriDataModule(train_manifest=test_manifest, dev_manifest=test_manifest, test_manifest=test_manifest,
                                      dev_bs=batch_size, num_worker=num_workers, labels=self.labels,
                                      device=str(self.model.encoder.native.device), act_dtype=self.model.encoder.native.act_dtype)
        trainer = Trainer(gpus=1, device=str(self.model.encoder.native.device))
        return trainer.test(self.model, datamodule=data_module)
