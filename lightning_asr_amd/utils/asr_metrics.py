"""WER/CER metric with the reference's interface (utils/asr_metrics.py:26-228).

``WER.update`` runs entirely on the GPU: greedy CTC collapse (csrc/ctc.hip), then the Levenshtein distance of every
utterance against its reference (csrc/editdist.hip, token units for CER, word units for WER); ``scores`` / ``words`` stay
device scalars, so a training step's metric costs no D2H of token ids and no Python loop - only whoever reads the value
(``self.log`` -> ``.item()``) synchronises.  The string helpers (``ctc_decoder_predictions_tensor``, ``decode_reference``,
``word_error_rate``) keep the reference's host behaviour; their Levenshtein is the library's host routine
``lasr_edit_distance`` (editdistance is not a dependency)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import _lib, ops

__all__ = ["word_error_rate", "WER"]


def _edit_distance(a: Sequence, b: Sequence) -> int:
    vocab = {}
    ia = np.asarray([vocab.setdefault(x, len(vocab)) for x in a], dtype=np.int32)
    ib = np.asarray([vocab.setdefault(x, len(vocab)) for x in b], dtype=np.int32)
    lib = _lib.load()
    d = lib.lasr_edit_distance(ia.ctypes.data_as(C.c_void_p), len(ia), ib.ctypes.data_as(C.c_void_p), len(ib))
    if d < 0:
        raise _lib.LasrError("lasr_edit_distance failed")
    return int(d)


def word_error_rate(hypotheses: List[str], references: List[str], use_cer=False) -> float:
    """sum of edit distances / sum of reference lengths (utils/asr_metrics.py:26-59)."""
    if len(hypotheses) != len(references):
        raise ValueError("In word error rate calculation, hypotheses and reference lists must have the same number "
                         "of elements. But I got:{0} and {1} correspondingly".format(len(hypotheses), len(references)))
    scores = words = 0
    for h, r in zip(hypotheses, references):
        h_list, r_list = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(r_list)
        scores += _edit_distance(h_list, r_list)
    return 1.0 * scores / words if words != 0 else float("inf")


class WER:
    """Callable metric: ``wer(predictions (B,T) ids, targets, target_lengths, t_lengths)`` returns the batch
    WER (scores/words) like the torchmetrics object in the reference (compute_on_step=True), and keeps
    ``scores`` / ``words`` of the last update (dist_reduce_fx='sum' is the host trainer's job)."""

    def __init__(self, vocabulary, batch_dim_index=0, use_cer=False, ctc_decode=True, log_prediction=True,
                 dist_sync_on_step=False):
        self.batch_dim_index = batch_dim_index
        self.blank_id = len(vocabulary)
        self.labels_map = dict((i, vocabulary[i]) for i in range(len(vocabulary)))
        self.use_cer = use_cer
        self.ctc_decode = ctc_decode
        self.log_prediction = log_prediction
        self.scores = torch.tensor(0)
        self.words = torch.tensor(0)
        # device path: token units need nothing; word units need every label to be one non-space character or ' '
        labels = [vocabulary[i] for i in range(len(vocabulary))]
        self.space_id = -1
        self.device_ok = True
        if not use_cer:
            self.device_ok = all(len(s) == 1 and (s == " " or not s.isspace()) for s in labels)
            self.space_id = labels.index(" ") if " " in labels else len(labels) + 1      # no space label: one word per utterance

    def ctc_decoder_predictions_tensor(self, predictions: torch.Tensor, predictions_len: Optional[torch.Tensor] = None) -> List[str]:
        """ids (B,T) [+ lengths] -> greedy-collapsed strings (utils/asr_metrics.py:138-171)."""
        if not predictions.is_cuda:
            raise _lib.LasrError("the greedy CTC collapse runs on the GPU; move the predictions there")
        ids = predictions.to(torch.int32).contiguous()
        lens = None if predictions_len is None else predictions_len.to(ids.device, torch.int32).contiguous()
        tokens, n = ops.greedy_decode(ids, lens, self.blank_id)
        tokens, n = tokens.cpu().numpy(), n.cpu().numpy()
        return ["".join(self.labels_map[int(c)] for c in tokens[b, :n[b]]) for b in range(tokens.shape[0])]

    def decode_reference(self, targets: torch.Tensor, target_lengths: torch.Tensor) -> List[str]:
        t = targets.long().cpu().numpy()
        l = target_lengths.long().cpu().numpy()
        return ["".join(self.labels_map[int(c)] for c in t[b, :l[b]]) for b in range(t.shape[0])]

    def update(self, predictions, targets, target_lengths, t_lengths=None):
        if not self.ctc_decode:
            raise NotImplementedError("Implement me if you need non-CTC decode on predictions")
        if self.device_ok and predictions.is_cuda and predictions.shape[1] <= 2048 and targets.shape[1] <= 2048:
            dev = predictions.device
            ids = predictions.to(torch.int32).contiguous()
            lens = None if t_lengths is None else t_lengths.to(dev, torch.int32).contiguous()
            tokens, n = ops.greedy_decode(ids, lens, self.blank_id)
            dist, units = ops.edit_distance_batch(tokens, n, targets.to(dev, torch.int64).contiguous(),
                                                  target_lengths.to(dev, torch.int32).contiguous(), self.space_id)
            self.scores, self.words = dist.sum(), units.sum()          # device scalars: nothing is copied to the host here
            return
        references = self.decode_reference(targets, target_lengths)
        hypotheses = self.ctc_decoder_predictions_tensor(predictions, t_lengths)
        words = scores = 0
        for h, r in zip(hypotheses, references):
            h_list, r_list = (list(h), list(r)) if self.use_cer else (h.split(), r.split())
            words += len(r_list)
            scores += _edit_distance(h_list, r_list)
        self.scores = torch.tensor(scores)
        self.words = torch.tensor(words)

    def compute(self):
        return self.scores.detach().float() / self.words.detach().float()

    def __call__(self, predictions, targets, target_lengths, t_lengths=None):
        self.update(predictions, targets, target_lengths, t_lengths)
        return self.compute()
