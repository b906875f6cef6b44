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
        # device path.  CER compares the CHARACTERS of the joined strings (utils/asr_metrics.py:215-216 list(h) / list(r)): token
        # ids are the same units only when every label is exactly one character and no two ids share it (a '<unk>' entry, or a
        # vocabulary-file line that strips to '' - train.py:217 - makes the host path the only faithful one).  WER additionally
        # needs the word separator to be a single label.
        labels = [vocabulary[i] for i in range(len(vocabulary))]
        self.space_id = -1
        self.device_ok = all(len(s) == 1 for s in labels) and len(set(labels)) == len(labels)
        if not use_cer:
            self.device_ok = self.device_ok and all(s == " " or not s.isspace() for s in labels)
            self.space_id = labels.index(" ") if " " in labels else len(labels) + 1      # no space label: one word per utterance
        self.world_reduce = None     # callable(scores, words) -> (scores, words) summed over the data-parallel ranks (set by the trainer)
        self._total = None           # running [sum scores, sum words] since reset()

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

    def device_path(self, predictions, targets) -> bool:
        return self.device_ok and predictions.is_cuda and predictions.shape[1] <= 2048 and targets.shape[1] <= 2048

    def device_distances(self, predictions, targets, target_lengths, t_lengths=None):
        """greedy collapse + Levenshtein on the device: (dist (B) i32, ref_units (B) i32), nothing copied to the host"""
        dev = predictions.device
        ids = predictions.to(torch.int32).contiguous()
        lens = None if t_lengths is None else t_lengths.to(dev, torch.int32).contiguous()
        tokens, n = ops.greedy_decode(ids, lens, self.blank_id)
        return ops.edit_distance_batch(tokens, n, targets.to(dev, torch.int64).contiguous(),
                                       target_lengths.to(dev, torch.int32).contiguous(), self.space_id)

    def update(self, predictions, targets, target_lengths, t_lengths=None):
        if not self.ctc_decode:
            raise NotImplementedError("Implement me if you need non-CTC decode on predictions")
        if self.device_path(predictions, targets):
            dist, units = self.device_distances(predictions, targets, target_lengths, t_lengths)
            self.scores, self.words = dist.sum(), units.sum()          # device scalars: nothing is copied to the host here
            self._accumulate()
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
        self._accumulate()

    def compute(self):
        """the last update's batch value (compute_on_step, dist_sync_on_step=False: not synchronised across ranks)"""
        return self.scores.detach().float() / self.words.detach().float()

    # ---- epoch-level state: what torchmetrics' compute() returns at the end of an epoch -------------------------------------
    def _accumulate(self) -> None:
        v = torch.stack([self.scores.detach().double().reshape(()), self.words.detach().double().reshape(())])
        self._total = v if self._total is None else self._total.to(v.device) + v

    def reset(self) -> None:
        self._total = None

    def compute_total(self):
        """sum(scores) / sum(words) over every update since reset() and - dist_reduce_fx='sum', utils/asr_metrics.py:114-115 -
        over the data-parallel ranks (``world_reduce``, installed by the trainer)"""
        if self._total is None:
            return torch.tensor(float("nan"))
        scores, words = self._total[0], self._total[1]
        if self.world_reduce is not None:
            scores, words = self.world_reduce(scores, words)
        return scores.float() / words.float()

    def __call__(self, predictions, targets, target_lengths, t_lengths=None):
        self.update(predictions, targets, target_lengths, t_lengths)
        return self.compute()
