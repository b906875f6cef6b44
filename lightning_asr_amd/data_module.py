"""Data contract of the reference's data_module.py with the feature front-end moved to the GPU.

Same public names and arguments: ``MyAudioDataset``, ``AudioParser``, ``LibriDataModule`` (manifest =
JSON lines ``{"audio_filepath","duration","text"}``, scripts/get_libri.py:135).  What differs, by design:
DataLoader workers only decode PCM; dither, pre-emphasis, STFT, mel, dB, SpecAugment zeros, per-utterance
normalisation and pad-to-longest collate run as ONE batched HIP call (csrc/mel.hip) in
``on_after_batch_transfer``, which hands ``training_step`` the reference's 5-tuple
``(inputs (B,1,64,Tmax), targets, input_percentages, target_sizes, paths)`` (data_module.py:248)."""
from __future__ import annotations

import json
import logging
import os
import random
import wave as _wave
from typing import List, Optional, Sequence, Union

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import _lib, ops
from .lightning_compat import LightningDataModule


def load_wav(path_or_file) -> torch.Tensor:
    """16-bit PCM wav -> (1, L) f32 in [-1, 1) (what torchaudio.load returns, data_module.py:153)."""
    try:
        with _wave.open(path_or_file, "rb") as w:
            if w.getsampwidth() != 2:
                raise ValueError("only 16-bit PCM wav is supported")
            n, ch = w.getnframes(), w.getnchannels()
            pcm = np.frombuffer(w.readframes(n), dtype="<i2").reshape(-1, ch).T
        return torch.from_numpy(pcm[:1].astype(np.float32) / 32768.0)
    except (ValueError, _wave.Error, EOFError) as e:
        # flac / 24-bit / float wav (LibriSpeech's native format): the site's `soundfile`, if it has one (the reference's own
        # requirements list it next to torchaudio; not installed in this image).  No torchaudio anywhere in this package.
        if hasattr(path_or_file, "seek"):
            path_or_file.seek(0)                         # a file-like input was partly consumed by the wave reader above
        try:
            import soundfile as sf
        except ImportError:
            raise ValueError("%s: %s (not a 16-bit PCM wav, and no `soundfile` module to decode other formats)" % (path_or_file, e)) from e
        try:
            data, _sr = sf.read(path_or_file, dtype="float32", always_2d=True)
        except Exception as e2:                          # soundfile raises RuntimeError / LibsndfileError for what it cannot read
            raise ValueError("%s: neither the wave reader (%s) nor soundfile (%s) can decode it" % (path_or_file, e, e2)) from e2
        return torch.from_numpy(np.ascontiguousarray(data.T[:1]))


class AudioParser:
    """``parse_audio(path, mask) -> (1, 64, T)`` (data_module.py:150-174), computed on the GPU."""

    def __init__(self, win_len=0.02, sr=16000, device="cuda"):
        if int(win_len * sr) != 320 or sr != 16000:
            raise NotImplementedError("the HIP front-end is built for win_len=0.02, sr=16000 (data_module.py:59)")
        self.win_len, self.sr = win_len, sr
        self.rand = random.Random()
        self.device = torch.device(device)

    # -- the two random pieces of the training-time chain, drawn on the host like the reference ----
    def sub_secquence(self, x: torch.Tensor, weight: float = 0.1) -> torch.Tensor:
        """Bug-compatible: the slice END is target_length, not location+target_length (:138-148)."""
        length = x.shape[1]
        target_length = int(length * np.random.uniform(weight, 1))
        location = int(np.random.uniform(0, length - target_length))
        return x[:, location:target_length]

    def crop_raw(self, x: torch.Tensor, weight: float = 0.98):
        """The same two draws applied to the RAW waveform, for the device chain: the reference dithers and pre-emphasises the
        whole clip and slices afterwards (:155-159), so the crop's first sample is ``y[loc] - 0.97 y[loc-1]``.  Returns
        (row, lead): ``row`` = the slice with the sample before it in front when ``loc > 0`` (lead = 1, ``_lib.LEN_LEAD`` in the
        length word handed to the mel kernel), which then produces exactly the reference's values."""
        length = x.shape[1]
        target_length = int(length * np.random.uniform(weight, 1))
        location = int(np.random.uniform(0, length - target_length))
        lead = 1 if (location > 0 and target_length > location) else 0
        return x[:, location - lead:target_length], lead

    def draw_spec_augment(self, n_time: int, freq_mask: Union[int, float] = 27, time_mask: Union[int, float] = 0.07):
        """(rect_x, w_x, rect_y, w_y) with the draw order of spec_augment (:97-122)."""
        if isinstance(freq_mask, float):
            freq_mask = int(64 * freq_mask)
        if isinstance(time_mask, float):
            time_mask = int(n_time * time_mask)
        w_x = int(self.rand.uniform(0, freq_mask))
        w_y = int(self.rand.uniform(0, time_mask))
        rect_x = int(self.rand.uniform(0, 64 - w_x))
        rect_y = int(self.rand.uniform(0, n_time - w_y))
        return rect_x, w_x, rect_y, w_y

    def spec_augment(self, x: torch.Tensor, freq_mask: Union[int, float] = 27, time_mask: Union[int, float] = 100) -> torch.Tensor:
        """The reference's public method (data_module.py:97-122), same signature and draw order: x (1, 64, T) -> a copy with
        `w_x` mel rows from `rect_x` and `w_y` frames from `rect_y` set to zero.  Inside `parse_audio` / the training loop the same
        rectangle travels into the mel kernel (`draw_spec_augment` + `lasr_mel_fwd`'s aug); this stand-alone form is one
        `lasr_spec_augment` launch on an existing feature tensor."""
        if x.dim() != 3:
            raise ValueError("spec_augment expects (1, F, T) features, got %s" % (tuple(x.shape),))
        if isinstance(freq_mask, float):
            freq_mask = int(x.shape[1] * freq_mask)
        if isinstance(time_mask, float):
            time_mask = int(x.shape[2] * time_mask)
        w_x = int(self.rand.uniform(0, freq_mask))
        w_y = int(self.rand.uniform(0, time_mask))
        rect_x = int(self.rand.uniform(0, x.shape[1] - w_x))
        rect_y = int(self.rand.uniform(0, x.shape[2] - w_y))
        xin = x.to(self.device, torch.float32).contiguous()
        out = torch.empty_like(xin)
        aug = torch.tensor([[rect_x, w_x, rect_y, w_y]] * xin.shape[0], dtype=torch.int32, device=self.device)
        ops.call("lasr_spec_augment", xin.data_ptr(), out.data_ptr(), aug.data_ptr(), xin.shape[0], xin.shape[1], xin.shape[2],
                 torch.cuda.current_stream(self.device).cuda_stream)
        return out.to(x.device)

    def parse_audio(self, audio_path, mask=False) -> torch.Tensor:
        if isinstance(audio_path, str) and not os.path.exists(path=audio_path):
            raise Exception("音频路径不存在 " + audio_path)
        y = load_wav(audio_path)
        lead = 0
        if mask:
            y, lead = self.crop_raw(y, weight=0.98)
        return self.features([y[0]], mask, leads=[lead])[0]

    # ---- the batched device front-end ---------------------------------------------------------------------------------------
    def device_dither(self):
        """``y += 1e-5 * randn_like(y)`` (:155) is drawn inside the mel kernel (Philox keyed by the process seed)"""
        if getattr(self, "_dither", None) is None:
            self._dither = ops.DeviceDither(int(torch.initial_seed()), self.device)
        return self._dither

    def draw_aug_batch(self, sample_lens) -> torch.Tensor:
        """(B, 4) int32 SpecAugment rectangles for utterances of `sample_lens` samples (host draws, as the reference)"""
        return torch.tensor([self.draw_spec_augment(1 + (int(l) + 64) // 160) for l in sample_lens], dtype=torch.int32)

    def features_device(self, wave: torch.Tensor, lens: Optional[torch.Tensor], aug: Optional[torch.Tensor] = None, dither: bool = True,
                        logical_len: Optional[int] = None):
        """wave (B, L) f32 or int16 PCM ALREADY in HBM -> (inputs (B,1,64,Tmax) f32 with its channels-last twin attached,
        input_percentages (B,)); frames past each utterance are zero (collate, :222-248).  logical_len: the longest utterance when
        the rows are wider than that - Tmax is ITS frame count (the reference pads to the longest feature matrix)."""
        bft, btf, frames, pct = ops.mel(wave, lens, self.device_dither() if dither else None, aug, True, self._act_dtype(),
                                        logical_len=logical_len)
        inputs = bft.unsqueeze(1)
        inputs._lasr_btf = btf                                           # channels-last twin for the model
        return inputs, pct

    def _staging(self, n: int) -> torch.Tensor:
        """two alternating pinned f32 staging buffers, reused across batches (a buffer is rewritten only after its last H2D copy
        has completed)"""
        st = getattr(self, "_stage", None)
        if st is None:
            st = self._stage = {"buf": [None, None], "ev": [None, None], "k": 0}
        k = st["k"] = st["k"] ^ 1
        if st["ev"][k] is not None:
            st["ev"][k].synchronize()
        if st["buf"][k] is None or st["buf"][k].numel() < n:
            st["buf"][k] = torch.empty(int(n * 1.25) + 1024, dtype=torch.float32).pin_memory()
        return st["buf"][k][:n]

    def features(self, waves: Sequence[torch.Tensor], mask: bool, dither: bool = True, leads: Optional[Sequence[int]] = None):
        """list of (L_i,) f32 host waves -> (inputs, input_percentages) on the GPU: padded into a reused pinned buffer, ONE H2D
        copy for the batch, then ``features_device``.  leads[i] = 1: waves[i] starts with a lead-in sample (``crop_raw``)."""
        B = len(waves)
        L = max(int(w.numel()) for w in waves)
        host = self._staging(B * L).view(B, L)
        host.zero_()
        lens = torch.empty(B, dtype=torch.int32)
        n_sig = torch.empty(B, dtype=torch.int32)
        for i, w in enumerate(waves):
            host[i, :w.numel()] = w.cpu() if w.is_cuda else w
            ld = int(leads[i]) if leads is not None else 0
            n_sig[i] = w.numel() - ld
            lens[i] = (w.numel() - ld) | (_lib.LEN_LEAD if ld else 0)
        dev = self.device
        wave = host.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stage["ev"][self._stage["k"]] = ev
        aug = self.draw_aug_batch(n_sig).to(dev) if mask else None
        # Tmax = the frames of the longest UTTERANCE (a row's lead-in sample is not part of it: it emits no frame of its own)
        return self.features_device(wave, lens.to(dev), aug, dither, logical_len=max(int(n_sig.max()), 1))

    def _act_dtype(self):
        return getattr(self, "act_dtype", torch.float32)


class MyAudioDataset(Dataset):
    def __init__(self, manifest_path: list, labels, max_duration=16.7, mask=False, win_len=0.02, sr=16000):
        self.datasets = []
        self.labels = labels
        self.mask = mask
        for item in manifest_path:
            total_count, total_duration = 0, 0.0
            with open(item, encoding="utf-8") as f:
                for line in f.readlines():
                    if not line.strip():
                        continue
                    data = json.loads(line)
                    if data["duration"] > max_duration:
                        total_count += 1
                        total_duration += data["duration"]
                        continue
                    self.datasets.append(data)
            logging.info("过滤音频条数:{:d}条".format(total_count))
            logging.info("过滤音频时长:{:.2f}分钟".format(total_duration / 60))
        self.index2char = dict((i, labels[i]) for i in range(len(labels)))
        self.char2index = dict((labels[i], i) for i in range(len(labels)))

    def __getitem__(self, index):
        """-> (wave (L,) f32 on the host, token ids, path): PCM decode only, features are batched on the GPU."""
        data = self.datasets[index]
        text2id = [self.char2index[char] for char in data["text"]]
        return load_wav(data["audio_filepath"])[0], text2id, data["audio_filepath"]

    def id2txt(self, id_list):
        for id in id_list:
            if id >= len(self.index2char):
                raise Exception("index out of the lengths请检查id的大小范围")
        return "".join(self.index2char[id] for id in id_list)

    def __len__(self):
        return len(self.datasets)


class BucketBatchSampler(torch.utils.data.Sampler):
    """Length-bucketed batches (BASELINE config 5; not in the reference, which pads to the batch's
    longest clip, data_module.py:225-230): utterances are sorted by duration inside shuffled mega-chunks
    of ``bucket_batches`` batches, so padding inside a batch stays small; batch order is reshuffled
    every epoch.  Data-parallel ranks take disjoint batches (rank::world)."""

    def __init__(self, durations, batch_size: int, bucket_batches: int = 50, shuffle: bool = True, drop_last: bool = True,
                 seed: int = 0, rank: int = 0, world: int = 1):
        self.durations = list(durations)
        self.batch_size, self.bucket_batches = batch_size, bucket_batches
        self.shuffle, self.drop_last, self.seed, self.rank, self.world = shuffle, drop_last, seed, rank, world
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _batches(self):
        n = len(self.durations)
        rng = random.Random(self.seed + self.epoch)
        idx = list(range(n))
        if self.shuffle:
            rng.shuffle(idx)
        chunk = self.batch_size * self.bucket_batches
        batches = []
        for i in range(0, n, chunk):
            part = sorted(idx[i:i + chunk], key=lambda j: self.durations[j])
            for k in range(0, len(part), self.batch_size):
                b = part[k:k + self.batch_size]
                if len(b) == self.batch_size or not self.drop_last:
                    batches.append(b)
        if self.shuffle:
            rng.shuffle(batches)
        usable = len(batches) - len(batches) % self.world if self.world > 1 else len(batches)
        return batches[:usable][self.rank::self.world]

    def __iter__(self):
        return iter(self._batches())

    def __len__(self):
        return len(self._batches())


def _rebuild_wave_batch(items, leads):
    wb = WaveBatch(items)
    wb.leads = leads
    return wb


class WaveBatch(tuple):
    """(waves list, targets (B,Smax) int64, target_sizes (B) int32, paths, mask flag) from the workers.  ``leads`` (list of 0/1,
    or None): waves[i] starts with a lead-in sample (``AudioParser.crop_raw``); it survives the DataLoader's pickling."""
    leads = None

    def __reduce__(self):
        return (_rebuild_wave_batch, (tuple(self), self.leads))


class LibriDataModule(LightningDataModule):
    def __init__(self, train_manifest, dev_manifest, test_manifest, labels: list, train_bs=16, dev_bs=16, num_worker=0,
                 train_max_duration=16.7, dev_max_duration=40, device="cuda", act_dtype=torch.float32,
                 bucket_by_length: bool = False, bucket_batches: int = 50, train_crop: bool = True):
        super().__init__()
        as_list = lambda m: list(m) if isinstance(m, (list, tuple)) else [m]  # noqa: E731
        self.train_manifest, self.dev_manifest, self.test_manifest = as_list(train_manifest), as_list(dev_manifest), as_list(test_manifest)
        self.train_bs, self.dev_bs = train_bs, dev_bs
        self.labels = labels
        self.num_worker = num_worker
        self.train_max_duration, self.dev_max_duration = train_max_duration, dev_max_duration
        self.audio_parser = AudioParser(device=device)
        self.audio_parser.act_dtype = act_dtype
        self.bucket_by_length = bool(bucket_by_length)       # BASELINE cfg5: length-bucketed batches (conf key data.bucket_by_length)
        self.bucket_batches = int(bucket_batches)            # batches per sorted mega-chunk (conf key data.bucket_batches)
        self.train_crop = bool(train_crop)                   # the reference's random sub-sequence of every training clip (data_module.py:158-159); conf key data.train_crop

    def setup(self, stage=None):
        self.train_datasets = MyAudioDataset(self.train_manifest, self.labels, mask=True, max_duration=self.train_max_duration)
        self.dev_datasets = MyAudioDataset(self.dev_manifest, self.labels, max_duration=self.dev_max_duration)
        self.test_datasets = MyAudioDataset(self.test_manifest, self.labels, max_duration=self.dev_max_duration)

    def _loader(self, ds, bs, train, distributed=None):
        if train and getattr(self, "bucket_by_length", False):
            world, rank = distributed if distributed is not None else (1, 0)
            bs_ = BucketBatchSampler([d["duration"] for d in ds.datasets], bs, bucket_batches=self.bucket_batches, rank=rank, world=world)
            return DataLoader(ds, batch_sampler=bs_, num_workers=self.num_worker, collate_fn=self._collate_train)
        sampler = None
        if distributed is not None:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(ds, num_replicas=distributed[0], rank=distributed[1], shuffle=train, drop_last=train)
        collate = self._collate_train if train else self._collate_eval
        return DataLoader(ds, batch_size=bs, num_workers=self.num_worker, pin_memory=False, collate_fn=collate, drop_last=train,
                          shuffle=train and sampler is None, sampler=sampler)

    def train_dataloader(self, distributed=None):
        return self._loader(self.train_datasets, self.train_bs, True, distributed)

    def val_dataloader(self):
        return self._loader(self.dev_datasets, self.dev_bs, False)

    def test_dataloader(self):
        return self._loader(self.test_datasets, self.dev_bs, False)

    def get_train_step(self):
        return len(self.train_dataloader())

    # ---- host half of the collate: ragged waves + padded targets (data_module.py:231-247) ---------
    def _collate_wave(self, batch, mask: bool) -> WaveBatch:
        waves = [b[0] for b in batch]
        leads = None
        if mask and getattr(self, "train_crop", True):   # training-time random sub-sequence (data_module.py:158-159)
            cr = [self.audio_parser.crop_raw(w.unsqueeze(0), weight=0.98) for w in waves]
            waves, leads = [c[0][0] for c in cr], [c[1] for c in cr]
        max_trans = max(len(b[1]) for b in batch)
        targets = torch.zeros(len(batch), max_trans, dtype=torch.int64)
        target_sizes = torch.zeros(len(batch), dtype=torch.int32)
        for i, b in enumerate(batch):
            target_sizes[i] = len(b[1])
            targets[i, :len(b[1])] = torch.tensor(b[1], dtype=torch.int64)
        wb = WaveBatch((waves, targets, target_sizes, [b[2] for b in batch], mask))
        wb.leads = leads
        return wb

    def _collate_train(self, batch):
        return self._collate_wave(batch, True)

    def _collate_eval(self, batch):
        return self._collate_wave(batch, False)

    _collate_fn = _collate_eval

    # ---- device half: ONE batched HIP mel call -> the reference's 5-tuple -------------------------
    def on_after_batch_transfer(self, batch, dataloader_idx=0):
        if not isinstance(batch, WaveBatch):
            return batch
        waves, targets, target_sizes, paths, mask = batch
        inputs, pct = self.audio_parser.features(waves, mask, leads=batch.leads)
        dev = inputs.device
        return inputs, targets.to(dev), pct, target_sizes.to(dev), paths
