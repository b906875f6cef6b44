"""Manifest -> GPU at training speed: what the reference's 6 DataLoader workers + pin_memory + H2D do (data_module.py:16-56,
199-201,222-248; conf/conf.yaml:14), rebuilt around ONE int16 batch buffer.

* ``read_wav_batch``: liblasr's host threads (``lasr_wav_read_batch``) decode a batch of 16-bit wav files - and take the
  training-time random sub-sequence as a slice of the file - straight into a pinned ring slot (GIL released).
* ``BatchProducer``: a background thread that walks the sampler's index lists, fills ring slots, builds the padded targets and
  draws the SpecAugment rectangles (host ``random.Random``, as the reference).
* ``DeviceFeeder``: two H2D copies per batch (PCM as int16, one packed metadata block) on a copy stream into a small ring of
  device buffers, one batch ahead of the training step, ordered by events only.

The samples are scaled (1/32768), dithered, pre-emphasised and turned into log-mel features on the device
(``lasr_mel_fwd_src``)."""
from __future__ import annotations

import ctypes as C
import queue
import threading
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib

SR = 16000


def wav_info(path: str):
    """(n_frames, channels, sample_rate, bits) of a RIFF/WAVE file"""
    n, ch, sr, bits = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
    _lib.call("lasr_wav_info", path.encode(), C.byref(n), C.byref(ch), C.byref(sr), C.byref(bits))
    return n.value, ch.value, sr.value, bits.value


def read_wav_batch(paths: Sequence[str], out: torch.Tensor, lens_out: torch.Tensor, crop_u: Optional[np.ndarray] = None,
                   crop_weight: float = 0.98, n_threads: int = 8, expect_rate: int = 0, lead_in: bool = False) -> int:
    """Decode ``paths`` into ``out`` (1-D int16 host tensor, normally pinned) as rows of pitch ``ld`` (returned);
    ``lens_out`` (>= len(paths) int32 host tensor) receives the valid samples per row.  crop_u: (n, 2) float64 uniforms for the
    training-time sub-sequence (data_module.py:138-148), or None.  expect_rate > 0 refuses files of another sample rate; the
    default takes any rate as it is, like the reference (data_module.py:153 drops the rate torchaudio.load returns).
    lead_in: a crop that starts past the file's first sample brings the sample before it along as the row's first entry and
    flags its ``lens_out`` word with ``_lib.LEN_LEAD`` - the mel kernel then pre-emphasises the crop's first sample against it,
    which is what cropping AFTER dither + pre-emphasis gives (data_module.py:155-159)."""
    n = len(paths)
    if out.dtype != torch.int16 or out.is_cuda or not out.is_contiguous():
        raise TypeError("read_wav_batch writes a contiguous int16 host tensor")
    if lens_out.dtype != torch.int32 or lens_out.is_cuda or lens_out.numel() < n:
        raise TypeError("lens_out must be an int32 host tensor with one entry per file")
    arr = (C.c_char_p * n)(*[p.encode() for p in paths])
    ld = C.c_int64(0)
    cu = None
    if crop_u is not None:
        crop_u = np.ascontiguousarray(crop_u, dtype=np.float64)
        if crop_u.shape != (n, 2):
            raise ValueError("crop_u must have shape (n, 2)")
        cu = crop_u.ctypes.data_as(C.c_void_p)
    _lib.call("lasr_wav_read_batch", arr, n, cu, float(crop_weight), out.data_ptr(), out.numel(), C.byref(ld), lens_out.data_ptr(),
              int(expect_rate), int(n_threads), int(bool(lead_in)))
    return int(ld.value)


class HostBatch:
    """one batch on the host: PCM rows in ring slot ``slot`` + the packed metadata block"""
    __slots__ = ("slot", "B", "ld", "S", "lens", "sizes", "aug", "targets", "meta", "meta_words", "paths", "mask", "seconds", "index")


def _meta_layout(B: int, S: int, with_aug: bool):
    """int32 word offsets of [lens B][sizes B][aug 4B]...[targets B*S int64] inside the metadata block"""
    o_lens, o_sizes, o_aug = 0, B, 2 * B
    o_tg = 2 * B + (4 * B if with_aug else 0)
    o_tg += o_tg & 1                                  # int64 view needs an even word offset
    return o_lens, o_sizes, o_aug, o_tg, o_tg + 2 * B * S


class PinnedRing:
    """n_slots host buffers of `capacity` int16 samples + `meta_words` int32 words each, pinned when a GPU is present"""

    def __init__(self, n_slots: int, capacity: int, meta_words: int, pin: bool = True):
        self.pin = pin and torch.cuda.is_available()
        self.n_slots, self.capacity, self.meta_words = n_slots, int(capacity), int(meta_words)
        mk = (lambda n, dt: torch.empty(n, dtype=dt).pin_memory()) if self.pin else (lambda n, dt: torch.empty(n, dtype=dt))
        self._mk = mk
        self.pcm = [mk(self.capacity, torch.int16) for _ in range(n_slots)]
        self.meta = [mk(self.meta_words, torch.int32) for _ in range(n_slots)]
        self.free: "queue.Queue[int]" = queue.Queue()
        for i in range(n_slots):
            self.free.put(i)

    def grow(self, slot: int, capacity: int = 0, meta_words: int = 0) -> None:
        if capacity > self.pcm[slot].numel():
            self.pcm[slot] = self._mk(int(capacity), torch.int16)
        if meta_words > self.meta[slot].numel():
            self.meta[slot] = self._mk(int(meta_words), torch.int32)


class BatchProducer(threading.Thread):
    """Walks ``index_batches`` (lists of dataset indices, e.g. a DataLoader's batch_sampler) over a stock ``MyAudioDataset``
    and puts ``HostBatch`` objects on ``self.out`` (None at the end, an Exception on failure).  ``ring.free`` gets a slot back
    from the consumer once its H2D copies are done."""

    def __init__(self, dataset, index_batches: Iterable[List[int]], ring: PinnedRing, mask: bool, audio_parser, n_threads: int = 8,
                 crop_weight: float = 0.98, depth: int = 2, crop: Optional[bool] = None, feeder: Optional["DeviceFeeder"] = None):
        super().__init__(daemon=True)
        self.feeder = feeder         # given: this thread also issues the H2D copies and hands over DevBatch objects
        self.ds, self.batches, self.ring, self.mask, self.ap = dataset, index_batches, ring, mask, audio_parser
        self.n_threads, self.crop_weight = max(1, int(n_threads)), crop_weight
        self.crop = mask if crop is None else crop
        self.out: "queue.Queue" = queue.Queue(maxsize=depth)
        self._halt = threading.Event()

    def stop(self) -> None:
        self._halt.set()
        try:                                   # unblock a producer waiting for a slot
            self.ring.free.put_nowait(-1)
        except queue.Full:
            pass

    def _put(self, item) -> bool:
        while not self._halt.is_set():
            try:
                self.out.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def run(self) -> None:
        try:
            for k, idx in enumerate(self.batches):
                if self._halt.is_set():
                    return
                slot = self.ring.free.get()
                if slot < 0 or self._halt.is_set():
                    return
                item = self.make(list(idx), slot, k)
                if self.feeder is not None:
                    item = self.feeder.upload(item)
                    item.mask = self.mask
                if not self._put(item):
                    return
            self._put(None)
        except BaseException as e:  # noqa: BLE001 - handed to the consumer, which re-raises
            self._put(e)

    def _ids(self, i: int) -> np.ndarray:
        """token ids of manifest entry i (text -> ids once per entry, then cached)"""
        cache = self.ds.__dict__.setdefault("_lasr_ids_cache", {})
        a = cache.get(i)
        if a is None:
            c2i = self.ds.char2index
            a = cache[i] = np.fromiter((c2i[ch] for ch in self.ds.datasets[i]["text"]), dtype=np.int64)
        return a

    def make(self, idx: List[int], slot: int, index: int = 0) -> HostBatch:
        """one batch into ring slot `slot`; everything but the wav decode is a handful of numpy writes into the slot's metadata block
        (this thread shares the GIL with the thread that enqueues the training step)"""
        ds, ring = self.ds, self.ring
        B = len(idx)
        paths = [ds.datasets[i]["audio_filepath"] for i in idx]
        ids = [self._ids(i) for i in idx]
        S = max(1, max(a.size for a in ids))
        o_lens, o_sizes, o_aug, o_tg, words = _meta_layout(B, S, self.mask)
        ring.grow(slot, meta_words=words)
        meta = ring.meta[slot]
        mnp = meta.numpy()                                   # shares the (pinned) memory
        crop_u = np.random.uniform(0.0, 1.0, size=(B, 2)) if self.crop else None      # the two draws of sub_secquence per clip
        lens = meta[o_lens:o_lens + B]
        while True:
            try:
                ld = read_wav_batch(paths, ring.pcm[slot], lens, crop_u, self.crop_weight, self.n_threads, lead_in=self.crop)
                break
            except _lib.LasrError as e:
                if "do not fit the buffer" not in str(e):
                    raise
                ring.grow(slot, capacity=int(ring.pcm[slot].numel() * 1.5) + 8 * B)    # a file longer than its manifest duration
        lens_np = mnp[o_lens:o_lens + B] & (_lib.LEN_LEAD - 1)       # (the device words keep their lead-in flags; the host counts samples)
        mnp[o_sizes:o_sizes + B] = [a.size for a in ids]
        tg_np = mnp[o_tg:o_tg + 2 * B * S].view(np.int64).reshape(B, S)
        tg_np[:] = 0
        for i, a in enumerate(ids):
            tg_np[i, :a.size] = a
        aug = None
        if self.mask:       # spec_augment(27, 0.07) rectangles, drawn per clip in the reference's order (data_module.py:97-122,165)
            draw = self.ap.draw_spec_augment
            mnp[o_aug:o_aug + 4 * B] = np.asarray([draw(1 + (int(l) + 64) // 160) for l in lens_np], dtype=np.int32).reshape(-1)
            aug = meta[o_aug:o_aug + 4 * B].view(B, 4)
        hb = HostBatch()
        hb.slot, hb.B, hb.ld, hb.S, hb.lens, hb.aug = slot, B, ld, S, lens, aug
        hb.sizes = meta[o_sizes:o_sizes + B]
        hb.targets = meta[o_tg:o_tg + 2 * B * S].view(torch.int64).view(B, S)
        hb.meta, hb.meta_words, hb.paths, hb.mask, hb.index = meta, words, paths, self.mask, index
        hb.seconds = float(lens_np.sum()) / SR
        return hb


class DevBatch:
    """one batch resident in HBM (views of a device ring slot), valid for the compute stream once ``ready`` has been waited on"""
    __slots__ = ("pcm", "lens", "sizes", "aug", "targets", "paths", "B", "ld", "S", "seconds", "ready", "dslot", "index", "key", "mask", "waited",
                 "L", "pitch")


class DeviceFeeder:
    """Host batches -> device ring slots on a copy stream.  ``upload`` never blocks the host on the GPU: the copy stream waits
    for the compute-stream event of the step that last read the slot (``release``), copies PCM + metadata, records ``ready``;
    a helper thread returns the pinned slot to the producer once the copies are done."""

    def __init__(self, ring: PinnedRing, device, n_slots: int = 3):
        self.ring, self.device = ring, torch.device(device)
        self.n_slots = n_slots
        self.pcm = [torch.empty(ring.capacity, dtype=torch.int16, device=self.device) for _ in range(n_slots)]
        self.meta = [torch.empty(ring.meta_words, dtype=torch.int32, device=self.device) for _ in range(n_slots)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._retired: list = []               # outgrown device blocks, kept alive until close()
        # device slots are handed back explicitly: (slot, event recorded on the compute stream after the last kernel that reads it).
        # The uploader (producer thread) BLOCKS here when every slot is in flight - it may run several batches ahead of the step.
        self.free_dev: "queue.Queue" = queue.Queue()
        for k in range(n_slots):
            self.free_dev.put((k, None))
        self._done_q: "queue.Queue" = queue.Queue()
        self._recycler = threading.Thread(target=self._recycle, daemon=True)
        self._recycler.start()

    def _recycle(self) -> None:
        while True:
            item = self._done_q.get()
            if item is None:
                return
            ev, slot = item
            ev.synchronize()                   # H2D copies out of the pinned slot are done
            self.ring.free.put(slot)

    def close(self) -> None:
        self._done_q.put(None)
        self.free_dev.put((-1, None))          # unblock an uploader waiting for a device slot

    def upload(self, hb: HostBatch) -> DevBatch:
        """(called from the producer thread: HIP calls of a copy are ordinary stream operations, the compute thread only ever
        reads ``released`` / writes it in ``release``)"""
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        k, released = self.free_dev.get()
        if k < 0:
            raise RuntimeError("device feeder closed")
        # The batch's frame count is the reference's "pad to the longest utterance" (data_module.py:222-248): T = frames(longest).
        # The device rows get ONE pitch per frame-count class - 160 (T - 1) + 96 samples: room for the longest utterance of the class
        # plus its lead-in sample - and the kernels are told L = 160 (T - 1) + 95, the longest length with that T (lasr_wave_src.pitch,
        # round 5).  Until round 4 the pitch was the host reader's (longest row rounded up to 8 samples) and T followed it: one frame
        # of padding too many whenever the rounding (or the lead-in sample) crossed a frame boundary (~5 % of cropped batches), and
        # a different shape for nearly every batch - no hipGraph replay under the reference's random crop.
        longest = int((hb.lens & (_lib.LEN_LEAD - 1)).max()) if hb.B else 0
        frames = 1 + (longest + 64) // 160
        L_log, pitch = 160 * (frames - 1) + 95, 160 * (frames - 1) + 96
        if pitch < hb.ld:                      # (cannot happen: ld = the longest row rounded up to 8 <= 160 (T - 1) + 96)
            L_log = pitch = hb.ld
        n = hb.B * pitch
        cs = self.copy_stream
        if n > self.pcm[k].numel() or hb.meta_words > self.meta[k].numel():
            # A batch outgrew its slot.  The new block is allocated UNDER THE COPY STREAM: the caching allocator keeps one pool per
            # stream, so it cannot be a block the compute thread just freed with kernels still queued on the compute stream (the
            # H2D copy below is not ordered against that stream).  The outgrown block is parked while steps still in flight may be
            # reading it: `released` was recorded behind the last kernel that reads this slot, so the block is provably idle once
            # that event has completed - parked blocks are dropped then (ragged epochs no longer pin every outgrown
            # block of every slot until close()).
            idle = released is None or released.query()
            with torch.cuda.stream(cs):
                if n > self.pcm[k].numel():
                    if not idle:
                        self._retired.append((self.pcm[k], released))
                    self.pcm[k] = torch.empty(int(n * 1.25) + 64, dtype=torch.int16, device=self.device)
                if hb.meta_words > self.meta[k].numel():
                    if not idle:
                        self._retired.append((self.meta[k], released))
                    self.meta[k] = torch.empty(int(hb.meta_words * 1.5) + 64, dtype=torch.int32, device=self.device)
        if self._retired:
            self._retired = [(blk, ev_) for blk, ev_ in self._retired if not ev_.query()]
        if released is not None:
            cs.wait_event(released)             # the step that read this device slot has finished with it
        with torch.cuda.stream(cs):
            src = self.ring.pcm[hb.slot][:hb.B * hb.ld]
            if pitch == hb.ld:
                self.pcm[k][:n].copy_(src, non_blocking=True)
            else:                               # rows at the host reader's pitch -> rows at the class pitch (the tail of a row is never read)
                self.pcm[k][:n].view(hb.B, pitch)[:, :hb.ld].copy_(src.view(hb.B, hb.ld), non_blocking=True)
            self.meta[k][:hb.meta_words].copy_(hb.meta[:hb.meta_words], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        self._done_q.put((ev, hb.slot))
        B, S = hb.B, hb.S
        o_lens, o_sizes, o_aug, o_tg, _ = _meta_layout(B, S, hb.aug is not None)
        m = self.meta[k]
        db = DevBatch()
        db.pcm = self.pcm[k][:n].view(B, pitch)
        db.L, db.pitch = L_log, pitch
        db.lens, db.sizes = m[o_lens:o_lens + B], m[o_sizes:o_sizes + B]
        db.aug = m[o_aug:o_aug + 4 * B].view(B, 4) if hb.aug is not None else None
        db.targets = m[o_tg:o_tg + 2 * B * S].view(torch.int64).view(B, S)
        db.paths, db.B, db.ld, db.S, db.seconds, db.ready, db.dslot, db.index = hb.paths, B, hb.ld, S, hb.seconds, ev, k, hb.index
        db.key = (B, pitch, S, hb.aug is not None)
        db.waited = False
        return db

    def release(self, db: DevBatch) -> None:
        """call after the last kernel that reads ``db`` has been enqueued on the current stream"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.free_dev.put((db.dslot, ev))


def fast_ingest_ok(dataset, probe: int = 4) -> bool:
    """the native reader applies to the stock dataset over a manifest of 16-bit PCM RIFF/WAVE files (what the reference's
    scripts/get_libri.py writes).  The first, last and a few middle entries are probed with ``lasr_wav_info``: a manifest that
    points at flac, 24-bit or float wav files keeps the DataLoader route (whatever ``load_wav`` can decode) instead of aborting
    the fit on its first batch."""
    from .data_module import MyAudioDataset
    if not (type(dataset).__getitem__ is MyAudioDataset.__getitem__ and hasattr(dataset, "datasets") and hasattr(dataset, "char2index")):
        return False
    cached = dataset.__dict__.get("_lasr_fast_ingest_ok")
    if cached is not None:
        return cached
    n = len(dataset.datasets)
    ok = True
    for i in sorted({0, n - 1, *(n * k // probe for k in range(1, probe))}) if n else []:
        try:
            _frames, _ch, _rate, bits = wav_info(dataset.datasets[i]["audio_filepath"])
            ok = ok and bits == 16
        except _lib.LasrError:
            ok = False
        if not ok:
            import logging
            logging.getLogger(__name__).warning("native wav ingest off: %s is not a 16-bit PCM RIFF/WAVE file (falling back to the "
                                                "DataLoader route)", dataset.datasets[i]["audio_filepath"])
            break
    dataset.__dict__["_lasr_fast_ingest_ok"] = ok
    return ok
