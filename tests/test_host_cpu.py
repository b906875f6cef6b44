"""CPU tier: host logic of the product (no compute calls: there is no GPU here) and the C-ABI surface."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "lasr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lasr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lightning_asr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", ROOT, "-j", "8"], check=True)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), "liblasr.so does not export %s declared in include/lasr.h" % s
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    assert _lib.load().lasr_version() >= 100
    assert _lib.load().lasr_last_error() is not None


def test_host_routines_without_gpu():
    from lightning_asr_amd import _lib
    from lightning_asr_amd.utils.asr_metrics import word_error_rate
    lib = _lib.load()
    assert lib.lasr_mel_num_frames(160000) == 1001
    assert lib.lasr_mel_num_frames(16000) == 101
    a = np.array([1, 2, 3, 4], dtype=np.int32)
    b = np.array([1, 3, 4, 5, 6], dtype=np.int32)
    d = lib.lasr_edit_distance(a.ctypes.data_as(ctypes.c_void_p), 4, b.ctypes.data_as(ctypes.c_void_p), 5)
    assert d == 3
    assert word_error_rate(["a b c", "x"], ["a c", "x y"]) == pytest.approx(0.5)
    assert word_error_rate(["kitten"], ["sitting"], use_cer=True) == pytest.approx(3 / 7)
    with pytest.raises(ValueError):
        word_error_rate(["a"], ["a", "b"])


def test_error_convention_and_no_cpu_fallback():
    from lightning_asr_amd import _lib, ops
    lib = _lib.load()
    rc = lib.lasr_gemm(None, None, None, 0, 0, 4, 4, 4, 0, 0, None, None, None, 0, None, 1, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.lasr_last_error()      # LASR_E_ARG, message set, nothing launched
    with pytest.raises(_lib.LasrError):
        ops.dwconv(torch.zeros(1, 8, 4), torch.zeros(4, 3))             # CPU tensors: refused, never emulated
    from lightning_asr_amd.engine import NativeModel
    with pytest.raises(_lib.LasrError):
        NativeModel("plain", 28, device="cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lightning_asr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert "oracle" not in text.replace("oracle/", "").lower() or f == "__init__.py" or "import oracle" not in text, f
                assert "from oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)


def test_config_loader_defaults_interpolation_overrides(tmp_path):
    from lightning_asr_amd.config import load_config
    cfg = load_config(os.path.join(ROOT, "conf"), "conf", ["train.learning_rate=5e-3", "data.train_manifest=[a.json,b.json]",
                                                            "model.variant=context_se", "+extra.flag=true"])
    assert cfg.get("train").get("learning_rate") == 5e-3
    assert cfg.get("data").get("train_manifest") == ["a.json", "b.json"]
    assert cfg.get("model").get("mask") is True and cfg.model.variant == "context_se" and cfg.extra.flag is True
    assert cfg.log.level == "INFO"                                        # defaults: - log: hypra_logger
    name = cfg.loggers.tensorboard.experiment_fixed_name
    assert "asr13x1-lr0.005-wc0.001-bs32" in name and "mask_True" in name   # ${a.b} interpolation sees the override
    assert cfg.output_dir == "outputs/asr13x1"
    assert cfg.get("train").get("checkpoint") is None and cfg.get("nothing", 7) == 7


def test_schedule_syncs_optimizer_groups():
    from lightning_asr_amd.schedule import CosineAnnealingWarmupRestarts
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-2)
    s = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=20, cycle_mult=2, max_lr=1e-2, min_lr=1e-4, warmup_steps=5, gamma=0.5)
    assert opt.param_groups[0]["lr"] == 1e-4                              # init_lr(): starts at min_lr
    lrs = []
    for _ in range(70):
        lrs.append(s.step())
        assert opt.param_groups[0]["lr"] == lrs[-1]
    assert max(lrs[:20]) == pytest.approx(1e-2) and max(lrs[20:]) == pytest.approx(5e-3)   # restart with gamma
    sd = s.state_dict()
    s2 = CosineAnnealingWarmupRestarts(None, first_cycle_steps=20, cycle_mult=2, max_lr=1e-2, min_lr=1e-4, warmup_steps=5, gamma=0.5)
    s2.load_state_dict(sd)
    assert s2.step() == s.step()


def test_dataset_manifest_and_wave_collate(tmp_path):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_data.py"), "--out", str(tmp_path), "--n-train", "3",
                    "--n-dev", "2", "--seconds", "1.0", "--ragged"], check=True)
    from lightning_asr_amd.data_module import MyAudioDataset, load_wav
    labels = [c.strip() for c in open(os.path.join(ROOT, "data", "labels.txt")).readlines()]
    assert len(labels) == 27 and labels[0] == "'"
    ds = MyAudioDataset([str(tmp_path / "train.json")], labels, max_duration=16.7, mask=True)
    assert len(ds) == 3
    wave, ids, path = ds[0]
    assert wave.dtype == torch.float32 and wave.dim() == 1 and wave.abs().max() <= 1.0
    assert ds.id2txt(ids) == __import__("json").loads(open(tmp_path / "train.json").readline())["text"]
    assert len(MyAudioDataset([str(tmp_path / "train.json")], labels, max_duration=0.1)) == 0      # duration filter
    assert load_wav(path).shape[0] == 1


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lightning_asr_amd.lightning_compat import GradSync
    torch.manual_seed(rank)
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    sync = GradSync([0, 3, 7, 10])
    works = sync.all_reduce(flat, async_op=True)
    for w in works:
        w.wait()
    q.put((rank, (flat * sync.grad_scale).tolist(), sync.world))
    dist.destroy_process_group()


def test_grad_sync_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = (torch.arange(10, dtype=torch.float32) * 1.5).tolist()       # mean of x*1 and x*2
    for rank, vals, world in res:
        assert world == 2 and vals == pytest.approx(expect)


def test_bucket_batch_sampler_padding_and_sharding():
    from lightning_asr_amd.data_module import BucketBatchSampler
    rng = np.random.default_rng(0)
    dur = rng.uniform(2.0, 16.0, 1000).tolist()
    s = BucketBatchSampler(dur, 32, bucket_batches=10, seed=1)
    batches = list(s)
    assert len(batches) == len(s) == 31 and all(len(b) == 32 for b in batches)
    flat = [i for b in batches for i in b]
    assert len(set(flat)) == len(flat)                               # no utterance twice in an epoch
    waste = np.mean([1 - np.sum([dur[i] for i in b]) / (32 * max(dur[i] for i in b)) for b in batches])
    assert waste < 0.10                                              # bucketed padding <= 10 % (BASELINE cfg5)
    rand = [list(range(i, i + 32)) for i in range(0, 992, 32)]
    assert waste < 0.3 * np.mean([1 - np.sum([dur[i] for i in b]) / (32 * max(dur[i] for i in b)) for b in rand])
    s.set_epoch(1)
    assert list(s) != batches                                        # reshuffled per epoch
    a = BucketBatchSampler(dur, 32, bucket_batches=10, seed=1, rank=0, world=2)
    b = BucketBatchSampler(dur, 32, bucket_batches=10, seed=1, rank=1, world=2)
    la, lb = list(a), list(b)
    assert len(la) == len(lb) == 15 and not ({i for x in la for i in x} & {i for x in lb for i in x})


def _write_wav(path, pcm, channels=1):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())


def test_native_wav_batch_reader_matches_python_wave(tmp_path):
    """lasr_wav_read_batch (host threads, no device work) == the python `wave` decode of data_module.load_wav, including the
    reference's bug-compatible random sub-sequence (data_module.py:138-148) taken as a slice of the file"""
    import numpy as np
    import torch
    from lightning_asr_amd import ingest
    from lightning_asr_amd.data_module import load_wav
    rng = np.random.default_rng(0)
    paths = []
    for i in range(9):
        n = 16000 + 321 * i
        ch = 2 if i == 4 else 1
        p = tmp_path / ("%d.wav" % i)
        _write_wav(p, rng.integers(-30000, 30000, size=(n, ch), dtype=np.int16), ch)
        paths.append(str(p))
    assert ingest.wav_info(paths[4]) == (16000 + 321 * 4, 2, 16000, 16)
    out = torch.empty(9 * 20000, dtype=torch.int16)
    lens = torch.empty(9, dtype=torch.int32)
    ld = ingest.read_wav_batch(paths, out, lens, n_threads=3)
    assert ld % 8 == 0 and ld >= 16000 + 321 * 8
    for i, p in enumerate(paths):
        ref = (load_wav(p)[0] * 32768).to(torch.int16)
        assert int(lens[i]) == ref.numel() and torch.equal(out[i * ld:i * ld + ref.numel()], ref)
        assert not out[i * ld + ref.numel():(i + 1) * ld].any()
    u = rng.uniform(size=(9, 2))
    ld2 = ingest.read_wav_batch(paths, out, lens, crop_u=u, crop_weight=0.98, n_threads=2)
    for i, p in enumerate(paths):
        x = load_wav(p)
        L = x.shape[1]
        tl = int(L * (0.98 + 0.02 * u[i, 0]))
        loc = int(u[i, 1] * (L - tl))
        ref = (x[:, loc:tl][0] * 32768).to(torch.int16)
        assert int(lens[i]) == ref.numel() and torch.equal(out[i * ld2:i * ld2 + ref.numel()], ref)
    # lead_in: the sample before the crop rides along (the reference crops AFTER pre-emphasis, data_module.py:157-159);
    # a crop that starts at the file's first sample has none; stereo files hand over channel 0's
    from lightning_asr_amd import _lib
    u[0, 1] = 0.0
    ld3 = ingest.read_wav_batch(paths, out, lens, crop_u=u, crop_weight=0.98, n_threads=2, lead_in=True)
    for i, p in enumerate(paths):
        x = load_wav(p)
        L = x.shape[1]
        tl = int(L * (0.98 + 0.02 * u[i, 0]))
        loc = int(u[i, 1] * (L - tl))
        lead = 1 if loc > 0 else 0
        assert (int(lens[i]) >> 30) & 1 == lead and int(lens[i]) & (_lib.LEN_LEAD - 1) == tl - loc
        ref = (x[:, loc - lead:tl][0] * 32768).to(torch.int16)
        assert torch.equal(out[i * ld3:i * ld3 + ref.numel()], ref) and not out[i * ld3 + ref.numel():(i + 1) * ld3].any()
    assert (int(lens[0]) >> 30) & 1 == 0 and ld3 >= max(int(v) & (_lib.LEN_LEAD - 1) for v in lens) + 1
    with pytest.raises(Exception, match="do not fit"):
        ingest.read_wav_batch(paths, out[:1000], lens)
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"not a wav file at all")
    with pytest.raises(Exception, match="RIFF"):
        ingest.read_wav_batch([str(bad)], out, lens)


def _riff(chunks):
    """a RIFF/WAVE file from (tag, payload) chunks (odd payloads get the pad byte the format asks for)"""
    import struct
    body = b"WAVE"
    for tag, payload in chunks:
        body += tag + struct.pack("<I", len(payload)) + payload + (b"\0" if len(payload) & 1 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_native_wav_reader_header_variants(tmp_path):
    """the header walk of lasr_wav_read_batch on the layouts real corpora contain: WAVE_FORMAT_EXTENSIBLE, chunks (odd-sized ones
    too) between "fmt " and "data", a streamed file whose data size field overshoots the file, an empty clip; what it must refuse
    (8-bit, float, another sample rate when one is demanded, a header cut short) it refuses with the file named"""
    import struct
    import numpy as np
    import torch
    from lightning_asr_amd import ingest
    rng = np.random.default_rng(3)
    pcm = rng.integers(-20000, 20000, size=3001, dtype=np.int16)
    fmt_pcm = struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)
    guid_pcm = struct.pack("<H", 1) + bytes.fromhex("000000001000800000aa00389b71")
    fmt_ext = struct.pack("<HHIIHHHHI", 0xFFFE, 1, 16000, 32000, 2, 16, 22, 16, 4) + guid_pcm
    cases = {
        "plain": _riff([(b"fmt ", fmt_pcm), (b"data", pcm.tobytes())]),
        "extensible": _riff([(b"fmt ", fmt_ext), (b"data", pcm.tobytes())]),
        "list_odd": _riff([(b"fmt ", fmt_pcm), (b"LIST", b"INFOabc"), (b"fact", struct.pack("<I", 3001)), (b"data", pcm.tobytes())]),
    }
    streamed = bytearray(cases["plain"])
    streamed[40:44] = struct.pack("<I", 0xFFFFFFFF)                 # data size written before the length was known
    cases["streamed"] = bytes(streamed)
    paths = []
    for name, blob in cases.items():
        p = tmp_path / (name + ".wav")
        p.write_bytes(blob)
        paths.append(str(p))
    empty = tmp_path / "empty.wav"
    empty.write_bytes(_riff([(b"fmt ", fmt_pcm), (b"data", b"")]))
    paths.append(str(empty))
    out = torch.full((len(paths) * 3008,), 7, dtype=torch.int16)
    lens = torch.empty(len(paths), dtype=torch.int32)
    ld = ingest.read_wav_batch(paths, out, lens, n_threads=2, expect_rate=16000)
    assert ld == 3008 and lens.tolist() == [3001] * 4 + [0]
    ref = torch.from_numpy(pcm.copy())
    for i in range(4):
        assert torch.equal(out[i * ld:i * ld + 3001], ref) and not out[i * ld + 3001:(i + 1) * ld].any()
    assert not out[4 * ld:5 * ld].any()
    assert ingest.wav_info(paths[1]) == (3001, 1, 16000, 16)
    refused = {
        "8-bit": (_riff([(b"fmt ", struct.pack("<HHIIHH", 1, 1, 16000, 16000, 1, 8)), (b"data", b"\x80" * 100)]), "16-bit"),
        "float": (_riff([(b"fmt ", struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32)), (b"data", b"\0" * 400)]), "integer PCM"),
        "cut": (cases["plain"][:30], "fmt|data"),
        "no_data": (_riff([(b"fmt ", fmt_pcm), (b"LIST", b"INFO")]), "no data chunk"),
    }
    for name, (blob, msg) in refused.items():
        p = tmp_path / (name + ".wav")
        p.write_bytes(blob)
        with pytest.raises(Exception, match=msg) as ei:
            ingest.read_wav_batch([str(p)], out, lens)
        assert name + ".wav" in str(ei.value)
    p8k = tmp_path / "8k.wav"
    p8k.write_bytes(_riff([(b"fmt ", struct.pack("<HHIIHH", 1, 1, 8000, 16000, 2, 16)), (b"data", pcm.tobytes())]))
    with pytest.raises(Exception, match="sample rate"):
        ingest.read_wav_batch([str(p8k)], out, lens, expect_rate=16000)
    assert ingest.read_wav_batch([str(p8k)], out, lens) == 3008        # the default takes any rate, like the reference (data_module.py:153)


def test_batch_producer_fills_ring_slots(tmp_path):
    """manifest -> BatchProducer -> HostBatch: PCM rows, lens, padded targets, SpecAugment rectangles, metadata layout"""
    import json
    import numpy as np
    import torch
    from lightning_asr_amd import ingest
    from lightning_asr_amd.data_module import AudioParser, MyAudioDataset
    rng = np.random.default_rng(1)
    labels = list("abcdefg")
    man = tmp_path / "m.json"
    with open(man, "w") as f:
        for i in range(6):
            n = 8000 + 1000 * i
            p = tmp_path / ("c%d.wav" % i)
            _write_wav(p, rng.integers(-1000, 1000, size=(n, 1), dtype=np.int16))
            f.write(json.dumps({"audio_filepath": str(p), "duration": n / 16000.0, "text": "abc"[: 1 + i % 3] + "g"}) + "\n")
    ds = MyAudioDataset([str(man)], labels, mask=True)
    assert ingest.fast_ingest_ok(ds)
    ring = ingest.PinnedRing(2, 3 * 14000, 64, pin=False)
    ap = AudioParser.__new__(AudioParser)
    import random
    ap.rand = random.Random(0)
    prod = ingest.BatchProducer(ds, [[0, 1, 2], [3, 4, 5]], ring, mask=True, audio_parser=ap, n_threads=2, crop=False)
    prod.start()
    got = []
    while True:
        hb = prod.out.get(timeout=30)
        if hb is None:
            break
        assert not isinstance(hb, BaseException), hb
        got.append(hb)
    assert len(got) == 2 and [hb.slot for hb in got] == [0, 1]
    hb = got[1]
    assert hb.B == 3 and hb.lens.tolist() == [11000, 12000, 13000] and hb.ld == 13000 and hb.S == 4
    assert hb.sizes.tolist() == [2, 3, 4] and hb.targets[2].tolist() == [0, 1, 2, 6] and hb.targets[0].tolist() == [0, 6, 0, 0]
    assert hb.aug.shape == (3, 4) and (hb.aug[:, 1] < 27).all() and abs(hb.seconds - 36000 / 16000.0) < 1e-9
    pcm = ring.pcm[hb.slot][:3 * hb.ld].view(3, hb.ld)
    from lightning_asr_amd.data_module import load_wav
    ref = (load_wav(ds.datasets[4]["audio_filepath"])[0] * 32768).to(torch.int16)
    assert torch.equal(pcm[1, :12000], ref) and not pcm[1, 12000:].any()
    # the packed metadata block holds the same values at the documented offsets
    o_lens, o_sizes, o_aug, o_tg, words = ingest._meta_layout(3, 4, True)
    assert hb.meta[o_lens:o_lens + 3].tolist() == [11000, 12000, 13000] and words == hb.meta_words
    assert hb.meta[o_tg:o_tg + 24].view(torch.int64).view(3, 4)[2].tolist() == [0, 1, 2, 6]


def test_batch_producer_hands_errors_to_the_consumer_and_grows_a_short_slot(tmp_path):
    """a missing wav file surfaces as an exception object on the producer's queue (the trainer re-raises it); a clip longer than the
    ring slot was sized for (manifest duration too small) makes the producer grow the slot instead of failing"""
    import json
    import random
    import numpy as np
    from lightning_asr_amd import ingest
    from lightning_asr_amd.data_module import AudioParser, MyAudioDataset
    rng = np.random.default_rng(2)
    man = tmp_path / "m.json"
    with open(man, "w") as f:
        for i in range(3):
            p = tmp_path / ("d%d.wav" % i)
            if i != 1:
                _write_wav(p, rng.integers(-1000, 1000, size=(20000, 1), dtype=np.int16))
            f.write(json.dumps({"audio_filepath": str(p), "duration": 0.1, "text": "ab"}) + "\n")      # duration understated
    ds = MyAudioDataset([str(man)], list("ab"), mask=False)
    ap = AudioParser.__new__(AudioParser)
    ap.rand = random.Random(0)

    def drain(batches, cap):
        ring = ingest.PinnedRing(2, cap, 64, pin=False)
        prod = ingest.BatchProducer(ds, batches, ring, mask=False, audio_parser=ap, n_threads=2)
        prod.start()
        out = []
        while True:
            it = prod.out.get(timeout=30)
            out.append(it)
            if it is None or isinstance(it, BaseException):
                return out, ring
    out, ring = drain([[0, 2]], 2 * 1600 + 64)               # sized for 0.1 s clips, the files hold 1.25 s
    assert isinstance(out[0], ingest.HostBatch) and out[0].lens.tolist() == [20000, 20000] and out[1] is None
    assert ring.pcm[out[0].slot].numel() >= 2 * 20000
    out, _ = drain([[0, 1]], 2 * 20000 + 64)
    assert isinstance(out[-1], BaseException) and "cannot open" in str(out[-1])


def test_native_ingest_probes_the_manifest_before_it_takes_over(tmp_path):
    """fast_ingest_ok: the native reader decodes 16-bit PCM RIFF/WAVE only; a manifest that points at anything else (flac, 24-bit
    wav) keeps the DataLoader route instead of aborting the fit on its first batch"""
    import json
    import numpy as np
    from lightning_asr_amd import ingest
    from lightning_asr_amd.data_module import MyAudioDataset
    labels = list("abc")
    good = []
    for i in range(5):
        p = tmp_path / ("g%d.wav" % i)
        _write_wav(p, np.zeros((400, 1), dtype=np.int16), 1)
        good.append(str(p))
    flac = tmp_path / "x.flac"
    flac.write_bytes(b"fLaC" + b"\0" * 64)

    def manifest(name, paths):
        m = tmp_path / name
        with open(m, "w") as f:
            for p in paths:
                f.write(json.dumps({"audio_filepath": p, "duration": 0.025, "text": "ab"}) + "\n")
        return str(m)
    assert ingest.fast_ingest_ok(MyAudioDataset([manifest("good.json", good)], labels))
    assert not ingest.fast_ingest_ok(MyAudioDataset([manifest("mixed.json", good[:2] + [str(flac)] + good[2:])], labels, ), probe=8)
    assert not ingest.fast_ingest_ok(MyAudioDataset([manifest("last.json", good + [str(flac)])], labels))


def test_fit_reserves_the_workspace_for_the_longest_clip():
    """FusedLoop sizes the model workspace once, for the longest clip (capped by train_max_duration) and transcript of the training
    set: growing it batch by batch stalled the GPU behind multi-GB allocations."""
    from types import SimpleNamespace
    from lightning_asr_amd import ops
    from lightning_asr_amd.fused_fit import FusedLoop
    calls = []
    items = [{"duration": 3.2, "text": "abc"}, {"duration": 12.5, "text": "a" * 41}, {"duration": 30.0, "text": "zz"}]
    me = SimpleNamespace(dm=SimpleNamespace(train_datasets=SimpleNamespace(datasets=items), train_max_duration=16.7, train_bs=24),
                         native=SimpleNamespace(workspace=lambda B, T, S: calls.append((B, T, S))))
    FusedLoop._reserve_workspace(me)
    assert calls == [(24, ops.mel_num_frames(int(16.7 * 16000 + 0.5) + 1), 41)]
    FusedLoop._reserve_workspace(me)                       # once per loop
    assert len(calls) == 1
    me2 = SimpleNamespace(dm=SimpleNamespace(train_datasets=SimpleNamespace(datasets=[]), train_bs=8), native=me.native)
    FusedLoop._reserve_workspace(me2)                      # nothing to size from: no call
    assert len(calls) == 1
