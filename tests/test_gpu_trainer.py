"""GPU tier: the reference's surface (LightingModule + LibriDataModule + Trainer.fit, train.py:64-86,233-252) drives the FUSED
native step - the path bench.py measures - and lands bit for bit where a hand-driven TrainStep lands on the same batches."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LABELS = [c.strip() for c in open(os.path.join(ROOT, "data", "labels.txt")).readlines()]


def _corpus(tmp_path, n_train=16, seconds=2.0, ragged=False, labels=None, n_dev=4):
    data = tmp_path / "synth"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "make_synth_data.py"), "--out", str(data), "--n-train", str(n_train),
           "--n-dev", str(n_dev), "--seconds", str(seconds)] + (["--ragged"] if ragged else [])
    if labels:
        cmd += ["--labels", labels]
    subprocess.run(cmd, check=True)
    return data


def test_mel_pcm16_device_dither_is_bit_identical_to_f32_with_noise_tensor(dev):
    """int16 PCM + dither generated in the kernel == the f32 waveform (pcm / 32768) + the same noise as a tensor"""
    from lightning_asr_amd import ops
    g = torch.Generator().manual_seed(3)
    B, L = 3, 16000 + 123
    pcm = torch.randint(-20000, 20000, (B, L), generator=g, dtype=torch.int16)
    lens = torch.tensor([L, 9000, 777], dtype=torch.int32)
    for i in range(B):
        pcm[i, lens[i]:] = 0
    pcm_d, lens_d = pcm.to(dev), lens.to(dev)
    f32 = (pcm.float() / 32768.0).to(dev)
    dd = ops.DeviceDither(1234, dev)
    noise = dd.noise(B, L)
    n = noise.cpu()
    assert abs(float(n.mean())) < 0.02 and abs(float(n.std()) - 1.0) < 0.02 and float(n.abs().max()) < 7.0
    assert abs(float((n[:, 1:] * n[:, :-1]).mean())) < 0.02                       # neighbouring samples uncorrelated
    bft_a, btf_a, fr_a, pct_a = ops.mel(pcm_d, lens_d, dd, None, True, torch.float32)
    assert int(dd.step.item()) == 1                                               # the call consumed one draw
    bft_b, btf_b, fr_b, pct_b = ops.mel(f32, lens_d, noise, None, True, torch.float32)
    assert torch.equal(bft_a, bft_b) and torch.equal(btf_a, btf_b) and torch.equal(fr_a, fr_b) and torch.equal(pct_a, pct_b)
    # no dither at all: the int16 path equals the f32 path; and dither moves the features by ~1e-5-sized noise only
    bft_c = ops.mel(pcm_d, lens_d, None, None, True, torch.float32)[0]
    bft_d = ops.mel(f32, lens_d, None, None, True, torch.float32)[0]
    assert torch.equal(bft_c, bft_d)
    assert 0 < float((bft_a - bft_c).abs().max()) < 0.05
    noise2 = dd.noise(B, L)
    assert not torch.equal(noise, noise2)                                         # next step, fresh noise
    # the prefetch route (features inside the CTC launch) draws the same noise as the stand-alone call
    from lightning_asr_amd.engine import NativeModel
    m = NativeModel("plain", 28, mask=True, dtype=torch.float32, device=dev)
    m.init_parameters(0)
    step_before = dd.step.clone()
    want = ops.mel(pcm_d, lens_d, dd, None, True, torch.float32, want_bft=False)[1]
    dd.step.copy_(step_before)
    f0 = ops.mel(pcm_d, lens_d, None, None, True, torch.float32, want_bft=False)
    nf, npct = m.arm_prefetch(pcm_d, lens_d, dd, None)
    tg = torch.randint(0, 27, (B, 5), generator=g).to(dev)
    m.loss_backward(f0[1], f0[3], tg, torch.full((B,), 5, dtype=torch.int32, device=dev))
    torch.cuda.synchronize()
    assert torch.equal(nf, want)


def _fit_and_replay(dev, tmp_path, dtype, steps, crop, monkeypatch, graph, drop_rate=0.0):
    from lightning_asr_amd import ops
    from lightning_asr_amd.data_module import LibriDataModule
    from lightning_asr_amd.engine import NativeModel
    from lightning_asr_amd.lightning_compat import Trainer, seed_everything
    from lightning_asr_amd.schedule import CosineAnnealingWarmupRestarts
    from lightning_asr_amd.step import TrainStep
    from lightning_asr_amd.train import LightingModule
    monkeypatch.setenv("LASR_TRAINER_GRAPH", "1" if graph else "0")
    data = _corpus(tmp_path, n_train=4 * steps, seconds=2.0)
    seed_everything(0)
    act = torch.float32 if dtype == "f32" else torch.bfloat16
    dm = LibriDataModule([str(data / "train.json")], str(data / "dev.json"), str(data / "dev.json"), LABELS, train_bs=4, dev_bs=4,
                         num_worker=2, device=str(dev), act_dtype=act, train_crop=crop)
    model = LightingModule(learning_rate=1e-2, weight_decay=1e-3, labels=LABELS, total_epoch=1, drop_rate=drop_rate, mask=True, use_cer=True,
                           dtype=dtype, device=str(dev), warmup_steps=2)
    init = {"params": model.encoder.native.params.clone(), "buffers": model.encoder.native.buffers.clone()}
    seen = []

    def hook(db):
        seen.append({"pcm": db.pcm.clone(), "lens": db.lens.clone(), "targets": db.targets.clone(), "sizes": db.sizes.clone(),
                     "aug": None if db.aug is None else db.aug.clone(), "L": db.L})      # L: the longest utterance (rows may be wider)
    tr = Trainer(max_epochs=1, default_root_dir=str(tmp_path / "run"), device=str(dev), check_val_every_n_epoch=1, log_every_n_steps=2)
    tr._fused_on_batch = hook
    hist = tr.fit(model, dm)
    assert tr.fused is not None and tr.fused.source_kind == "NativeSource"            # the fused step over the native ingest ran
    assert tr.global_step == steps and len(seen) == steps
    assert np.isfinite(hist[-1]["train_loss"]) and hist[-1]["train_wer"] >= 0 and "val_wer_total" in hist[-1]
    if graph:
        assert tr.fused.graph_steps > 0, "fixed-shape batches must reach the captured graph"
    else:
        assert tr.fused.graph_steps == 0
    # the same batches through a hand-driven TrainStep (the bench path): same seed for the in-kernel dither
    m2 = NativeModel("plain", 28, mask=True, act="relu", dtype=act, device=dev)
    m2.params.copy_(init["params"]); m2.buffers.copy_(init["buffers"])
    if drop_rate:      # nn.Dropout(p=drop_rate): the same counter-based masks (seed from pl.seed_everything, step counter on the device)
        m2.set_dropout(drop_rate, model.encoder.native.drop_seed)
    sched = CosineAnnealingWarmupRestarts(None, first_cycle_steps=1 * steps, cycle_mult=2, max_lr=1e-2, min_lr=1e-4, warmup_steps=2, gamma=0.5)
    ts = TrainStep(m2, 1e-2, 1e-3, schedule=sched)
    dd = ops.DeviceDither(tr.fused.dither.seed, dev)
    for i, b in enumerate(seen):
        nxt = seen[i + 1] if i + 1 < len(seen) else None
        ts.step(b["pcm"], b["targets"], b["sizes"], sample_lens=b["lens"], dither=dd, aug=b["aug"],
                prefetch_wave=None if nxt is None else nxt["pcm"], prefetch_lens=None if nxt is None else nxt["lens"],
                prefetch_dither=None if nxt is None else dd, prefetch_aug=None if nxt is None else nxt["aug"], want_logp=False,
                logical_len=b["L"], prefetch_logical_len=None if nxt is None else nxt["L"])
    torch.cuda.synchronize()
    # (validation ran at the end of the epoch and does not touch the parameters; BN running statistics are training-only)
    assert torch.equal(model.encoder.native.params, m2.params), float((model.encoder.native.params - m2.params).abs().max())
    assert torch.equal(model.encoder.native.buffers, m2.buffers)
    return tr


def test_trainer_fit_equals_trainstep_bit_identical_eager_f32(dev, tmp_path, monkeypatch):
    _fit_and_replay(dev, tmp_path, "f32", 4, True, monkeypatch, graph=False)


def test_trainer_fit_equals_trainstep_bit_identical_graph_bf16(dev, tmp_path, monkeypatch):
    """fixed-length clips, no crop: from the third sighting of the batch shape the steps replay a captured hipGraph"""
    tr = _fit_and_replay(dev, tmp_path, "bf16", 8, False, monkeypatch, graph=True)
    assert tr.fused.eager_steps >= 2


def test_trainer_fit_with_dropout_replays_fresh_masks_from_the_graph(dev, tmp_path, monkeypatch):
    """model.drop_rate > 0 through Trainer.fit: the masks are regenerated from (seed, device step counter), so the graph-replayed steps
    draw the masks the eager hand-driven steps draw - same parameters bit for bit - and validation (eval mode) runs without dropout"""
    _fit_and_replay(dev, tmp_path, "bf16", 6, False, monkeypatch, graph=True, drop_rate=0.1)


def test_trainer_uses_the_lean_head_for_a_large_vocabulary(dev, tmp_path):
    """C = 4334 (AISHELL) in bf16: Trainer.fit must not materialise (B, T', C) f32 log-probs (BASELINE cfg5 through train.py)"""
    vocab = os.path.join(ROOT, "data", "aishell1-vocab.txt")
    data = _corpus(tmp_path, n_train=8, seconds=3.0, ragged=True, labels=vocab)
    from lightning_asr_amd.train import main
    out = tmp_path / "run"
    tr = main(["data.train_manifest=[%s]" % (data / "train.json"), "data.val_manifest=%s" % (data / "dev.json"),
               "data.test_manifest=%s" % (data / "dev.json"), "data.labels=%s" % vocab, "train.train_batch_size=4",
               "train.dev_batch_size=4", "train.total_epoch=1", "train.precision=16", "train.warmup_steps=0",
               "data.bucket_by_length=true", "data.bucket_batches=2", "output_dir=%s" % out])
    assert tr.fused is not None and tr.global_step == 2
    native = tr.fused.native
    assert native.lean_head and native._last_logp is None          # the step ran on bf16 logits + row statistics
    rec = tr.history[-1]
    assert np.isfinite(rec["train_loss"]) and rec["train_loss"] > 0 and np.isfinite(rec["val_loss"])


def test_aishell2_labels_take_the_host_cer_path(dev, tmp_path):
    """data/aishell2-labels.txt (5 206 labels; its first line is a blank, so `c.strip()` at train.py:217 makes label 0 the EMPTY
    string): C = 5207 through train.main in bf16.  The lean head carries the step (256 <= C <= 9216); the CER units of such a
    vocabulary are not token ids (an emitted label 0 adds no character to the joined string, utils/asr_metrics.py:215-216), so the
    metric must take the host path every step - and agree with the reference's definition on a hand-made case."""
    from lightning_asr_amd.utils.asr_metrics import WER
    vocab = os.path.join(ROOT, "data", "aishell2-labels.txt")
    labels = [c.strip() for c in open(vocab, encoding="utf-8").readlines()]
    assert len(labels) == 5206 and labels[0] == "" and all(len(c) == 1 for c in labels[1:]) and len(set(labels)) == 5206
    w = WER(labels, use_cer=True)
    assert not w.device_ok
    blank = len(labels)
    ids = torch.tensor([[5, 0, blank, 7, 7, blank, 9]], dtype=torch.int32, device=dev)     # collapses to labels 5, 0 (= ''), 7, 9
    tg = torch.tensor([[5, 7, 8]])
    val = float(w(ids, tg, torch.tensor([3]), torch.tensor([7], dtype=torch.int32, device=dev)))
    assert val == pytest.approx(1 / 3)               # "ab?" vs "abc"-like: the empty label is invisible, one substitution of three
    data = _corpus(tmp_path, n_train=8, seconds=3.0, ragged=True, labels=vocab)
    from lightning_asr_amd.train import main
    tr = main(["data.train_manifest=[%s]" % (data / "train.json"), "data.val_manifest=%s" % (data / "dev.json"),
               "data.test_manifest=%s" % (data / "dev.json"), "data.labels=%s" % vocab, "train.train_batch_size=4",
               "train.dev_batch_size=4", "train.total_epoch=1", "train.precision=16", "train.warmup_steps=0",
               "output_dir=%s" % (tmp_path / "run")])
    assert tr.fused is not None and tr.global_step == 2
    assert tr.fused.native.n_class == 5207 and tr.fused.native.lean_head and tr.fused.native._last_logp is None
    assert not tr.fused.model.wer.device_ok
    rec = tr.history[-1]
    assert np.isfinite(rec["train_loss"]) and rec["train_loss"] > 0 and np.isfinite(rec["train_wer"]) and np.isfinite(rec["val_wer"])


def test_bucketed_fit_keeps_padding_under_ten_percent(dev, tmp_path):
    """BASELINE cfg5 'bucketed padding': data.bucket_by_length from the config, ragged 2-8 s manifest"""
    data = _corpus(tmp_path, n_train=96, seconds=8.0, ragged=True)
    from lightning_asr_amd.train import main
    common = ["data.train_manifest=[%s]" % (data / "train.json"), "data.val_manifest=%s" % (data / "dev.json"),
              "data.test_manifest=%s" % (data / "dev.json"), "data.labels=%s" % os.path.join(ROOT, "data", "labels.txt"),
              "train.train_batch_size=8", "train.dev_batch_size=4", "train.total_epoch=1", "train.precision=16", "train.warmup_steps=0",
              "train.check_val_every_n_epoch=5"]
    tr = main(common + ["data.bucket_by_length=true", "data.bucket_batches=12", "output_dir=%s" % (tmp_path / "b")])
    f = tr.fused
    pad_b = 1.0 - f.samples_real / f.samples_padded
    tr2 = main(common + ["output_dir=%s" % (tmp_path / "u")])
    f2 = tr2.fused
    pad_u = 1.0 - f2.samples_real / f2.samples_padded
    assert tr.global_step == 12 and pad_b <= 0.10 < pad_u, (pad_b, pad_u)


def test_wer_host_fallback_for_multichar_labels(dev):
    """CER over a vocabulary with a multi-character entry: the device token units are not the reference's characters
    (utils/asr_metrics.py:215-216), so the metric takes the host path; both paths agree when labels are single characters"""
    from lightning_asr_amd.utils.asr_metrics import WER
    labels = ["a", "b", "<unk>", "c"]
    w = WER(labels, use_cer=True)
    assert not w.device_ok
    ids = torch.tensor([[0, 4, 2, 2, 4, 3]], dtype=torch.int32, device=dev)        # "a<unk>c"
    tg = torch.tensor([[0, 1, 3]])                                                   # "abc"
    val = float(w(ids, tg, torch.tensor([3]), torch.tensor([6], dtype=torch.int32, device=dev)))
    assert val == pytest.approx(5 / 3)               # characters: "a<unk>c" vs "abc" = 5 edits over 3 reference characters
    w2 = WER(["a", "b", "d", "c"], use_cer=True)
    assert w2.device_ok
    val2 = float(w2(ids, tg, torch.tensor([3]), torch.tensor([6], dtype=torch.int32, device=dev)))
    assert val2 == pytest.approx(1 / 3)
    assert float(w2.compute_total()) == pytest.approx(1 / 3)


def test_lean_head_computes_the_prefetched_features_inside_its_lattice_launch(dev):
    """large-vocabulary head (bf16, C >= 256): the next batch's log-mel features ride in the grid of the compact lattice kernel;
    same features and the same loss / gradients, bit for bit, as the separate launches"""
    from lightning_asr_amd import ops
    from lightning_asr_amd.engine import NativeModel
    g = torch.Generator().manual_seed(11)
    B, L, S, C = 3, 24000, 9, 300
    wave = (0.1 * torch.randn(B, L, generator=g)).to(dev)
    nxt = (0.1 * torch.randn(B, 31000, generator=g)).to(dev)
    nlens = torch.tensor([31000, 20000, 9000], dtype=torch.int32, device=dev)
    tg = torch.randint(0, C - 1, (B, S), generator=g).to(dev)
    tl = torch.tensor([9, 5, 1], dtype=torch.int32, device=dev)
    res = []
    for fused in (False, True):
        m = NativeModel("plain", C, mask=True, dtype=torch.bfloat16, device=dev)
        m.init_parameters(4)
        assert m.lean_head
        _, feats, _, pct = ops.mel(wave, None, None, None, True, torch.bfloat16, want_bft=False)
        nf = None
        if fused:
            nf, npct = m.arm_prefetch(nxt, nlens)
        loss, nll, logp, am = m.loss_backward(feats, pct, tg, tl, want_logp=False)
        assert logp is None
        if not fused:
            _, nf, _, npct = ops.mel(nxt, nlens, None, None, True, torch.bfloat16, want_bft=False)
        torch.cuda.synchronize()
        res.append((loss.clone(), nll.clone(), m.grads.clone(), nf.clone(), npct.clone(), am.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[0][2]).all()


def test_custom_dataset_and_overridden_step_keep_working(dev, tmp_path):
    """(1) a dataset that overrides __getitem__ cannot use the native wav reader: the fused step then takes its batches from the
    DataLoader (HostWaveSource); (2) a module that overrides training_step keeps the autograd route of Trainer.fit"""
    from lightning_asr_amd.data_module import LibriDataModule, MyAudioDataset
    from lightning_asr_amd.lightning_compat import Trainer, seed_everything
    from lightning_asr_amd.train import LightingModule
    data = _corpus(tmp_path, n_train=8, seconds=2.0)

    class MyDS(MyAudioDataset):
        def __getitem__(self, index):
            w, ids, path = super().__getitem__(index)
            return w * 0.5, ids, path

    class MyDM(LibriDataModule):
        def setup(self, stage=None):
            super().setup(stage)
            self.train_datasets = MyDS(self.train_manifest, self.labels, mask=True, max_duration=self.train_max_duration)

    def run(module_cls, dm_cls, root):
        seed_everything(0)
        dm = dm_cls([str(data / "train.json")], str(data / "dev.json"), str(data / "dev.json"), LABELS, train_bs=4, dev_bs=4, num_worker=0,
                    device=str(dev), act_dtype=torch.bfloat16)
        model = module_cls(learning_rate=1e-2, weight_decay=1e-3, labels=LABELS, total_epoch=1, mask=True, use_cer=True, dtype="bf16",
                           device=str(dev), warmup_steps=0)
        tr = Trainer(max_epochs=1, default_root_dir=str(tmp_path / root), device=str(dev))
        hist = tr.fit(model, dm)
        assert tr.global_step == 2 and np.isfinite(hist[-1]["train_loss"]) and hist[-1]["train_wer"] >= 0
        return tr

    tr = run(LightingModule, MyDM, "a")
    assert tr.fused is not None and tr.fused.source_kind == "HostWaveSource"

    class MyModule(LightingModule):
        def training_step(self, batch, batch_idx):
            return super().training_step(batch, batch_idx) * 1.0

    tr = run(MyModule, LibriDataModule, "b")
    assert tr.fused is None                                  # training_step -> loss.backward() -> Novograd.step


def test_rows_wider_than_the_longest_utterance_keep_the_reference_frame_count(dev, tmp_path):
    """The reference pads a batch to its longest FEATURE matrix (data_module.py:222-248): Tmax = 1 + (longest + 64) // 160.  Until
    round 5 Tmax followed the device rows' width - the longest row rounded up to 8 samples, lead-in sample included - which is one
    frame too many whenever the rounding crosses a frame boundary.  Now the rows' width travels as `lasr_wave_src.pitch` and the
    call's L is the longest utterance:
      (1) ops.mel(wave (B, P), logical_len = L): bit-identical to the tight (B, L) call, T = frames(L);
      (2) the native ingest on files of 16 095 and 12 000 samples (16 095 + 64 = 101 * 160 - 1: rounding the row to 16 096 samples used
          to give 102 frames): 101 frames, pct as R.collate computes it, features within MEL_TOL of the oracle, and the device rows
          have ONE pitch per frame-count class (16 096 = 160 * 100 + 96) whatever the individual lengths are."""
    import wave as wavmod
    from conftest import MEL_TOL
    from lightning_asr_amd import ops
    from lightning_asr_amd.ingest import BatchProducer, DeviceFeeder, PinnedRing
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(5)
    # (1)
    B, L, P = 3, 16095, 16096 + 64
    wave = 0.1 * torch.randn(B, L, generator=g)
    lens = torch.tensor([L, 12000, 300], dtype=torch.int32)
    wide = torch.full((B, P), 7.0)                       # the rows' tails hold garbage: never read
    wide[:, :L] = wave
    noise = torch.randn(B, L, generator=g)
    wide_n = torch.zeros(B, P); wide_n[:, :L] = noise
    a = ops.mel(wave.to(dev), lens.to(dev), noise.to(dev), None, True)
    b = ops.mel(wide.to(dev), lens.to(dev), wide_n.to(dev), None, True, logical_len=L)
    assert a[0].shape == b[0].shape == (B, 64, 101) and all(torch.equal(x, y) for x, y in zip(a, b))
    c = ops.mel(wide.to(dev), lens.to(dev), wide_n.to(dev), None, True)          # without logical_len T follows the rows: 102 frames
    assert c[0].shape[2] == 1 + (P + 64) // 160 and torch.equal(c[0][:, :, :101], a[0]) and bool((c[0][:, :, 101:] == 0).all())
    pcm = (wave.clamp(-1, 1) * 32767).round().to(torch.int16)
    wide16 = torch.full((B, P), 1234, dtype=torch.int16); wide16[:, :L] = pcm
    dd1, dd2 = ops.DeviceDither(77, dev), ops.DeviceDither(77, dev)
    a16 = ops.mel(pcm.to(dev), lens.to(dev), dd1, None, True)
    b16 = ops.mel(wide16.to(dev), lens.to(dev), dd2, None, True, logical_len=L)
    assert all(torch.equal(x, y) for x, y in zip(a16, b16))
    # (2)
    files, lens_f = [], [16095, 12000]
    for i, n in enumerate(lens_f):
        x = (0.1 * torch.randn(n, generator=g)).clamp(-1, 1).mul(32767).round().to(torch.int16)
        p = str(tmp_path / ("f%d.wav" % i))
        with wavmod.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.numpy().tobytes())
        files.append((p, x))

    class DS:                                            # the stock dataset's attributes the producer reads
        datasets = [{"audio_filepath": p, "text": "ab", "duration": x.numel() / 16000} for p, x in files]
        char2index = {"a": 0, "b": 1}
    ring = PinnedRing(2, 2 * 16200, 4096)
    feeder = DeviceFeeder(ring, dev, n_slots=2)
    prod = BatchProducer(DS(), [[0, 1]], ring, mask=False, audio_parser=None, n_threads=2, crop=False, feeder=feeder)
    db = feeder.upload(prod.make([0, 1], ring.free.get(), 0))
    assert db.pitch == 160 * 100 + 96 and db.L == 160 * 100 + 95 and tuple(db.pcm.shape) == (2, db.pitch) and db.key[1] == db.pitch
    torch.cuda.current_stream().wait_event(db.ready)
    bft, _, frames, pct = ops.mel(db.pcm, db.lens, None, None, True, logical_len=db.L)
    feats = [R.parse_wave((x.float() / 32768.0).unsqueeze(0)) for _, x in files]
    inputs, _, pct_ref, _ = R.collate(feats, [[0, 1], [0, 1]])
    assert bft.shape[2] == inputs.shape[3] == 101                                  # (the row width alone would say 102)
    assert frames.cpu().tolist() == [f.shape[2] for f in feats] and torch.equal(pct.cpu(), pct_ref)
    assert float((bft.cpu() - inputs[:, 0]).abs().max() / inputs.abs().max()) < MEL_TOL + 5e-5      # (f32 oracle: + its own round-off)
    feeder.close()
