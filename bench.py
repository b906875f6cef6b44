#!/usr/bin/env python
"""Throughput of the QuartzNet-CTC training hot path on MI355X (BASELINE.json metric): audio-seconds/sec of training.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg4|cfg5] [--dtype bf16|f32]

  cfg2 (default; the configuration BASELINE.json's metric is quoted on): asr13x1, bs=32/GPU, 10 s 16 kHz clips, labels.txt (C=28)
  cfg4: QuartNetContextSE (SE + BiLSTM context block), bs=32, 10 s clips, C=28
  cfg5: asr13x1 with the AISHELL character vocabulary (data/aishell1-vocab.txt, C=4334), bs=32, variable-length 2-16 s clips in
        length buckets (<= 10 % padding); audio-seconds count the REAL (unpadded) audio

A "step" is one full pass of the hot path over one batch already resident in HBM:
    wave -> log-mel -> forward -> mean CTC -> backward -> grad all-reduce (N>1) -> NovoGrad -> LR step.
Not in the timed step (stated in config.excluded): the H2D copy of the PCM (20 MB/step at cfg2, ~0.35 ms at 63 GB/s) and the
per-step greedy decode + WER logging that the reference's training_step does for its progress bar (train.py:80).
By default the log-mel stage is software-pipelined across steps like a data-loader prefetch: step i computes the features of
step i+1's waveforms inside its CTC launch (one batch of features per step either way; --no-prefetch puts them back at the head).

--path trainer: the SAME step driven through the reference's surface (LightingModule + LibriDataModule + Trainer.fit over a
synthetic wav corpus + manifest): wav decode by the library's host threads, int16 H2D, the training-time random crop and SpecAugment
draws, and the per-step greedy decode + WER logging are then INSIDE the timed region (stated in config.included).

N>1: ``python bench.py --gpus N`` starts its own N per-GPU worker processes (as the reference's Lightning DDP does from a plain
``python train.py``); under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (the driver's form) every
rank supervises one worker.  Either way a failed attempt is retried in fresh processes one rung down (graph + lasr_comm -> eager +
lasr_comm -> eager + torch.distributed; lightning_asr_amd/launch.py) and the line's `launcher` record says which rung ran.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 16000
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_MFMA_TF = 157.3     # dense f32-input MFMA
PEAK_BF16_MFMA_TF = 2500.0   # dense bf16 MFMA

# algorithmic HBM bytes of one training step (two-pass-BN minimum-materialisation model, bf16 activations)
STEP_MB_PER_UTT = {"cfg2": 108.5, "cfg4": 119.0}      # per 10 s utterance
STEP_MB_PER_AUDIO_S = {"cfg5": 16.9}                  # per second of (unpadded) audio
PROF_KINDS = ("gemm", "dwconv", "bn", "head", "other")

CONFIGS = {
    "cfg2": {"variant": "plain", "vocab": "data/labels.txt", "clip_s": 10.0, "ragged": False,
             "workload": "asr13x1 QuartzNet-CTC train step, bs=%d/GPU, 10 s synthetic 16 kHz clips, labels.txt vocab (C=28)"},
    "cfg4": {"variant": "context_se", "vocab": "data/labels.txt", "clip_s": 10.0, "ragged": False,
             "workload": "QuartNetContextSE (SE + BiLSTM context block) train step, bs=%d/GPU, 10 s synthetic 16 kHz clips, labels.txt vocab (C=28)"},
    "cfg5": {"variant": "plain", "vocab": "data/aishell1-vocab.txt", "clip_s": None, "ragged": True,
             "workload": "asr13x1 train step, AISHELL char vocab (C=4334), bs=%d/GPU, variable-length 2-16 s synthetic clips in 8 length "
                         "buckets (<=10%% padding), audio-seconds = unpadded audio"},
}


def vocab_size(path: str) -> int:
    with open(os.path.join(ROOT, path), encoding="utf-8") as f:
        return sum(1 for _ in f)


def _targets(B, S, V, g):
    tg = torch.randint(0, V, (B, S), generator=g)
    for s in range(1, S):
        same = tg[:, s] == tg[:, s - 1]
        tg[same, s] = (tg[same, s] + 1) % V
    return tg.long()


def synth_batch(B: int, n_samples: int, S: int, seed: int, device, V: int = 27):
    """wave = 0.1 N(0,1); targets U{0..V-1} without adjacent repeats (always CTC-feasible)."""
    g = torch.Generator().manual_seed(seed)
    wave = 0.1 * torch.randn(B, n_samples, generator=g)
    tg = _targets(B, S, V, g)
    return wave.to(device), tg.to(device), torch.full((B,), S, dtype=torch.int32, device=device)


def synth_buckets(B: int, n_buckets: int, V: int, seed: int, device, lo_s: float = 2.0, hi_s: float = 16.0):
    """L ~ U{2 s .. 16 s}, sorted into length buckets of B clips; targets S = floor(2.8 * seconds).
    Returns [(wave (B, Lmax) zero-padded, sample_lens (B) i32, targets (B, Smax), target_lens (B) i32, real_seconds)]."""
    g = torch.Generator().manual_seed(seed)
    n = B * n_buckets
    lens = torch.randint(int(lo_s * SR), int(hi_s * SR) + 1, (n,), generator=g).sort().values
    out = []
    for k in range(n_buckets):
        l = lens[k * B:(k + 1) * B]
        Lmax = int(l.max())
        wave = 0.1 * torch.randn(B, Lmax, generator=g)
        wave *= (torch.arange(Lmax).unsqueeze(0) < l.unsqueeze(1))
        tl = (l.float() / SR * 2.8).floor().int().clamp(min=1)
        tg = _targets(B, int(tl.max()), V, g)
        out.append((wave.to(device), l.int().to(device), tg.to(device), tl.to(device), float(l.sum()) / SR,
                    1.0 - float(l.sum()) / (B * Lmax)))
    return out


def source_id() -> str:
    """hash of the kernel sources: ties a PMC traffic file under profiles/ to the build it was measured on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lightning_asr_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/lasr.h"]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def step_algorithmic_bytes(config: str, B: int, audio_s_per_step: float) -> float:
    if config in STEP_MB_PER_UTT:
        return STEP_MB_PER_UTT[config] * 1e6 * B
    return STEP_MB_PER_AUDIO_S[config] * 1e6 * audio_s_per_step


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name: str, V: int):
    """The CPU oracle (oracle/ref_cpu.py, the pinned restatement of the reference's path) timed on this box's host cores
    (SURVEY 8d protocol): f32, features precomputed, fwd / CTC / bwd / NovoGrad timed separately, >= 3 warm-up + >= 5 timed
    steps on the SAME batch shape as the GPU workload where that fits ~30 s of CPU work (cfg2/cfg4: B=32 x 10 s), thread count
    chosen by a short sweep over the cores this process may use, CPU model string reported."""
    from oracle import ref_cpu as R
    cfg = CONFIGS[cfg_name]
    torch.manual_seed(0)
    if cfg["ragged"]:
        B, warm, steps = 8, 1, 3
        bk = synth_buckets(B, 3, V, 4321, "cpu")[1]            # the middle bucket of 24 clips: ~9 s clips
        wave, slens, tg, tl, real_s = bk[0], bk[1], bk[2], bk[3], bk[4]
        feats_l = [R.parse_wave(wave[i:i + 1, :int(slens[i])]) for i in range(B)]
        feats, _, pct, _ = R.collate(feats_l, [tg[i, :int(tl[i])].tolist() for i in range(B)])
        sample = "%d clips of one length bucket (%.1f s of audio, %.0f %% padding)" % (B, real_s, 100 * bk[5])
    else:
        B, warm, steps = 32, 3, 5
        wave, tg, tl = R.synth_batch(B, int(cfg["clip_s"] * SR), 100, V, 1234)
        # the CPU front-end beside it: the oracle's torch.stft restatement per clip, one thread (a DataLoader worker of
        # the reference, data_module.py:150-174) - not part of `value`, whose features are precomputed like the GPU's prefetch
        torch.set_num_threads(1)
        R.parse_wave(wave[0:1])
        t0 = time.perf_counter()
        for i in range(4):
            R.parse_wave(wave[i:i + 1])
        mel_s = (time.perf_counter() - t0) / 4
        feats = torch.stack([R.parse_wave(wave[i:i + 1])[0] for i in range(B)]).unsqueeze(1)
        pct = torch.ones(B)
        real_s = B * cfg["clip_s"]
        sample = "%d x %.0f s clips (the GPU workload's batch)" % (B, cfg["clip_s"])
    model = R.OracleModel(cfg["variant"], V + 1, mask=True, state=R.random_state(cfg["variant"], V + 1, 0))
    st = R.NovogradState(len(model.parameters()))
    params = model.parameters()

    def step():
        model.training = True
        model.requires_grad_(True)
        for p in params:
            p.grad = None
        t0 = time.perf_counter()
        lp = model.forward(feats, pct)
        t1 = time.perf_counter()
        loss = R.training_loss(lp, tg, pct, tl, blank=V)
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        R.novograd_step([p.data for p in params], [p.grad for p in params], st, 1e-4, 0.8, 0.5, 1e-8, 1e-3)
        t4 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2, t4 - t3

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    sweep = {}
    for n in sorted({min(avail, n) for n in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(n)
        step()
        sweep[n] = sum(step())
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    for _ in range(max(0, warm - 2)):
        step()
    parts = [step() for _ in range(steps)]
    dt = sum(sum(p) for p in parts) / steps
    names = ("fwd", "ctc_fwd", "bwd", "novograd")
    out_extra = {"mel_s_per_clip_1thread": mel_s} if not cfg["ragged"] else {}
    return {**out_extra, "value": real_s / dt, "unit": "audio-seconds/sec", "cores": best, "kind": "port",
            "sample": "%s, %d warm-up + %d timed steps of fwd+CTC+bwd+NovoGrad, f32, features precomputed" % (sample, warm, steps),
            "cpu_model": cpu_model(), "cores_available": avail, "s_per_step": dt,
            "split_s": {k: sum(p[i] for p in parts) / steps for i, k in enumerate(names)},
            "thread_sweep_s_per_step": {str(k): v for k, v in sweep.items()}}


def watchdog_sync(what: str, timeout_s: float = 180.0) -> None:
    """the first replay of a captured step that holds RCCL collectives must COMPLETE before anything is timed: a capture failure falls
    back to eager launches, a replay hang is caught here (exit code 3 with a message; never a re-exec) - step.sync_with_timeout"""
    from lightning_asr_amd.step import sync_with_timeout
    sync_with_timeout("bench.py: " + what, timeout_s)


def comm_record(ts, n_steps: int, use_graph: bool, buckets_per_step: int):
    """bench.py's `comm` object (N > 1, or LASR_FORCE_OVERLAP=1 on one GPU): which path carries the gradient exchange and, measured
    with events inside liblasr during the eager roofline steps (lasr_comm_timing), how long each bucket's all-reduce took on the side
    stream (peers' arrival included) and how long the optimiser's stream actually stalled at lasr_comm_wait - the exposed part."""
    rec = {"path": "lasr_comm (librccl called by liblasr on its own side stream)" if ts.comm is not None else "torch.distributed all_reduce",
           "timed_region_launch": "hipGraph replay (collectives inside the capture)" if use_graph else "eager",
           "staged_backward": bool(ts.overlap and (ts.world > 1 or ts.force_staged)), "buckets_per_step": buckets_per_step,
           "rccl_library": os.environ.get("LASR_RCCL_PATH", "librccl.so"),
           "max_channels": os.environ.get("NCCL_MAX_NCHANNELS"), "world": ts.world}
    if ts.comm is None:
        rec["note"] = "no per-bucket timing on the torch.distributed fallback path"
        return rec
    coll, waits = ts.comm.timing_collect()
    ts.comm.timing(False)
    if not coll or n_steps <= 0:
        return rec
    per_step = max(1, len(coll) // n_steps)
    buckets = []
    for k in range(per_step):
        xs = coll[k::per_step]
        us = sum(x[0] for x in xs) / len(xs)
        mb = xs[0][1] / 1e6
        buckets.append({"mb": mb, "allreduce_us": us, "algbw_gbs": mb * 1e3 / us if us > 0 else None})
    rec.update({"buckets": buckets, "exposed_wait_us_per_step": sum(waits) / n_steps if waits else None,
                "allreduce_us_per_step": sum(b["allreduce_us"] for b in buckets),
                "measured_over": "%d eager steps after the timed region (events on the side stream / the optimiser's stream)" % n_steps})
    return rec


def write_corpus(root: str, cfg: dict, labels, B: int, n_files_batches: int, n_steps: int, seed: int):
    """synthetic wav corpus (0.1 N(0,1) at 16 kHz, 16-bit PCM) + a JSON-lines manifest in the reference's format
    (scripts/get_libri.py:135) with n_steps * B lines cycling over n_files_batches * B files; one epoch = n_steps batches"""
    import wave as wavmod
    import numpy as np
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(seed)
    items = []
    for i in range(B * n_files_batches):
        secs = float(rng.uniform(2.0, 16.0)) if cfg["ragged"] else float(cfg["clip_s"])
        L = int(secs * SR)
        pcm = np.clip(0.1 * rng.standard_normal(L) * 32768, -32768, 32767).astype("<i2")
        path = os.path.join(root, "clip_%05d.wav" % i)
        with wavmod.open(path, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(SR); w.writeframes(pcm.tobytes())
        S = max(1, int(2.8 * secs)) if cfg["ragged"] else 100
        ids = rng.integers(0, len(labels), S)
        for k in range(1, S):                       # no adjacent repeats: CTC-feasible (SURVEY note N9)
            if ids[k] == ids[k - 1]:
                ids[k] = (ids[k] + 1) % len(labels)
        items.append({"audio_filepath": path, "duration": L / SR, "text": "".join(labels[j] for j in ids)})
    man = os.path.join(root, "train.json")
    with open(man, "w", encoding="utf-8") as f:
        for k in range(n_steps * B):
            f.write(json.dumps(items[k % len(items)], ensure_ascii=False) + "\n")
    dev_man = os.path.join(root, "dev.json")
    with open(dev_man, "w", encoding="utf-8") as f:
        for it in items[:B]:
            f.write(json.dumps(it, ensure_ascii=False) + "\n")
    return man, dev_man


def trainer_path(args, cfg, emit: bool = True, steps=None, warmup=None):
    """The metric through the reference's own surface: python -m lightning_asr_amd.train's objects (LibriDataModule, LightingModule,
    Trainer.fit) over a synthetic wav corpus.  Timed: K consecutive training steps of Trainer.fit after W warm-up steps, bracketed
    by barrier + synchronize - including wav decode (host threads), int16 H2D, random crop + SpecAugment draws, dither, and the
    per-step greedy decode + WER accumulation the reference logs (train.py:79-81)."""
    import tempfile
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is hand-written HIP with no CPU fallback")
    from lightning_asr_amd.data_module import LibriDataModule
    from lightning_asr_amd.lightning_compat import Trainer, rank_device_index, seed_everything
    local_rank = rank_device_index()         # (LASR_DIST_BACKEND=gloo: ranks share the devices round-robin - the one-GPU rehearsal)
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    from lightning_asr_amd.train import LightingModule
    labels = [c.strip() for c in open(os.path.join(ROOT, cfg["vocab"]), encoding="utf-8").readlines()]
    V, B = len(labels), args.batch
    W = args.warmup if warmup is None else warmup
    K = args.steps if steps is None else steps
    root = tempfile.mkdtemp(prefix="lasr_bench_rank%d_" % rank)
    # (N > 1: the DistributedSampler hands every rank 1 / world of the manifest it is given - this rank's own corpus here - so the
    #  manifest holds world x the batches one rank takes)
    man, dev_man = write_corpus(root, cfg, labels, B, 8, (W + K + 2) * world, 1234 + rank)
    seed_everything(0)
    act = torch.float32 if args.dtype == "f32" else torch.bfloat16
    crop = os.environ.get("LASR_BENCH_CROP", "1") != "0"
    dm = LibriDataModule([man], dev_man, dev_man, labels, train_bs=B, dev_bs=B, num_worker=args.ingest_threads, device=dev, act_dtype=act,
                         bucket_by_length=cfg["ragged"], bucket_batches=8, train_crop=crop)
    model = LightingModule(learning_rate=1e-2, weight_decay=1e-3, labels=labels, total_epoch=1, drop_rate=0.0, mask=True, use_cer=True,
                           variant=cfg["variant"], dtype=args.dtype, device=dev, warmup_steps=min(1000, (W + K) // 2))

    model.print = lambda *a: sys.stderr.write(" ".join(str(x) for x in a) + "\n")     # stdout carries the ONE JSON line only

    class Clock:
        t0 = t1 = None
        a0 = a1 = 0.0
        h0 = h1 = (0.0, 0.0, 0.0)

        def on_train_batch_end(self, tr):
            if tr.global_step in (W, W + K):
                if world > 1:
                    import torch.distributed as dist
                    dist.barrier()
                torch.cuda.synchronize()
                f_ = tr.fused
                snap = (f_.ingest_wait_s, f_.host_step_s, f_.host_cpu_s)
                if tr.global_step == W:
                    self.t0, self.a0, self.h0 = time.perf_counter(), f_.audio_seconds, snap
                else:
                    self.t1, self.a1, self.h1 = time.perf_counter(), f_.audio_seconds, snap
    clock = Clock()
    tr = Trainer(max_epochs=1, max_steps=W + K, default_root_dir=os.path.join(root, "run"), device=dev, check_val_every_n_epoch=1000,
                 callbacks=[clock], log_every_n_steps=50)
    tr.fit(model, dm)
    if tr.fused is None:
        raise SystemExit("Trainer.fit did not take the fused path")
    dt = clock.t1 - clock.t0
    audio = clock.a1 - clock.a0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    host_ms = {"waiting_for_ingest": 1e3 * (clock.h1[0] - clock.h0[0]) / K,      # (over the K timed steps)
               "enqueuing_the_step": 1e3 * (clock.h1[1] - clock.h0[1]) / K,
               "enqueue_cpu_time": 1e3 * (clock.h1[2] - clock.h0[2]) / K}
    by_rank = None
    if world > 1:          # every rank's host-side figures in rank 0's line (the 8-rank rehearsal: VERDICT r4 item 7)
        by_rank = [None] * world
        dist.all_gather_object(by_rank, dict(host_ms, rank=rank, graph_steps=tr.fused.graph_steps, eager_steps=tr.fused.eager_steps,
                                             ingest_threads=getattr(tr.fused, "ingest_threads", None)))
    import shutil
    shutil.rmtree(root, ignore_errors=True)
    if world > 1 and emit:
        tr.teardown()
    if rank != 0:
        return
    ms_per_step = 1e3 * dt / K
    step_bytes = step_algorithmic_bytes(args.config, B, audio / K)
    step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
    f = tr.fused
    out = {
        "metric": "audio-seconds/sec training (asr13x1, bs=32, 10 s clips)" if args.config == "cfg2" else "audio-seconds/sec training (%s)" % args.config,
        "value": audio * world / dt, "unit": "audio-seconds/sec", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (cfg["workload"] % B) + ", through LibriDataModule + LightingModule + Trainer.fit (the reference's surface)",
                   "name": args.config, "path": "trainer", "global_batch": B * world, "n_class": V + 1, "parallelism": "dp%d" % world,
                   "audio_seconds_per_step_per_gpu": audio / K, "padding_frac": 1.0 - f.samples_real / max(f.samples_padded, 1),
                   "ingest": f.source_kind, "ingest_threads": getattr(f, "ingest_threads", None) or args.ingest_threads, "train_crop": crop,
                   "hip_graph_steps": f.graph_steps, "eager_steps": f.eager_steps, "lean_head": bool(f.native.lean_head),
                   "host_ms_per_step_by_rank": by_rank,
                   "host_ms_per_step": {**host_ms,
                                        "note": "enqueuing_the_step is wall time inside FusedLoop.step: once the GPU's queue is full the runtime "
                                                "makes the thread wait, so a GPU-bound loop reads ~the step time there; enqueue_cpu_time is the "
                                                "thread's CPU time (time.thread_time) over the same calls"},
                   "included": "wav decode on host threads, int16 H2D, random sub-sequence crop + SpecAugment draws, in-kernel dither, "
                               "per-step greedy decode + edit distance + loss/WER accumulation (train.py:79-81), all of TrainStep",
                   "excluded": "validation, checkpoint writes (epoch-end work; the timed steps sit inside one epoch)"},
        "final_loss": tr.history[-1].get("train_loss") if tr.history else None,
        "roofline": {"bound": "hbm", "achieved": step_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": step_gbs / PEAK_HBM_GBS,
                     "traffic": None, "kernel": "whole step (SURVEY 8d algorithmic bytes: %.1f MB)" % (step_bytes / 1e6),
                     "step_frac": step_gbs / PEAK_HBM_GBS},
    }
    if emit:
        print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 10 warm-up, >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=os.environ.get("LASR_BENCH_CONFIG", "cfg2"), choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default=os.environ.get("LASR_BENCH_DTYPE", "bf16"), choices=["f32", "bf16"],
                    help="activation dtype: bf16 (BASELINE config) or f32 (exact parity mode)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", dest="graph", action="store_true", default=os.environ.get("LASR_BENCH_GRAPH", "1") == "1",
                    help="replay the step from a captured hipGraph (default for one GPU; LASR_BENCH_GRAPH=0 / --no-graph: eager launches)")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="compute each step's features at the head of the step instead of inside the previous step's CTC launch")
    ap.add_argument("--path", default=os.environ.get("LASR_BENCH_PATH", "step"), choices=["step", "trainer"],
                    help="step: TrainStep on batches resident in HBM (default); trainer: manifest -> LibriDataModule -> Trainer.fit")
    ap.add_argument("--ingest-threads", type=int, default=8, help="--path trainer: host threads decoding wav files (data.num_worker)")
    ap.add_argument("--no-trainer-record", dest="trainer_record", action="store_false",
                    help="skip the `trainer` sub-record (60 + 30 steps through Trainer.fit) of the default one-GPU line")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    # N > 1 from the plain command (`python bench.py --gpus N`, no outer launcher) or under torch.distributed.run: a supervisor that
    # never touches the GPU starts FRESH worker processes, one per GPU, relays rank 0's JSON line (plus a `launcher` record: which
    # rung of graph + lasr_comm -> eager + lasr_comm -> eager + torch.distributed ran) and exits with their code - launch.py
    from lightning_asr_amd import launch
    if launch.role(args.gpus) == "parent" and os.environ.get("LASR_BENCH_BACKEND", "nccl") == "nccl" and \
            os.environ.get("LASR_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < args.gpus:
        # (device_count() does not initialise the GPU; the rehearsal backends share devices round-robin and are exempt)
        raise SystemExit("bench.py --gpus %d: this host shows %d GPU(s) - one process per GPU over RCCL needs %d"
                         % (args.gpus, torch.cuda.device_count(), args.gpus))
    rc = launch.maybe_launch(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    if args.path == "trainer":
        return trainer_path(args, cfg)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is hand-written HIP with no CPU fallback")
    # rehearsal of the N > 1 path on a box with fewer GPUs than ranks: LASR_BENCH_BACKEND=gloo shares the devices round-robin
    # (RCCL refuses two ranks on one device); the driver's runs use one GPU per rank over nccl = RCCL
    backend = os.environ.get("LASR_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if args.gpus > 1 or world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl backend IS RCCL on ROCm
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()
    else:
        dist = None

    from lightning_asr_amd import _lib
    from lightning_asr_amd.engine import NativeModel
    from lightning_asr_amd.schedule import CosineAnnealingWarmupRestarts
    from lightning_asr_amd.step import GraphedTrainStep, TrainStep

    V = vocab_size(cfg["vocab"])
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    model = NativeModel(cfg["variant"], V + 1, mask=True, act="relu", dtype=dtype, device=dev)
    model.init_parameters(seed=0)                               # pl.seed_everything(0), train.py:203
    sched = CosineAnnealingWarmupRestarts(None, first_cycle_steps=100 * 1000, cycle_mult=2, max_lr=1e-2, min_lr=1e-4,
                                          warmup_steps=1000, gamma=0.5)
    comm1 = None
    if world == 1 and os.environ.get("LASR_FORCE_OVERLAP") == "1":     # measure the staged (N > 1) form of the step on one GPU: 1-rank RCCL communicator
        from lightning_asr_amd.comm import Communicator
        comm1 = Communicator.single(dev)
    ts = TrainStep(model, 1e-2, 1e-3, schedule=sched, comm=comm1)
    ts.broadcast_parameters()
    B = args.batch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Synthetic batches, resident in HBM.  Every step computes the log-mel features of exactly one batch: with the
    # prefetch (default) those of the NEXT step's waveforms, inside this step's CTC launch (lattice + feature workgroups in one
    # grid - the data-loader prefetch of the reference's workers, on the GPU).  Same results either way.
    if cfg["ragged"]:
        batches = synth_buckets(B, 8, V, 1234 + rank, dev)
        order = torch.randperm(len(batches), generator=torch.Generator().manual_seed(7)).tolist()   # bucket order as a sampler would shuffle it
        batches = [batches[i] for i in order]
        Lmax = max(b[0].shape[1] for b in batches)
        model.workspace(B, int(_lib.load().lasr_mel_num_frames(Lmax)), max(b[2].shape[1] for b in batches))   # size it once for the largest bucket
    else:
        n = int(cfg["clip_s"] * SR)
        batches = []
        for seed in (1234 + rank, 991234 + rank):
            w, t_, l_ = synth_batch(B, n, 100, seed, dev, V)
            batches.append((w, None, t_, l_, B * cfg["clip_s"], 0.0))
    step_no = [0]
    audio_s = [0.0]
    from lightning_asr_amd.step import graph_dp_enabled
    use_graph = args.graph and (world == 1 or graph_dp_enabled())   # N > 1: the staged step with its RCCL all-reduces is captured too (LASR_GRAPH_DP=0: eager)
    graphs = {}
    if use_graph:
        # hipGraph replay: the ~200 launches of a step become one graph launch (host-enqueue time no longer bounds the step).  The
        # graphs read the resident synthetic batches IN PLACE: nothing is copied per step.
        try:
            if cfg["ragged"] and args.prefetch:
                # one graph per bucket; graph i trains on bucket i's features (computed by the previous replay) and computes bucket
                # i+1's inside its loss launch, into that bucket's own feature buffer: nothing is copied between replays
                nb_ = len(batches)
                FP = []
                for bk in batches:
                    f_, p_ = ts.features(bk[0], bk[1])
                    FP.append((f_.clone(), p_.clone()))
                for i, bk in enumerate(batches):
                    nx = batches[(i + 1) % nb_]
                    graphs[i] = GraphedTrainStep(ts, B, nx[0].shape[1], bk[2].shape[1], ragged=True, prefetch=True,
                                                 inputs=(nx[0], nx[1], bk[2], bk[3]), feats_in=FP[i], feats_out=FP[(i + 1) % nb_])
                    graphs[i].capture()
                f_, p_ = ts.features(batches[0][0], batches[0][1])     # (the captures' warm-up passes ran through the buffers)
                FP[0][0].copy_(f_); FP[0][1].copy_(p_)
            elif cfg["ragged"]:                          # --no-prefetch: features + step of the same bucket in one replay
                for i, bk in enumerate(batches):
                    graphs[i] = GraphedTrainStep(ts, B, bk[0].shape[1], bk[2].shape[1], ragged=True, prefetch=False,
                                                 inputs=(bk[0], bk[1], bk[2], bk[3]))
                    graphs[i].capture()
            elif args.prefetch:                          # two graphs ping-pong the feature buffers: A trains on F0 and writes F1, B the reverse
                F0, p0 = ts.features(batches[0][0])
                F0, p0 = F0.clone(), p0.clone()
                F1, p1 = torch.empty_like(F0), torch.empty_like(p0)
                for i in range(2):
                    cur, nxt = batches[i], batches[1 - i]
                    graphs[i] = GraphedTrainStep(ts, B, cur[0].shape[1], cur[2].shape[1], prefetch=True, inputs=(nxt[0], None, cur[2], cur[3]),
                                                 feats_in=(F0, p0) if i == 0 else (F1, p1), feats_out=(F1, p1) if i == 0 else (F0, p0))
                    graphs[i].capture()
                f, p_ = ts.features(batches[0][0])       # (the captures' warm-up passes ran through the buffers: re-prime step 0's features)
                F0.copy_(f); p0.copy_(p_)
            else:
                for i, bk in enumerate(batches):
                    graphs[i] = GraphedTrainStep(ts, B, bk[0].shape[1], bk[2].shape[1], prefetch=False, inputs=(bk[0], None, bk[2], bk[3]))
                    graphs[i].capture()
        except Exception as e:                           # a box whose runtime refuses the capture: eager launches, said in the JSON
            sys.stderr.write("graph capture failed (%s): falling back to eager launches\n" % (e,))
            use_graph, graphs = False, {}
            torch.cuda.synchronize()
        if dist is not None:                             # every rank replays, or none does (a lone eager rank would issue its collectives differently)
            okf = torch.tensor([1 if use_graph else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if int(okf.item()) == 0 and use_graph:
                sys.stderr.write("another rank could not capture: eager launches on all ranks\n")
                use_graph, graphs = False, {}

    def one_step(eager=False):
        i = step_no[0]
        w, sl, t_, l_, secs, _pad = batches[i % len(batches)]
        nxt = batches[(i + 1) % len(batches)] if args.prefetch else None
        step_no[0] += 1
        audio_s[0] += secs
        if eager or not use_graph:
            return ts.step(w, t_, l_, sample_lens=sl, prefetch_wave=None if nxt is None else nxt[0],
                           prefetch_lens=None if nxt is None else nxt[1], want_logp=False)   # the training step reads loss + argmax only
        return graphs[i % len(batches)].replay()

    for k_ in range(args.warmup):
        loss, *_ = one_step()
        if k_ == 0 and use_graph and (world > 1 or ts.force_staged):
            watchdog_sync("the first replay of the captured data-parallel step")
    if args.warmup == 0 and use_graph and (world > 1 or ts.force_staged):
        loss, *_ = one_step()                      # (never time a graph with collectives that has not completed once)
        watchdog_sync("the first replay of the captured data-parallel step")
    barrier()
    audio_s[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, *_ = one_step()
    barrier()
    dt = time.perf_counter() - t0
    timed_audio_s = audio_s[0]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    ms_per_step = 1e3 * dt / args.steps
    value = timed_audio_s * world / dt

    # ---- roofline leg: the same steps again with HIP events around every launch of the dominant
    # kernel class (the 1x1-conv GEMMs), recorded on the launch stream inside liblasr.
    # Every rank runs these steps (they contain the gradient all-reduce); only rank 0 brackets its launches with events.
    lib = _lib.load()
    roofline = None
    n_prof = max(2, min(args.steps, 5))
    if cfg["ragged"]:
        n_prof = len(batches)
    if rank == 0:
        lib.lasr_prof_enable(1)
        if ts.comm is not None:
            ts.comm.timing(True)
    for _ in range(n_prof):
        one_step(eager=True)       # (the in-library event brackets live in the launch path: a graph replay does not pass through it)
    torch.cuda.synchronize()
    comm_rec = None
    if rank == 0 and (world > 1 or ts.force_staged):
        comm_rec = comm_record(ts, n_prof, use_graph, int(os.environ.get("LASR_DP_BUCKETS", "2")))
    if rank == 0:
        lib.lasr_prof_enable(0)
        NK = 8
        ms = (C.c_double * NK)(); fl = (C.c_double * NK)(); by = (C.c_double * NK)(); cnt = (C.c_int64 * NK)()
        _lib.check(lib.lasr_prof_collect(ms, fl, by, cnt), "lasr_prof_collect")
        gemm_ms, gemm_fl, gemm_by, gemm_n = ms[0], fl[0], by[0], cnt[0]
        # Every bracketed launch carries its two event packets (measured below with empty pairs: ~5 us).  The per-launch time is
        # quoted RAW - rocprofv3's kernel trace of the same command puts the truth between raw and raw - overhead, so the raw
        # figure is the conservative one (round 2 subtracted the overhead and came out above the trace).
        ovh = C.c_double(0.0)
        _lib.check(lib.lasr_prof_overhead_ms(C.c_void_p(torch.cuda.current_stream().cuda_stream), 512, C.byref(ovh)),
                   "lasr_prof_overhead_ms")
        # HBM traffic of the same kernel class from rocprofv3 --pmc passes (tools/pmc_pass.sh): only quoted when the file was
        # measured on THIS build of the kernels (source hash) and this config
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic_%s_%s.json" % (args.config, args.dtype))
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("source_id") == source_id():
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), os.path.relpath(tpath, ROOT)
            else:
                traffic_src = "stale: %s was measured on build %s, this is %s" % (os.path.relpath(tpath, ROOT), tj.get("source_id"), source_id())
        # whole-step traffic at the L2s' memory side (every kernel; same two PMC passes, tools/pmc_traffic.py::whole_step)
        step_traffic, step_traffic_src = None, None
        spath = os.path.join(ROOT, "profiles", "step_traffic_%s_%s.json" % (args.config, args.dtype))
        if os.path.exists(spath):
            sj = json.load(open(spath))
            if sj.get("source_id") == source_id():
                step_traffic, step_traffic_src = sj.get("traffic_mb_per_step"), os.path.relpath(spath, ROOT)
            else:
                step_traffic_src = "stale: %s was measured on build %s, this is %s" % (os.path.relpath(spath, ROOT), sj.get("source_id"), source_id())
        if args.dtype == "f32":
            ach = gemm_fl / (gemm_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TF}
        else:
            ach = gemm_by / (gemm_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
        # the whole step against the HBM roofline: SURVEY 8(d)'s algorithmic bytes of the step / measured step time / 8 TB/s
        step_bytes = step_algorithmic_bytes(args.config, B, timed_audio_s / args.steps)
        step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
        classes = {}
        for k, name in enumerate(PROF_KINDS):
            if cnt[k]:
                t_ms = ms[k] / n_prof
                classes[name] = {"ms_per_step": t_ms, "brackets_per_step": cnt[k] / n_prof, "algorithmic_mb_per_step": by[k] / n_prof / 1e6,
                                 "hbm_frac": (by[k] / (ms[k] * 1e-3) / 1e9 / PEAK_HBM_GBS) if by[k] and ms[k] else None}
        roofline.update({"traffic": traffic, "traffic_source": traffic_src, "kernel": "gemm (1x1 conv fwd/dgrad/wgrad)",
                         "launches_per_step": gemm_n // n_prof, "avg_launch_us": 1e3 * gemm_ms / max(gemm_n, 1),
                         "event_overhead_us": 1e3 * ovh.value, "gemm_ms_per_step": gemm_ms / n_prof,
                         "algorithmic_gflop_per_step": gemm_fl / n_prof / 1e9, "algorithmic_mb_per_step": gemm_by / n_prof / 1e6,
                         "mfma_frac": gemm_fl / (gemm_ms * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TF if args.dtype == "bf16" else PEAK_F32_MFMA_TF),
                         "step_frac": step_gbs / PEAK_HBM_GBS, "step_achieved_gbs": step_gbs, "step_algorithmic_mb": step_bytes / 1e6,
                         "step_traffic_mb": step_traffic, "step_traffic_source": step_traffic_src,
                         "classes": classes,
                         "classes_note": "HIP-event brackets inside liblasr during %d extra eager steps (times include ~event_overhead_us "
                                         "per bracket; a BN bracket spans the 2-3 launches of a unit's pass)" % n_prof,
                         "source_id": source_id()})
    if dist is not None:
        dist.barrier()

    # ---- the drop-in path in the same record: K = 30 steps (after 60 warm-up steps) of the SAME metric through the reference's own surface (manifest ->
    # LibriDataModule -> LightingModule -> Trainer.fit; wav decode, int16 H2D, random crop + SpecAugment, dither, per-step decode + WER
    # inside the timed region).  One GPU only: Trainer owns its process group.  ~5 s including the synthetic corpus.
    trainer_rec = None
    if world == 1 and dist is None and args.trainer_record and not args.no_cpu_baseline:      # (--no-cpu-baseline = the lean run the profiling scripts use)
        del graphs
        torch.cuda.synchronize()
        try:
            # (60 warm-up steps: while the ingest threads are still filling their ring for the first time they compete with the
            #  enqueuing thread for the interpreter - the same run read 2.25 ms per step with 5 warm-up steps and 2.16 with 25 or 60 -
            #  and under the random crop the two common batch shapes are captured into hipGraphs within the first ~25 +- 13 steps)
            tr_out = trainer_path(args, cfg, emit=False, steps=30, warmup=60)
            c_ = tr_out["config"]
            trainer_rec = {"what": "the same metric through LibriDataModule + LightingModule + Trainer.fit (bench.py --path trainer), 60 warm-up + 30 timed steps",
                           "value": tr_out["value"], "unit": tr_out["unit"], "ms_per_step": tr_out["ms_per_step"], "steps": 30, "warmup": 60,
                           "vs_step_line": tr_out["ms_per_step"] / ms_per_step, "host_ms_per_step": c_["host_ms_per_step"],
                           "hip_graph_steps": c_["hip_graph_steps"], "eager_steps": c_["eager_steps"], "train_crop": c_["train_crop"],
                           "ingest": c_["ingest"], "ingest_threads": c_["ingest_threads"], "padding_frac": c_["padding_frac"],
                           "included": c_["included"], "final_loss": tr_out["final_loss"]}
        except Exception as e:      # the headline number above stands on its own; say why the sub-record is missing
            trainer_rec = {"error": "%s: %s" % (type(e).__name__, e)}

    if dist is not None:
        dist.destroy_process_group()        # (rank 0 goes on alone from here: nobody waits in a collective for its CPU baseline)
        dist = None
    if rank != 0:
        return
    metric = "audio-seconds/sec training (asr13x1, bs=32, 10 s clips)" if args.config == "cfg2" else \
        "audio-seconds/sec training (%s)" % args.config
    out = {
        "metric": metric, "value": value, "unit": "audio-seconds/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (cfg["workload"] % B) + ", HIP mel+conv+CTC+NovoGrad, random-init weights", "name": args.config,
                   "global_batch": B * world, "clip_seconds": cfg["clip_s"] if cfg["clip_s"] else "2-16 (ragged)",
                   "audio_seconds_per_step_per_gpu": timed_audio_s / args.steps,
                   "padding_frac": sum(b[5] for b in batches) / len(batches),
                   "n_class": V + 1, "parallelism": "dp%d" % world, "feature_prefetch": bool(args.prefetch), "hip_graph": bool(use_graph),
                   "staged_backward": bool(ts.overlap and (ts.world > 1 or ts.force_staged)),
                   "grad_exchange": None if world == 1 else ("lasr_comm (librccl on the library's side stream, %d buckets overlapped with backward)"
                                                             % int(os.environ.get("LASR_DP_BUCKETS", "2")) if ts.comm is not None
                                                             else "torch.distributed all_reduce"),
                   "excluded": "H2D of the PCM (waves resident in HBM); the reference's per-step greedy decode + WER logging (train.py:80)"},
        "final_loss": final_loss,
        "roofline": roofline,
    }
    if comm_rec is not None:
        out["comm"] = comm_rec
    if not args.no_cpu_baseline:
        # rank 0 only; at N > 1 AFTER the process group is gone and the other ranks have left the GPU box's cores alone
        out["cpu_baseline"] = cpu_baseline(args.config, V)
    if launch.rung_info() is not None:
        out["config"]["launch_rung"] = launch.rung_info()["rung_index"]
    if trainer_rec is not None:
        out["trainer"] = trainer_rec
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
