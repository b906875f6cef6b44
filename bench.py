#!/usr/bin/env python
"""Throughput of the QuartzNet-CTC training hot path on MI355X (BASELINE.json metric):
audio-seconds/sec of training, asr13x1, bs=32/GPU, 10 s 16 kHz synthetic clips.

A "step" is one full pass of the hot path over one batch already resident in HBM:
    wave -> log-mel -> forward -> mean CTC -> backward -> grad all-reduce (N>1) -> NovoGrad -> LR step.
By default the log-mel stage is software-pipelined across steps like a data-loader prefetch: step i computes the
features of step i+1's waveforms inside its CTC launch (one batch of features per step either way; --no-prefetch
puts them back at the head of the step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|bf16]
N>1 is launched by the driver as ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIP_S = 10.0
SR = 16000
V = 27                       # data/labels.txt
S_TGT = 100                  # target length for 10 s clips
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_MFMA_TF = 157.3     # dense f32-input MFMA
PEAK_BF16_MFMA_TF = 2500.0   # dense bf16 MFMA


def synth_batch(B: int, n_samples: int, S: int, seed: int, device):
    """wave = 0.1 N(0,1); targets U{0..V-1} without adjacent repeats (always CTC-feasible)."""
    g = torch.Generator().manual_seed(seed)
    wave = 0.1 * torch.randn(B, n_samples, generator=g)
    tg = torch.randint(0, V, (B, S), generator=g)
    for s in range(1, S):
        same = tg[:, s] == tg[:, s - 1]
        tg[same, s] = (tg[same, s] + 1) % V
    return wave.to(device), tg.long().to(device), torch.full((B,), S, dtype=torch.int32, device=device)


def cpu_baseline(n_clips: int = 8, steps: int = 3):
    """The CPU oracle (oracle/ref_cpu.py, a port of the reference's path) timed on this box's host
    cores on a bounded sample of the same workload: n_clips x 10 s clips, features precomputed,
    fwd + CTC + bwd + NovoGrad."""
    from oracle import ref_cpu as R
    torch.manual_seed(0)
    wave, tg, tl = R.synth_batch(n_clips, int(CLIP_S * SR), S_TGT, V, 1234)
    feats = torch.stack([R.parse_wave(wave[i:i + 1])[0] for i in range(n_clips)]).unsqueeze(1)
    pct = torch.ones(n_clips)
    model = R.OracleModel("plain", V + 1, mask=True, state=R.random_state("plain", V + 1, 0))
    st = R.NovogradState(len(model.parameters()))
    R.train_step(model, st, feats, tg, pct, tl, 1e-4)          # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        R.train_step(model, st, feats, tg, pct, tl, 1e-4)
    dt = (time.perf_counter() - t0) / steps
    return {"value": n_clips * CLIP_S / dt, "unit": "audio-seconds/sec", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "%d x 10 s clips, %d steps of fwd+CTC+bwd+NovoGrad, f32, features precomputed"
                                      % (n_clips, steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 10 warm-up, >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default=os.environ.get("LASR_BENCH_DTYPE", "bf16"), choices=["f32", "bf16"],
                    help="activation dtype: bf16 (BASELINE config) or f32 (exact parity mode)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="compute each step's features at the head of the step instead of inside the previous step's CTC launch")
    args = ap.parse_args()

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
    from lightning_asr_amd.step import TrainStep

    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    model = NativeModel("plain", V + 1, mask=True, act="relu", dtype=dtype, device=dev)
    model.init_parameters(seed=0)                               # pl.seed_everything(0), train.py:203
    sched = CosineAnnealingWarmupRestarts(None, first_cycle_steps=100 * 1000, cycle_mult=2, max_lr=1e-2, min_lr=1e-4,
                                          warmup_steps=1000, gamma=0.5)
    ts = TrainStep(model, 1e-2, 1e-3, schedule=sched)
    ts.broadcast_parameters()
    B = args.batch
    wave, tg, tl = synth_batch(B, int(CLIP_S * SR), S_TGT, 1234 + rank, dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Two alternating synthetic batches.  Every step computes the log-mel features of exactly one batch: with the prefetch
    # (default) those of the NEXT step's waveforms, inside this step's CTC launch (lattice + feature workgroups in one grid -
    # the data-loader prefetch of the reference's workers, on the GPU); the features a step trains on were produced by the
    # step before it.  --no-prefetch computes them at the head of the step instead.  Same results either way.
    batches = [(wave, tg, tl), synth_batch(B, int(CLIP_S * SR), S_TGT, 991234 + rank, dev)]
    step_no = [0]

    def one_step():
        w, t_, l_ = batches[step_no[0] & 1]
        nxt = batches[(step_no[0] + 1) & 1][0] if args.prefetch else None
        step_no[0] += 1
        return ts.step(w, t_, l_, prefetch_wave=nxt)

    for _ in range(args.warmup):
        loss, *_ = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, *_ = one_step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    ms_per_step = 1e3 * dt / args.steps
    value = B * world * CLIP_S * args.steps / dt

    # ---- roofline leg: the same steps again with HIP events around every launch of the dominant
    # kernel class (the 1x1-conv GEMMs), recorded on the launch stream inside liblasr.
    # Every rank runs these steps (they contain the gradient all-reduce); only rank 0 brackets its launches with events.
    lib = _lib.load()
    roofline = None
    n_prof = max(2, min(args.steps, 5))
    if rank == 0:
        lib.lasr_prof_enable(1)
    for _ in range(n_prof):
        one_step()
    torch.cuda.synchronize()
    if rank == 0:
        lib.lasr_prof_enable(0)
        ms = (C.c_double * 4)(); fl = (C.c_double * 4)(); by = (C.c_double * 4)(); cnt = (C.c_int64 * 4)()
        _lib.check(lib.lasr_prof_collect(ms, fl, by, cnt), "lasr_prof_collect")
        gemm_ms_raw, gemm_fl, gemm_by, gemm_n = ms[0], fl[0], by[0], cnt[0]
        # every bracketed launch carries the cost of its two event packets: measure it (empty pairs on the same
        # stream) and take it out, so the per-launch time is the kernel's own (agrees with rocprofv3's trace)
        ovh = C.c_double(0.0)
        _lib.check(lib.lasr_prof_overhead_ms(C.c_void_p(torch.cuda.current_stream().cuda_stream), 512, C.byref(ovh)),
                   "lasr_prof_overhead_ms")
        gemm_ms = max(gemm_ms_raw - gemm_n * ovh.value, 0.5 * gemm_ms_raw)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic_%s.json" % args.dtype)
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        if args.dtype == "f32":
            ach = gemm_fl / (gemm_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TF}
        else:
            ach = gemm_by / (gemm_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}
        roofline.update({"traffic": traffic, "kernel": "gemm (1x1 conv fwd/dgrad/wgrad)", "launches_per_step": gemm_n // n_prof,
                         "avg_launch_us": 1e3 * gemm_ms / max(gemm_n, 1), "avg_launch_us_raw": 1e3 * gemm_ms_raw / max(gemm_n, 1),
                         "event_overhead_us": 1e3 * ovh.value, "gemm_ms_per_step": gemm_ms / n_prof,
                         "algorithmic_gflop_per_step": gemm_fl / n_prof / 1e9, "algorithmic_mb_per_step": gemm_by / n_prof / 1e6,
                         "dwconv_ms_per_step": ms[1] / n_prof})
    if dist is not None:
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    out = {
        "metric": "audio-seconds/sec training (asr13x1, bs=32, 10 s clips)", "value": value, "unit": "audio-seconds/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "asr13x1 QuartzNet-CTC train step, bs=%d/GPU, 10 s synthetic 16 kHz clips, labels.txt vocab "
                               "(C=28), HIP mel+conv+CTC+NovoGrad, random-init weights" % B,
                   "global_batch": B * world, "clip_seconds": CLIP_S, "target_len": S_TGT, "parallelism": "dp%d" % world,
                   "feature_prefetch": bool(args.prefetch)},
        "final_loss": final_loss,
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
