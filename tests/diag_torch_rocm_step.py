"""Yardstick (run by hand on the GPU box, not collected by pytest): the reference's OWN stack on the same MI355X.

The reference trains through PyTorch eager + native AMP (train.py:64-86, conf/conf.yaml:27-29).  Its Python cannot travel to the
GPU box, but the oracle (oracle/ref_cpu.py) is its pinned plain-PyTorch restatement, so moved to `cuda` it issues what the
reference would issue on this hardware: MIOpen / hipBLASLt convolutions, torch's batch_norm, the CUDA-path `ctc_loss`, and the
per-tensor NovoGrad loop of scheduler/novograd.py:75-145 - under `torch.autocast` (bf16, or fp16 with a GradScaler as the
reference configures it).  Prints the step time of the BASELINE workload (cfg2: asr13x1, 32 x 10 s clips, C = 28), features
precomputed as in `bench.py`.  Nothing here is used by the product or by `bench.py`'s `value`.

usage: python tests/diag_torch_rocm_step.py [--dtype bf16|fp16|f32] [--steps 20] [--variant plain|context_se] [--no-tune]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_cpu as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "f32"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variant", default="plain")
    ap.add_argument("--n-vocab", type=int, default=27, help="labels (the blank is added): 27 = labels.txt, 4333 = the AISHELL vocabulary")
    ap.add_argument("--clip-s", type=float, default=10.0)
    ap.add_argument("--no-tune", action="store_true", help="MIOpen immediate mode instead of its kernel search (minutes per dtype)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, clip_s, V = 32, a.clip_s, a.n_vocab
    torch.manual_seed(0)
    wave, tg, tl = R.synth_batch(B, int(clip_s * 16000), int(10 * clip_s) if V == 27 else int(2.8 * clip_s), V, 1234)
    feats = torch.stack([R.parse_wave(wave[i:i + 1])[0] for i in range(B)]).unsqueeze(1).to(dev)
    pct, tg, tl = torch.ones(B, device=dev), tg.to(dev), tl.to(dev)
    state = {k: v.to(dev) for k, v in R.random_state(a.variant, V + 1, 0).items()}
    model = R.OracleModel(a.variant, V + 1, mask=True, state=state)
    params = model.parameters()
    st = R.NovogradState(len(params))
    amp = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": None}[a.dtype]
    scale = [65536.0] if a.dtype == "fp16" else None      # dynamic loss scale of native AMP (GradScaler's rule: halve on overflow)
    torch.backends.cudnn.benchmark = not a.no_tune   # MIOpen picks its kernels by measurement, as a tuned run of the reference would

    def step():
        model.training = True
        model.requires_grad_(True)
        for p in params:
            p.grad = None
        with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
            lp = model.forward(feats, pct)
        loss = R.training_loss(lp.float(), tg, pct, tl, blank=V)
        if scale is not None:
            (loss * scale[0]).backward()
            grads = [p.grad / scale[0] for p in params]
            if not all(bool(torch.isfinite(g).all()) for g in grads):       # overflow: skip the step, halve the scale
                scale[0] *= 0.5
                return loss
        else:
            loss.backward()
            grads = [p.grad for p in params]
        with torch.no_grad():
            R.novograd_step([p.data for p in params], grads, st, 1e-4, 0.8, 0.5, 1e-8, 1e-3)
        return loss

    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps({"yardstick": "oracle (plain PyTorch restatement of the reference) on cuda through PyTorch-ROCm eager",
                      "variant": a.variant, "n_class": V + 1, "autocast": a.dtype, "B": B, "clip_s": clip_s, "ms_per_step": round(ms, 3),
                      "audio_s_per_s": round(B * clip_s / (ms * 1e-3), 1), "loss": float(loss.detach()), "miopen_search": not a.no_tune, "torch": torch.__version__,
                      "device": torch.cuda.get_device_name(0)}))


if __name__ == "__main__":
    main()
