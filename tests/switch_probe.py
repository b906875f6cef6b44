"""Helper of tests/test_gpu_switches.py: one bf16 training step (and one eval forward) of the plan under whatever LASR_* switches the
environment carries; prints a JSON line with the loss, the gradient norms per parameter group and an eval checksum."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lightning_asr_amd.engine import NativeModel  # noqa: E402
from lightning_asr_amd.step import TrainStep  # noqa: E402


def main():
    import ctypes as C
    from lightning_asr_amd import _lib
    variant, n_class = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda:0")
    B, L, S = 24, 64000, 20                      # 24 x 201 = 4824 rows: past the 4096-row threshold of the sliced BN backward
    ragged = False
    if len(sys.argv) > 5:                        # B L ragged: e.g. 8 160000 1 -> T' = 501, the half-tile depthwise forward's shape class
        B, L, ragged = int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5]))
    act = sys.argv[6] if len(sys.argv) > 6 else "relu"
    wave, tg, tl = bench.synth_batch(B, L, S, 77, dev, n_class - 1)
    nxt, _, _ = bench.synth_batch(B, L, S, 78, dev, n_class - 1)
    sl = None
    if ragged:                                   # utterances of 100 %, 93 %, ... of the batch's length: MaskCNN rows + padded frames in play
        sl = torch.tensor([L - (i % 5) * (L // 14) for i in range(B)], dtype=torch.int32, device=dev)
        wave = wave * (torch.arange(L, device=dev).unsqueeze(0) < sl.unsqueeze(1))
    m = NativeModel(variant, n_class, mask=True, act=act, dtype=torch.bfloat16, device=dev)
    m.init_parameters(seed=1)
    ts = TrainStep(m, 1e-2, 1e-3)
    lib = _lib.load()
    lib.lasr_prof_enable(1)                      # bracket counts per kernel class: which launches this build of the step made
    loss, nll, logp, am = ts.step(wave, tg, tl, sample_lens=sl, prefetch_wave=nxt, want_logp=False)
    torch.cuda.synchronize()
    lib.lasr_prof_enable(0)
    ms = (C.c_double * 8)(); fl = (C.c_double * 8)(); by = (C.c_double * 8)(); cnt = (C.c_int64 * 8)()
    _lib.check(lib.lasr_prof_collect(ms, fl, by, cnt), "lasr_prof_collect")
    g = m.grads.double()
    groups = {}
    for t in m.param_infos():
        key = t.name.split(".")[1] if t.name.startswith("encoder.") else t.name.split(".")[0]
        groups[key] = groups.get(key, 0.0) + float((g[t.offset:t.offset + t.numel] ** 2).sum())
    feats, pct = ts.features(wave, sl)
    lp, _ = m.forward(feats, pct, training=False)
    print(json.dumps({"prof_brackets": {k: int(cnt[i]) for i, k in enumerate(bench.PROF_KINDS)}, "out_checksum": float(m.tap("block5").double().abs().sum()),
                      "loss": float(loss), "grad_norm": {k: v ** 0.5 for k, v in groups.items()}, "eval_checksum": float(lp.double().abs().mean()),
                      "params_after": float(m.params.double().abs().sum())}))


if __name__ == "__main__":
    main()
