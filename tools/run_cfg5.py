"""BASELINE config 5 shape class on one GPU: AISHELL char vocab (C=4334), variable-length 2-16 s clips
in one length bucket, bf16, full train steps.  Checks that the large-C / large-T path runs and reports
its throughput (dev tool; the driver's bench is cfg2)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd.engine import NativeModel
from lightning_asr_amd.step import TrainStep
dev = torch.device('cuda')
V = 4333
B = 32
g = torch.Generator().manual_seed(5)
secs = torch.empty(B).uniform_(14.4, 16.0, generator=g)          # one bucket: padding <= 10 %
lens = (secs * 16000).int()
lens[0] = 256000
L = int(lens.max())
wave = (0.1 * torch.randn(B, L, generator=g)).to(dev)
for i in range(B):
    wave[i, lens[i]:] = 0
S = int(2.8 * 16)
tg = torch.randint(0, V, (B, S), generator=g)
for s in range(1, S):
    same = tg[:, s] == tg[:, s - 1]; tg[same, s] = (tg[same, s] + 1) % V
tl = (2.8 * secs).int().clamp(max=S)
m = NativeModel("plain", V + 1, mask=True, act="relu", dtype=torch.bfloat16, device=dev); m.init_parameters(0)
ts = TrainStep(m, 1e-2, 1e-3)
args = (wave, tg.to(dev), tl.to(dev), lens.to(dev))
for _ in range(3):
    loss, nll, logp, am = ts.step(*args)
torch.cuda.synchronize()
assert torch.isfinite(loss).all() and torch.isfinite(m.grads).all(), "non-finite"
t0 = time.perf_counter(); n = 10
for _ in range(n):
    loss, *_ = ts.step(*args)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("cfg5-like: B=%d T_in=%d T'=%d C=%d  %.2f ms/step  %.0f audio-s/s (valid audio)  loss %.3f  ws %.2f GB" %
      (B, logp.shape[1] * 2 - 1, logp.shape[1], V + 1, dt * 1e3, float(secs.sum()) / dt, loss.item(), m._ws.numel() / 1e9))
