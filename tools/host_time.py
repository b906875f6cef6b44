"""Dev tool: host-side enqueue time of one training step (no device sync inside the timed loop) against the device step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lightning_asr_amd.engine import NativeModel
from lightning_asr_amd.step import TrainStep
dev = torch.device("cuda")
m = NativeModel("plain", 28, mask=True, act="relu", dtype=torch.bfloat16, device=dev)
m.init_parameters(seed=0)
ts = TrainStep(m, 1e-2, 1e-3)
wave, tg, tl = bench.synth_batch(32, 160000, 100, 1234, dev)
for _ in range(10):
    ts.step(wave, tg, tl)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    ts.step(wave, tg, tl)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/step (loop returned after %.1f ms); device finished %.1f ms later; total %.3f ms/step"
      % ((t1 - t0) / n * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) / n * 1e3))
