"""Dev tool: device time of an eval-mode forward (B=32, 10 s clips), bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lightning_asr_amd import ops
from lightning_asr_amd.engine import NativeModel
dev = torch.device("cuda")
m = NativeModel("plain", 28, mask=True, act="relu", dtype=torch.bfloat16, device=dev)
m.init_parameters(seed=0)
wave, tg, tl = bench.synth_batch(32, 160000, 100, 1234, dev)
_, feats, _, pct = ops.mel(wave, None, None, None, True, torch.bfloat16, want_bft=False, want_btf=True)
for _ in range(5):
    m.forward(feats, pct, training=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    out = m.forward(feats, pct, training=False)
e1.record(); torch.cuda.synchronize()
logp = out[0] if isinstance(out, (tuple, list)) else out
print("%s: eval forward %.1f us (B=32 x 10 s: %.0f audio-s/s); logp checksum %.6f"
      % (os.environ.get("LASR_LIB_PATH", "default"), e0.elapsed_time(e1) / 30 * 1e3, 320.0 / (e0.elapsed_time(e1) / 30 * 1e-3), logp.double().sum().item()))
