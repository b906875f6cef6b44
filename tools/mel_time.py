"""Dev tool: time lasr_mel_fwd alone at the bench shape (B=32, 160 000 samples).  python tools/mel_time.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(3)
wave = (0.1 * torch.randn(32, 160000, generator=g)).to(dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(3):
    out = ops.mel(wave, None, None, None, True, torch.bfloat16, want_bft=False, want_btf=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = ops.mel(wave, None, None, None, True, torch.bfloat16, want_bft=False, want_btf=True)
e1.record(); torch.cuda.synchronize()
f32 = ops.mel(wave, None, None, None, True, torch.float32, want_bft=False, want_btf=True)[1]
print("%s: %.1f us per mel_fwd (db + normalise kernels); checksum %.6f abs-sum %.3f"
      % (os.environ.get("LASR_LIB_PATH", "default"), e0.elapsed_time(e1) / reps * 1e3, f32.double().sum().item(), f32.double().abs().sum().item()))
