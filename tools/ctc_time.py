"""Dev tool: time lasr_ctc_loss alone at the bench shape (B=32, T'=501, C=28, S=100).  python tools/ctc_time.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import ops
import bench
dev = torch.device("cuda")
B, T, C, S = 32, int(os.environ.get("CTC_T", "501")), 28, int(os.environ.get("CTC_S", "100"))
_, tg, tl = bench.synth_batch(B, 16, S, 1, dev)
logp = torch.randn(B, T, C, device=dev).log_softmax(-1)
il = torch.full((B,), T, dtype=torch.int32, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(3):
    nll, grad = ops.ctc_loss(logp, tg, il, tl, blank=C - 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    nll, grad = ops.ctc_loss(logp, tg, il, tl, blank=C - 1)
e1.record(); torch.cuda.synchronize()
ref = torch.nn.functional.ctc_loss(logp.cpu().transpose(0, 1), tg.cpu(), il.cpu().long(), tl.cpu().long(), blank=C - 1, reduction="none")
print("T=%d S=%d " % (T, S) + "%s: %.1f us per ctc_loss (alpha/beta + grad); nll max rel err vs torch CPU %.2e"
      % (os.environ.get("LASR_LIB_PATH", "default"), e0.elapsed_time(e1) / reps * 1e3, ((nll.cpu() - ref).abs() / ref.abs()).max().item()))
