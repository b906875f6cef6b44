"""Dev tool (GPU box): time lasr_dwconv_bwd_fused at the cfg2 layer shapes.  LASR_DW_UNI=0|32|64 selects the kernel (read once per process).
usage: python tools/dw_bwd_time.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_asr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (C, k, T, B) in [(512, 51, 501, 32), (512, 63, 501, 32), (512, 75, 501, 32), (256, 33, 501, 32), (256, 39, 501, 32), (512, 63, 801, 32)]:
    x = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
    dy = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
    w = (torch.randn(C, 1, k, generator=g) / math.sqrt(k)).to(dev)
    add = torch.randn(B, T, C, generator=g).bfloat16().to(dev)
    junk = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    for _ in range(3):
        ops.dwconv_bwd_fused(x, dy, w, add)
    ts = []
    for _ in range(20):
        junk.zero_()                      # evict: the step never finds these tensors in the memory-side cache
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.dwconv_bwd_fused(x, dy, w, add)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("LASR_DW_UNI=%s C=%d k=%d T=%d B=%d: median %.1f us, min %.1f us (events, incl. ~5 us bracket)" % (os.environ.get("LASR_DW_UNI", "default"), C, k, T, B, ts[len(ts) // 2], ts[0]))
