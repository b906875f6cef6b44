"""Dev tool: is the stage's weight-gradient launch (lasr_gemm_multi_split_partials, 256 x 256 tiles, one round) bound by its operand
traffic or by the MFMA pipe?  The same 16 problems (cfg2's stage: K = 16 032 rows, 512 x 512) on 16 distinct operand pairs (~525 MB
to fetch) and on ONE pair 16 times over (33 MB, cache-resident after the first tiles): same flops, same tiles, same slabs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import ops
dev = torch.device("cuda")
rows, C, n = 16032, 512, 16
g = torch.Generator().manual_seed(1)
dys = [torch.randn(rows, C, generator=g).bfloat16().to(dev) for _ in range(n)]
xs = [torch.randn(rows, C, generator=g).bfloat16().to(dev) for _ in range(n)]
cold = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev)   # sweeps the caches between timed launches


def time_it(a, b, flush):
    ts = []
    for _ in range(12):
        if flush:
            cold.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.wgrad_multi(a, b, split_k=4); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for flush in (True, False):
    d = time_it(dys, xs, flush)
    s = time_it([dys[0]] * n, [xs[0]] * n, flush)
    print("caches %s: 16 distinct operand pairs %.1f us, one pair 16 times %.1f us (launch + reduction; median of 12)"
          % ("swept before each launch" if flush else "warm", d, s))
