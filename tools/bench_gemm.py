"""Micro-benchmark of lasr_gemm on the cfg2 shapes (dev tool).  torch.matmul (hipBLASLt) is timed
beside it only as a yardstick for what the machine can do on the same shape."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import ops

dev = torch.device('cuda')
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

N = 16032
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
for (Co, Ci) in [(512, 512), (256, 256), (1024, 512)]:
    x = torch.randn(N, Ci, device=dev).to(dt); w = (torch.randn(Co, Ci, device=dev) / 16).to(dt)
    dy = torch.randn(N, Co, device=dev).to(dt)
    lens = torch.full((32,), 501, dtype=torch.int32, device=dev)
    t_fwd = timeit(lambda: ops.gemm(x, w, N, Co, Ci, want_stats=True, row_lens=lens, rows_per_seq=501))
    t_fwd_plain = timeit(lambda: ops.gemm(x, w, N, Co, Ci))
    t_dg = timeit(lambda: ops.gemm(dy, w, N, Ci, Co, transB=True))
    t_wg = timeit(lambda: ops.gemm(dy, x, Co, Ci, N, transA=True, transB=True, split_k=16, out_dtype=torch.float32))
    t_ref = timeit(lambda: torch.matmul(x, w.t()))
    t_ref_wg = timeit(lambda: torch.matmul(dy.t(), x))
    fl = 2.0 * N * Co * Ci
    by = (N * Ci + Co * Ci + N * Co) * x.element_size()
    print("Co=%4d Ci=%4d  fwd+stats %6.1f us (plain %6.1f; %5.0f TF, %5.0f GB/s) | dgrad %6.1f | wgrad %6.1f | torch.matmul fwd %6.1f wgrad %6.1f"
          % (Co, Ci, t_fwd, t_fwd_plain, fl / t_fwd_plain / 1e6, by / t_fwd_plain / 1e3, t_dg, t_wg, t_ref, t_ref_wg))
