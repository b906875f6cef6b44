"""Per-kernel micro-benchmark on cfg2 shapes through the raw C ABI (dev tool).  Usage: bench_ops.py [dw|bn|all]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib
dev = torch.device('cuda'); lib = _lib.load()
B, T, C, k = 32, 501, 512, int(os.environ.get("K", "63"))
N = B * T
bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
which = sys.argv[1] if len(sys.argv) > 1 else "all"
x = torch.randn(B, T, C, device=dev).to(bf); y = torch.empty_like(x); w = torch.randn(C, k, device=dev) / 8
dy = torch.randn(B, T, C, device=dev).to(bf); add = torch.randn(B, T, C, device=dev).to(bf)
mb = lambda nbytes, us: nbytes / us / 1e6
if which in ("dw", "all"):
    t = timeit(lambda: lib.lasr_dwconv_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 1, B, T, C, k, 1, 0, st()))
    print("dwconv fwd      k=%d: %6.1f us  (%.2f TB/s algorithmic)" % (k, t, mb(2 * N * C * 2, t)))
    t = timeit(lambda: lib.lasr_dwconv_fwd(dy.data_ptr(), w.data_ptr(), add.data_ptr(), y.data_ptr(), 1, B, T, C, k, 1, 1, st()))
    print("dwconv flip+add k=%d: %6.1f us  (%.2f TB/s)" % (k, t, mb(3 * N * C * 2, t)))
    nb = lib.lasr_dwconv_wgrad_workspace_bytes(B, T, C, k); ws = torch.empty(nb, dtype=torch.uint8, device=dev); dw = torch.empty(C, k, device=dev)
    t = timeit(lambda: lib.lasr_dwconv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, B, T, C, k, 1, ws.data_ptr(), nb, st()))
    print("dwconv wgrad    k=%d: %6.1f us  (%.2f TB/s)" % (k, t, mb(2 * N * C * 2, t)))
if which in ("bn", "all"):
    y2 = torch.randn(B, T, C, device=dev).to(bf); out = torch.empty_like(x)
    coef = torch.randn(2 * C, device=dev); saved = torch.rand(2 * C, device=dev) + 0.5; gam = torch.ones(C, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    t = timeit(lambda: lib.lasr_bn_act_fwd(x.data_ptr(), coef.data_ptr(), y2.data_ptr(), coef.data_ptr(), None, out.data_ptr(), 1, B, T, C, 1, st()))
    print("bn_act_fwd          : %6.1f us  (%.2f TB/s)" % (t, mb(3 * N * C * 2, t)))
    sums = torch.zeros(2 * C, device=dev); sums2 = torch.zeros(2 * C, device=dev)
    nb = lib.lasr_bn_bwd_workspace_bytes(B, T, C); ws2 = torch.empty(nb, dtype=torch.uint8, device=dev)
    t = timeit(lambda: lib.lasr_bn_act_bwd_stats(dy.data_ptr(), x.data_ptr(), coef.data_ptr(), saved.data_ptr(), y2.data_ptr(), coef.data_ptr(), saved.data_ptr(),
                                                 None, None, sums.data_ptr(), sums2.data_ptr(), 1, B, T, C, 1, ws2.data_ptr(), nb, st()))
    print("bn_bwd_stats(+reduce): %6.1f us  (%.2f TB/s)" % (t, mb(3 * N * C * 2, t)))
    d1 = torch.empty_like(x); d2 = torch.empty_like(x); dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    t = timeit(lambda: lib.lasr_bn_act_bwd_apply(dy.data_ptr(), x.data_ptr(), coef.data_ptr(), saved.data_ptr(), gam.data_ptr(), y2.data_ptr(), coef.data_ptr(),
                                                 saved.data_ptr(), gam.data_ptr(), None, None, sums.data_ptr(), sums2.data_ptr(), lens.data_ptr(), d1.data_ptr(),
                                                 d2.data_ptr(), dg.data_ptr(), db.data_ptr(), dg.data_ptr(), db.data_ptr(), 1, B, T, C, 1, ws2.data_ptr(), nb, st()))
    print("bn_bwd_apply        : %6.1f us  (%.2f TB/s)" % (t, mb(5 * N * C * 2, t)))
    z = torch.empty_like(x)
    t = timeit(lambda: z.copy_(x))
    print("torch copy (yardstick): %6.1f us  (%.2f TB/s)" % (t, mb(2 * N * C * 2, t)))
