"""Dev tool: does a split-K weight-gradient GEMM launch amortise its fixed costs with longer K slices?
Two 512x512 problems per launch, split 16: K = 16032 (today's per-unit launch) against K = 4 x 16032 (what a
slice would see if four units' gradients were batched into one launch with split 4)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib
from gemm_sweep import Prob
lib = _lib.load()
lib.lasr_gemm_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
def run(K, split, reps=20):
    ps = (Prob * 2)(); keep = []
    for i in range(2):
        a = torch.randn(K, 512, device=dev).bfloat16(); b = torch.randn(K, 512, device=dev).bfloat16(); c = torch.empty(512, 512, device=dev)
        keep += [a, b, c]
        ps[i].A, ps[i].B, ps[i].C, ps[i].M, ps[i].N, ps[i].K = a.data_ptr(), b.data_ptr(), c.data_ptr(), 512, 512, K
        ps[i].bias = None; ps[i].row_lens = None; ps[i].rows_per_seq = 0; ps[i].stats = None
    for _ in range(3): _lib.check(lib.lasr_gemm_batch(ps, 2, 1, 0, 1, 1, split, ws.data_ptr(), ws.numel(), st), "g")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): _lib.check(lib.lasr_gemm_batch(ps, 2, 1, 0, 1, 1, split, ws.data_ptr(), ws.numel(), st), "g")
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for K, split in [(16032, 16), (16032 * 2, 16), (16032 * 4, 16), (16032 * 4, 32), (16032 * 8, 32)]:
    t = run(K, split)
    print("K=%6d split=%2d: %7.1f us per launch  = %6.1f us per 16032 rows of K  (%.0f TFLOP/s)" % (K, split, t, t * 16032 / K, 2 * 2.0 * 512 * 512 * K / t / 1e6))
