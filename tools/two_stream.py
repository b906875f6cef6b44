"""Does running independent GEMMs on two HIP streams overlap one kernel's store tail with the next one's
loads?  (dev experiment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib
dev = torch.device('cuda')
lib = _lib.load()
N, Co, Ci = 16032, 512, 512
x = torch.randn(N, Ci, device=dev).bfloat16(); u = torch.randn(N, Ci, device=dev).bfloat16()
w1 = (torch.randn(Co, Ci, device=dev) / 16).bfloat16(); w2 = (torch.randn(Co, Ci, device=dev) / 16).bfloat16()
y1 = torch.empty(N, Co, dtype=torch.bfloat16, device=dev); y2 = torch.empty_like(y1)
def gemm(a, w, y, st):
    lib.lasr_gemm(a.data_ptr(), w.data_ptr(), y.data_ptr(), 1, 1, N, Co, Ci, 0, 0, None, None, None, 0, None, 1, None, 0, st)
s0 = torch.cuda.current_stream(); s1 = torch.cuda.Stream()
def seq(n):
    for _ in range(n):
        gemm(u, w1, y1, s0.cuda_stream); gemm(x, w2, y2, s0.cuda_stream)
def par(n):
    for _ in range(n):
        e = torch.cuda.Event(); e.record(s0); s1.wait_event(e)
        gemm(u, w1, y1, s0.cuda_stream); gemm(x, w2, y2, s1.cuda_stream)
        e2 = torch.cuda.Event(); e2.record(s1); s0.wait_event(e2)
for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    fn(5); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(40); b.record(); torch.cuda.synchronize()
    print("%-12s %.1f us per pair" % (name, a.elapsed_time(b) / 40 * 1e3))
