import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import ops, _lib
dev = torch.device('cuda')
N, Co, Ci = 16032, 512, 512
x = torch.randn(N, Ci, device=dev).bfloat16(); w = (torch.randn(Co, Ci, device=dev) / 16).bfloat16()
lib = _lib.load()
y = torch.empty(N, Co, dtype=torch.bfloat16, device=dev)
ws = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
def run(n, wsp=None):
    for _ in range(n):
        lib.lasr_gemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), 1, 1, N, Co, Ci, 0, 0, None, None, None, 0, None, 1,
                      wsp.data_ptr() if wsp is not None else None, wsp.numel() * 8 if wsp is not None else 0, torch.cuda.current_stream().cuda_stream)
for dbg in (0, 4):
    os.environ["LASR_GEMM_DBG"] = str(dbg)
    run(5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(50); e1.record(); torch.cuda.synchronize()
    print("dbg=%d: %.1f us" % (dbg, e0.elapsed_time(e1) / 50 * 1e3))
os.environ["LASR_GEMM_DBG"] = "256"
run(3, ws); torch.cuda.synchronize()
ws.zero_(); torch.cuda.synchronize()
run(1, ws); torch.cuda.synchronize()
t = ws[:504 * 8].view(504, 8).cpu().double()
t0 = t[:, 0].min()
names = ["start", "loop_end", "ldswrite_end", "sync2", "stores_issued", "stores_done"]
print("per-block stamps (s_memtime ticks, 100 MHz?) relative to earliest start: median / max")
for i, nme in enumerate(names):
    d = t[:, i] - t0
    print("  %-14s median %8.0f  min %8.0f  max %8.0f" % (nme, d.median(), d.min(), d.max()))
d = t[:, 1:6] - t[:, 0:5]
for i in range(5):
    print("  phase %s->%s: median %8.0f max %8.0f" % (names[i], names[i + 1], d[:, i].median(), d[:, i].max()))
