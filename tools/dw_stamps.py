"""Dev tool (debug build -DLASR_DW_STAMPS): phase times of the MFMA depthwise conv."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib
lib = _lib.load(); dev = torch.device("cuda")
lib.lasr_debug_set_dw_stamps.argtypes = [C.c_void_p]
B, T, Cc = 32, 501, 512
st = lambda: torch.cuda.current_stream().cuda_stream
stamps = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
assert lib.lasr_debug_set_dw_stamps(stamps.data_ptr()) == 0
for k in (33, 63, 75):
    x = torch.randn(B, T, Cc, device=dev).bfloat16(); y = torch.empty_like(x); w = torch.randn(Cc, k, device=dev) / 8
    for _ in range(3):
        stamps.zero_(); torch.cuda.synchronize()
        _lib.check(lib.lasr_dwconv_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 1, B, T, Cc, k, 1, 0, st()), "dw")
        torch.cuda.synchronize()
    t = stamps.view(256, 8)[:, :4].cpu().double() * 0.01
    ph = t[:, 1:] - t[:, :-1]
    print("k=%d: span %.1f us | stage+transpose %.2f | tables+MFMA %.2f | stores %.2f us (means; max %s)" % (
        k, float(t[:, 3].max() - t[:, 0].min()), *[float(ph[:, i].mean()) for i in range(3)], [round(float(ph[:, i].max()), 1) for i in range(3)]))
