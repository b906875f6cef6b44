"""Dev tool (debug build -DLASR_DW_STAMPS): phase times of the MFMA depthwise weight gradient (first u tile)."""
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
    x = torch.randn(B, T, Cc, device=dev).bfloat16(); dy = torch.randn(B, T, Cc, device=dev).bfloat16(); dw = torch.empty(Cc, k, device=dev)
    nb = lib.lasr_dwconv_wgrad_workspace_bytes(B, T, Cc, k); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    for _ in range(3):
        stamps.zero_(); torch.cuda.synchronize()
        _lib.check(lib.lasr_dwconv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, B, T, Cc, k, 1, ws.data_ptr(), nb, st()), "dwg")
        torch.cuda.synchronize()
    t = stamps.view(256, 8)[:, :4].cpu().double() * 0.01
    print("k=%d: span %.1f us | tile 0: stage+transpose %.2f | issue of tile 1's loads %.2f | then (MFMA tile 0 + stage/MFMA tile 1 + stores) %.2f us" % (
        k, float(t[:, 3].max() - t[:, 0].min()), float((t[:, 1] - t[:, 0]).mean()), float((t[:, 2] - t[:, 1]).mean()), float((t[:, 3] - t[:, 2]).mean())))
