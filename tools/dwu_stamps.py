"""Dev tool (debug build -DLASR_DW_STAMPS, LASR_LIB_PATH=<that build>): phase times of dwconv_bwd_uni_kernel (first tile of every workgroup)."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda")
lib.lasr_debug_set_dw_stamps.argtypes = [C.c_void_p]
B, T = 32, 501
stamps = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
assert lib.lasr_debug_set_dw_stamps(stamps.data_ptr()) == 0
g = torch.Generator().manual_seed(0)
for (Cc, k) in ((512, 63), (256, 33)):
    x = torch.randn(B, T, Cc, generator=g).bfloat16().to(dev); dy = torch.randn(B, T, Cc, generator=g).bfloat16().to(dev)
    w = (torch.randn(Cc, 1, k, generator=g) / math.sqrt(k)).to(dev); add = torch.randn(B, T, Cc, generator=g).bfloat16().to(dev)
    for _ in range(3):
        stamps.zero_(); torch.cuda.synchronize()
        ops.dwconv_bwd_fused(x, dy, w, add)
        torch.cuda.synchronize()
    t = stamps.view(2048, 8).cpu().double() * 0.01
    t = t[t[:, 6] > 0]
    ph = t[:, 1:7] - t[:, 0:6]
    names = ["stage+transpose(t0, incl. load latency)", "issue next loads", "wgrad MFMA", "dgrad MFMA", "out image + stores", "rest (tile 1..)"]
    print("C=%d k=%d: %d workgroups, span %.1f us (first start -> last end), per-WG total mean %.1f" % (Cc, k, t.shape[0], float(t[:, 6].max() - t[:, 0].min()), float((t[:, 6] - t[:, 0]).mean())))
    for i, n in enumerate(names):
        print("   %-42s mean %.2f  max %.2f us" % (n, float(ph[:, i].mean()), float(ph[:, i].max())))
    print("   start skew (max t0 - min t0): %.2f us" % float(t[:, 0].max() - t[:, 0].min()))
