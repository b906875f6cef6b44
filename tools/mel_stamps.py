"""Dev tool (debug build -DLASR_MEL_STAMPS, LASR_LIB_PATH=<that build>): phase times of the log-mel kernel (wave 0 of every workgroup)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda")
lib.lasr_debug_set_mel_stamps.argtypes = [C.c_void_p]
B, L = 32, 160000
nbx = (ops.mel_num_frames(L) + 15) // 16
stamps = torch.zeros(B * nbx * 8, dtype=torch.int64, device=dev)
assert lib.lasr_debug_set_mel_stamps(stamps.data_ptr()) == 0
g = torch.Generator().manual_seed(3)
wave = (0.1 * torch.randn(B, L, generator=g)).to(dev)
for _ in range(3):
    stamps.zero_(); torch.cuda.synchronize()
    ops.mel(wave, None, None, None, True, torch.bfloat16, want_bft=False, want_btf=True)
    torch.cuda.synchronize()
t = stamps.view(B * nbx, 8).cpu().double() * 0.01
ph = t[:, 1:8] - t[:, 0:7]
names = ["tables + filter weights + signal -> LDS", "it0: window + pass 0 + exchange", "it0: pass 1", "it0: pass 2", "it0: split spectra",
         "it0: mel + dB + store", "iteration 1 (whole)"]
print("%d workgroups, span %.1f us (first start -> last end), per-WG total mean %.1f us" % (t.shape[0], float(t[:, 7].max() - t[:, 0].min()),
      float((t[:, 7] - t[:, 0]).mean())))
for i, n in enumerate(names):
    print("   %-44s mean %.2f  max %.2f us" % (n, float(ph[:, i].mean()), float(ph[:, i].max())))
print("   start skew (max t0 - min t0): %.2f us" % float(t[:, 0].max() - t[:, 0].min()))
