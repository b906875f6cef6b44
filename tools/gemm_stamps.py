"""Dev tool (debug library only): where a 256x256-tile GEMM workgroup spends its time.
Build: hipcc ... -DLASR_GEMM_STAMPS gemm_bf16.hip -> build/liblasr_stamps.so; run with LASR_LIB_PATH set to it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib
from gemm_sweep import Prob

lib = _lib.load()
assert hasattr(lib, "lasr_debug_set_gemm_stamps"), "needs the -DLASR_GEMM_STAMPS debug build"
lib.lasr_gemm_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
lib.lasr_debug_set_gemm_stamps.argtypes = [C.c_void_p]
dev = torch.device("cuda")
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
assert lib.lasr_debug_set_gemm_stamps(stamps.data_ptr()) == 0
M = 16032
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for (N, K, tB) in [(512, 512, 0), (512, 64, 0), (256, 256, 0), (512, 512, 1), (1024, 512, 0)]:
    ps = (Prob * 2)(); keep = []
    for i in range(2):
        a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(K, N, device=dev) if tB else torch.randn(N, K, device=dev)).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); s = torch.empty(2 * N, device=dev)
        keep += [a, b, c, s]
        ps[i].A, ps[i].B, ps[i].C, ps[i].M, ps[i].N, ps[i].K = a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K
        ps[i].bias = None; ps[i].row_lens = None; ps[i].rows_per_seq = 0; ps[i].stats = None if tB else s.data_ptr()
    for _ in range(3):
        stamps.zero_(); torch.cuda.synchronize()
        _lib.check(lib.lasr_gemm_batch(ps, 2, 1, 1, 0, tB, 1, ws.data_ptr(), ws.numel(), st), "gemm")
        torch.cuda.synchronize()
    t = stamps.view(4096, 8).cpu()
    cyc, bar = t[:, 6].double(), t[:, 7].double()   # shader cycles (s_memtime) across the K loop; of them, wave 0 at the loop's barriers
    e = t[:, [2, 5, 3]].double() * 0.01
    nb = int((t[:, 0] != 0).sum())
    t = t[:nb, :5].double() * 0.01          # 100 MHz ticks -> us
    t0 = t[:, 0].min()
    ph = (t[:, 1:] - t[:, :-1])
    print("N=%d K=%d tB=%d: %d workgroups; launch span %.1f us; start skew mean %.1f max %.1f us" % (
        N, K, tB, nb, float(t[:, 4].max() - t0), float((t[:, 0] - t0).mean()), float((t[:, 0] - t0).max())))
    kl = ph[:, 1]
    print("   K loop: %.0f shader cycles mean (%.0f per 64-deep K step) in %.2f us -> shader clock %.0f MHz while it runs; wave 0 at the steady-state barriers: %.0f cycles per step"
          % (float(cyc[:nb].mean()), float(cyc[:nb].mean()) / max(K // 64, 1), float(kl.mean()), float((cyc[:nb] / kl).mean()),
             float(bar[:nb].mean()) / max(K // 64 - 2, 1)))
    ee = (e[:nb, 1:] - e[:nb, :-1])
    print("   epilogue of wave 0: convert + LDS image + barrier %.2f | 16 row-pair stores %.2f us (means); max %s" % (
        *[float(ee[:, i].mean()) for i in range(2)], [round(float(ee[:, i].max()), 1) for i in range(2)]))
    for i, nm in enumerate(["prologue (1st loads + LDS store)", "K loop", "epilogue (convert, LDS transpose, stores issued)", "stats + store drain"]):
        print("   %-52s mean %6.2f  min %6.2f  max %6.2f us" % (nm, float(ph[:, i].mean()), float(ph[:, i].min()), float(ph[:, i].max())))
