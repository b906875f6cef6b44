"""Dev tool: time lasr_gemm_batch (two equal problems per launch) over K / N, kernel-only when run under
rocprofv3 (tools/gemm_sweep_report.py reads the trace).  Usage: python tools/gemm_sweep.py [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import _lib

class Prob(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
                ("bias", C.c_void_p), ("row_lens", C.c_void_p), ("rows_per_seq", C.c_int64), ("stats", C.c_void_p)]

lib = _lib.load()
lib.lasr_gemm_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 16032
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
CONFIGS = [(n, k, tb) for tb in (0, 1) for n in (256, 512, 1024) for k in (64, 128, 256, 512, 1024)]
if __name__ == "__main__":
    for (N, K, tB) in CONFIGS:
        ps = (Prob * 2)()
        keep = []
        for i in range(2):
            a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(K, N, device=dev) if tB else torch.randn(N, K, device=dev)).bfloat16()
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); s = torch.empty(2 * N, device=dev)
            keep += [a, b, c, s]
            ps[i].A, ps[i].B, ps[i].C, ps[i].M, ps[i].N, ps[i].K = a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K
            ps[i].bias = None; ps[i].row_lens = None; ps[i].rows_per_seq = 0; ps[i].stats = None if tB else s.data_ptr()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rc = lib.lasr_gemm_batch(ps, 2, 1, 1, 0, tB, 1, ws.data_ptr(), ws.numel(), st)
            _lib.check(rc, "lasr_gemm_batch")
        e1.record(); torch.cuda.synchronize()
        print("N=%4d K=%4d transB=%d  %7.1f us/launch (events, incl. host)" % (N, K, tB, e0.elapsed_time(e1) / reps * 1e3), flush=True)
