import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightning_asr_amd import ops
dev = torch.device('cuda')
N, Co, Ci = 16032, 512, 512
x = torch.randn(N, Ci, device=dev).bfloat16(); w = (torch.randn(Co, Ci, device=dev) / 16).bfloat16()
dy = torch.randn(N, Co, device=dev).bfloat16()
for _ in range(3):
    ops.gemm(x, w, N, Co, Ci)                                             # fwd NT
    ops.gemm(dy, w, N, Ci, Co, transB=True)                               # dgrad
    ops.gemm(dy, x, Co, Ci, N, transA=True, transB=True, split_k=16, out_dtype=torch.float32)   # wgrad
torch.cuda.synchronize()
