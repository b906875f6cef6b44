"""What hipBLASLt (through torch.matmul) needs for the cfg2 GEMM shapes: yardstick only, not used by the product."""
import torch
dev = torch.device('cuda')
N, Co, Ci = 16032, 512, 512
x = torch.randn(N, Ci, device=dev).bfloat16(); w = (torch.randn(Co, Ci, device=dev) / 16).bfloat16()
dy = torch.randn(N, Co, device=dev).bfloat16()
for _ in range(5):
    y = torch.matmul(x, w.t())          # fwd   NT
    dx = torch.matmul(dy, w)            # dgrad NN
    dw = torch.matmul(dy.t(), x)        # wgrad TN
torch.cuda.synchronize()
