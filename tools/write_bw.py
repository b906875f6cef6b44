"""Dev tool: what the memory system takes for pure writes / reads / copies of GEMM-output-sized buffers."""
import torch, sys
dev = torch.device("cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (16, 33, 66, 132, 512):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev); y = torch.empty_like(x)
    t_fill = timeit(lambda: x.zero_())
    t_copy = timeit(lambda: y.copy_(x))
    t_sum = timeit(lambda: x.view(torch.int16).sum())
    print("%4d MB: fill %6.1f us (%.2f TB/s)  copy %6.1f us (%.2f TB/s r+w)  read(sum) %6.1f us (%.2f TB/s)" % (
        mb, t_fill, mb * 1.048576 / t_fill, t_copy, 2 * mb * 1.048576 / t_copy, t_sum, mb * 1.048576 / t_sum))
