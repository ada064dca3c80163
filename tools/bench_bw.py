import sys; sys.path.insert(0, ".")
import torch
x = torch.randn(8, 6656, 512, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
xf = torch.randn(8 * 6656 * 512 // 2, device="cuda")
def timeit(name, fn, nbytes):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1000
    print(f"{name:28s} {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s")
timeit("torch copy bf16 55MB", lambda: y.copy_(x), 2 * x.numel() * 2)
timeit("torch sum fp32 55MB", lambda: xf.sum(), xf.numel() * 4)
timeit("torch mul_ in place", lambda: x.mul_(1.0), 2 * x.numel() * 2)
big = torch.randn(8, 106496, 128, device="cuda").to(torch.bfloat16); big2 = torch.empty_like(big)
timeit("torch copy bf16 218MB", lambda: big2.copy_(big), 2 * big.numel() * 2)
