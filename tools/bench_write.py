"""Developer microbench: what a pure streaming WRITE reaches on this chip (torch fill_ over rotating buffers, graph-replayed) -- the
yardstick for the wide-output GEMM epilogues (fc1 forward / fc2 data gradient of encoder stages 1-2: 54.5 / 27 MB of bf16)."""
import torch
for mb in (6.8, 27.3, 54.5, 218.0):
    n = int(mb * 1e6 / 2)
    bufs = [torch.empty(n, dtype=torch.bfloat16, device="cuda") for _ in range(12)]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for b in bufs:
            b.fill_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for r in range(48):
                bufs[r % 12].fill_(1.0)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 48
    print(f"fill {mb:6.1f} MB: {us:7.2f} us  {mb / us:5.2f} TB/s")
