#!/usr/bin/env python3
"""Microbenchmark of the depthwise 3x3 launches at encoder sizes (B = 8, 256 x 416): forward (+ input norm, + output sums), data
gradient with / without the fused GroupNorm-backward reduce, 30 launches per graph replay.   python tools/prof_dwconv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import camradepth_amd.lib as L
from tests.util import to_stat, zsum
lb = L.load(); P = lambda t: t.data_ptr() if t is not None else None


def timed(fn, n=30):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 8
for (H, W, C, tag) in [(64, 104, 512, "stage 1"), (32, 52, 1024, "stage 2"), (16, 26, 1280, "stage 3"), (8, 13, 2048, "stage 4")]:
    x = torch.randn(B, H * W, C, device="cuda").to(torch.bfloat16); dy = torch.randn(B, H * W, C, device="cuda").to(torch.bfloat16)
    y = torch.zeros_like(x)
    w9 = torch.randn(9, C, device="cuda"); bias = torch.zeros(C, device="cuda")
    st_in = to_stat(torch.stack([torch.zeros(B, C // 16), torch.ones(B, C // 16) * H * W * 16], -1)).cuda()
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    ost = zsum(B, C // 16, 2)
    r = zsum(B * C * 2 + B * (C // 16) * 2)
    st = L.stream
    t_f = timed(lambda: lb.crd_dwconv3x3(P(x), B, H, W, C, P(w9), P(bias), 0, P(y), P(ost), P(st_in), 1, P(gam), P(bet), None, None, None, None, st()))
    t_f0 = timed(lambda: lb.crd_dwconv3x3(P(x), B, H, W, C, P(w9), P(bias), 0, P(y), None, None, 1, None, None, None, None, None, None, st()))
    t_b = timed(lambda: lb.crd_dwconv3x3(P(dy), B, H, W, C, P(w9), None, 1, P(y), None, None, 1, None, None, P(x), P(st_in), P(gam), P(r), st()))
    t_b0 = timed(lambda: lb.crd_dwconv3x3(P(dy), B, H, W, C, P(w9), None, 1, P(y), None, None, 1, None, None, None, None, None, None, st()))
    dw10 = zsum(16, 10, C)
    t_w = timed(lambda: lb.crd_dwconv3x3_wgrad(P(x), P(dy), B, H, W, C, P(dw10), 16, P(st_in), 1, P(gam), P(bet), st()))
    mb = B * H * W * C * 2 / 1e6
    print(f"{tag}: {B}x{H}x{W}x{C} ({mb:5.1f} MB): fwd(+norm,+sums) {t_f:6.2f}  fwd plain {t_f0:6.2f}  dgrad + fused reduce {t_b:6.2f}  dgrad plain {t_b0:6.2f}  wgrad(+norm) {t_w:6.2f} us")
