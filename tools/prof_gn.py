#!/usr/bin/env python3
"""Microbenchmark of the GroupNorm kernels at encoder sizes (B = 8): crd_gn_apply, crd_gn_bwd_reduce, crd_gn_bwd_apply with / without
the parameter gradients, 50 launches per graph replay.   python tools/prof_gn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import camradepth_amd.lib as L
from tests.util import to_stat, zsum
lb = L.load(); P = lambda t: t.data_ptr() if t is not None else None


def timed(fn, n=50):
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
for (Pn, C, gmul, act, xf32, tag) in [(104, 512, 1, 0, 1, "Block.norm st4 (fp32 x)"), (104, 2048, 4, 1, 0, "Mlp.norm2+GELU st4"), (416, 320, 1, 0, 1, "Block.norm st3"),
                                      (416, 1280, 4, 1, 0, "Mlp.norm2+GELU st3"), (416, 1280, 1, 0, 0, "Mlp.norm1 st3"), (1664, 1024, 8, 1, 0, "Mlp.norm2+GELU st2"),
                                      (6656, 512, 8, 1, 0, "Mlp.norm2+GELU st1"), (6656, 64, 1, 0, 1, "Block.norm st1")]:
    x = (torch.randn(B, Pn, C, device="cuda") if xf32 else torch.randn(B, Pn, C, device="cuda").to(torch.bfloat16))
    dy = torch.randn(B, Pn, C, device="cuda").to(torch.bfloat16)
    G = C // (16 * gmul)
    stats = to_stat(torch.stack([torch.zeros(B, C // 16), torch.ones(B, C // 16) * Pn * 16], -1)).cuda()
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    r = zsum(B * C * 2 + B * G * 2)
    y = torch.zeros(B, Pn, C, dtype=torch.bfloat16, device="cuda")
    dx = torch.zeros(B, Pn, C, dtype=torch.bfloat16, device="cuda")
    dgam, dbet = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    st = L.stream
    t_ap = timed(lambda: lb.crd_gn_apply(P(x), xf32, C, 0, B, Pn, C, P(stats), gmul, P(gam), P(bet), act, None, P(y), 0, C, 0, st()))
    t_red = timed(lambda: lb.crd_gn_bwd_reduce(P(x), xf32, C, 0, P(dy), 0, C, 0, B, Pn, C, P(stats), gmul, P(gam), P(bet), act, None, P(r), None, 0, st()))
    t_ba = timed(lambda: lb.crd_gn_bwd_apply(P(x), xf32, C, 0, P(dy), 0, C, 0, B, Pn, C, P(stats), gmul, P(gam), P(bet), act, None, P(r), P(dgam), P(dbet),
                                             P(dx), 0, C, 0, 0, None, 0, None, st()))
    t_bn = timed(lambda: lb.crd_gn_bwd_apply(P(x), xf32, C, 0, P(dy), 0, C, 0, B, Pn, C, P(stats), gmul, P(gam), P(bet), act, None, P(r), None, None,
                                             P(dx), 0, C, 0, 0, None, 0, None, st()))
    mb = B * Pn * C * 2 / 1e6
    print(f"{tag:26s} {B}x{Pn}x{C} ({mb:6.1f} MB bf16): apply {t_ap:6.2f}  bwd_reduce {t_red:6.2f}  bwd_apply {t_ba:6.2f}  bwd_apply w/o dgamma,dbeta {t_bn:6.2f} us")
