#!/usr/bin/env python3
"""Microbenchmark of the attention forward launches at encoder shapes: crd_attn_scores alone, crd_attn_xbar_proj alone and the
fused crd_attn_fwd, 50 launches per graph replay.   python tools/prof_attn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import camradepth_amd.lib as L
from tests.util import to_stat

lb = L.load()
P = lambda t: t.data_ptr()


def timed(fn, n=50):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
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


for (B, N, M, heads, d, tag) in [(1, 20800, 325, 1, 64, "416x800 stage 1"), (1, 5200, 325, 2, 64, "416x800 stage 2"), (1, 1300, 325, 5, 64, "416x800 stage 3"),
                                 (1, 325, 325, 8, 64, "416x800 stage 4"), (8, 6656, 104, 1, 64, "256x416 B=8 stage 1"), (8, 416, 104, 5, 64, "256x416 B=8 stage 3"),
                                 (8, 104, 104, 8, 64, "256x416 B=8 stage 4")]:
    C = heads * d
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, N, C, generator=g).to(torch.bfloat16).cuda()
    k = torch.randn(B, M, C, generator=g).to(torch.bfloat16).cuda()
    S = torch.zeros(B, N, device="cuda"); idx = torch.zeros(B, N, heads, dtype=torch.int16, device="cuda")
    chan = to_stat(torch.randn(B, C, 2, generator=g)).cuda()
    st = to_stat(torch.stack([torch.randn(B, C // 16, generator=g), 20.0 + torch.rand(B, C // 16, generator=g)], -1) * N).cuda()
    gam, bet = torch.ones(C).cuda(), torch.zeros(C).cuda()
    wf = (0.2 * torch.randn(C, C, generator=g)).to(torch.bfloat16).cuda()
    xb, u = torch.zeros(B, C, dtype=torch.bfloat16, device="cuda"), torch.zeros(B, C, device="cuda")
    scale = d ** -0.5
    t_s = timed(lambda: lb.crd_attn_scores(P(q), P(k), B, N, M, heads, d, scale, P(S), P(idx), L.stream()))
    t_v = timed(lambda: lb.crd_attn_xbar_proj(P(chan), P(st), P(gam), P(bet), P(wf), B, N, C, P(xb), P(u), L.stream()))
    t_f = timed(lambda: lb.crd_attn_fwd(P(q), P(k), B, N, M, heads, d, scale, P(S), P(idx), P(chan), P(st), P(gam), P(bet), P(wf), P(xb), P(u), L.stream()))
    print(f"{tag:24s} B={B} N={N} M={M} heads={heads}: scores {t_s:6.2f} us   xbar+proj {t_v:6.2f} us   fused {t_f:6.2f} us")
