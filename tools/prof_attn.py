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

# ---- backward pieces at the benchmark size (B = 8, 256 x 416)
from tests.util import zsum
print("backward (B = 8, 256 x 416): crd_attn_out_bwd, crd_attn_scores_bwd (with partial copies), crd_sum_partials_bf16")
for (N, M, heads, d, tag) in [(6656, 104, 1, 64, "stage 1"), (1664, 104, 2, 64, "stage 2"), (416, 104, 5, 64, "stage 3"), (104, 104, 8, 64, "stage 4")]:
    B, C = 8, heads * d
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, N, C, generator=g).to(torch.bfloat16).cuda(); k = torch.randn(B, M, C, generator=g).to(torch.bfloat16).cuda()
    idx = torch.randint(0, M, (B, N, heads), generator=g).to(torch.int16).cuda()
    dS = torch.randn(B, N, generator=g).cuda(); S = torch.randn(B, N, generator=g).cuda()
    dx1 = torch.randn(B, N, C, generator=g).cuda(); u = torch.randn(B, C, generator=g).cuda()
    t, dbp = zsum(B, C), zsum(B, C)
    dq = torch.zeros(B, N, C, dtype=torch.bfloat16, device="cuda"); dk = zsum(B, M, C)
    Pn = lb.crd_attn_scores_bwd_partials(B, N, M, heads, d)
    parts = torch.zeros(max(Pn, 1), B, M, C, device="cuda")
    dkb = torch.zeros(B, M, C, dtype=torch.bfloat16, device="cuda")
    scale = d ** -0.5
    t_o = timed(lambda: lb.crd_attn_out_bwd(P(dx1), P(u), P(S), None, B, N, C, P(t), P(dbp), P(dS), L.stream()))
    t_s = timed(lambda: lb.crd_attn_scores_bwd(P(q), P(k), P(dS), P(idx), B, N, M, heads, d, scale, P(dq), P(dk), P(parts) if Pn > 0 else None, L.stream()))
    t_p = timed(lambda: lb.crd_sum_partials_bf16(P(parts), max(Pn, 1), B * M * C, P(dkb), B * M * C, L.stream()))
    print(f"{tag}: N={N} heads={heads} partial copies {Pn}: out_bwd {t_o:6.2f} us   scores_bwd {t_s:6.2f} us   sum_partials {t_p:6.2f} us")
