#!/usr/bin/env python3
"""Microbenchmark of the encoder's write-heavy GEMMs at the benchmark size (B = 8, 256 x 416): fc1 behind Block.norm2 through
crd_gn_conv (stage 1: 64 -> 512 on 6656 pixels, stage 2: 128 -> 1024 on 1664), with / without the output GroupNorm sums and the
stored normalised operand, and under the kernel's ablation bits (CRD_DBG: 1 = no (scale, shift) table, 4 = no epilogue).
    python tools/prof_fc1.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import camradepth_amd.lib as lib
from tests.util import to_stat, zsum

L = lib.load()


def timed(fn, n=20):
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


for (B, Cin, H, W, Cout, tag) in [(8, 64, 64, 104, 512, "fc1 stage 1"), (8, 128, 32, 52, 1024, "fc1 stage 2"), (8, 320, 16, 26, 1280, "fc1 stage 3")]:
    x = torch.randn(B, H * W, Cin, device="cuda")
    stats = to_stat(torch.stack([torch.zeros(B, Cin // 16), torch.ones(B, Cin // 16) * H * W * 16], -1)).cuda()
    w = (torch.randn(Cout, Cin) / Cin ** 0.5).to(torch.bfloat16).cuda()
    bias = torch.zeros(Cout, device="cuda"); gam = torch.ones(Cin, device="cuda"); bet = torch.zeros(Cin, device="cuda")
    y = torch.zeros(B, H * W, Cout, dtype=torch.bfloat16, device="cuda")
    xn = torch.zeros(B, H * W, Cin, dtype=torch.bfloat16, device="cuda")
    ost = zsum(B, Cout // 16, 2)
    res = {}
    for with_stats in (1, 0):
        for with_xn in (1, 0):
            d, n = lib.ConvDesc(), lib.GnInput()
            d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), Cin, 0, B, H, W, Cin
            d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), Cout, 1, 1, 1, 0, H, W
            d.bias = bias.data_ptr()
            d.y, d.y_ld, d.y_f32 = y.data_ptr(), Cout, 0
            if with_stats:
                d.stats = ost.data_ptr()
            n.x_f32, n.gmul, n.act = 1, 1, 0
            n.stats, n.gamma, n.beta = stats.data_ptr(), gam.data_ptr(), bet.data_ptr()
            if with_xn:
                n.xn, n.xn_ld = xn.data_ptr(), Cin
            res[(with_stats, with_xn)] = timed(lambda: lib.check(L.crd_gn_conv(C.byref(d), C.byref(n), lib.stream()), "gn_conv"))
    mb = (x.numel() * 4 + y.numel() * 2) / 1e6
    print(f"{tag}: {Cin}->{Cout} on {B}x{H * W} px, {mb:.0f} MB in+out, CRD_DBG={os.environ.get('CRD_DBG', '0')}: "
          + "  ".join(f"stats={k[0]} xn={k[1]}: {v:6.1f} us ({mb / v / 1e3 * 1e3:.2f} TB/s)" for k, v in res.items()))
