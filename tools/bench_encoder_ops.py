"""Developer microbench: the HBM-bound encoder kernels at stage-1 shapes (B=8, 64x104 tokens, C=64, hidden 512)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from camradepth_amd import lib
L = lib.load()
B, H, W, Cs, hid = 8, 64, 104, 64, 512
if len(sys.argv) > 1 and sys.argv[1] == "s2":
    H, W, Cs, hid = 32, 52, 128, 1024
N = H * W
bf = torch.bfloat16
def t(*shape, dtype=bf): return (torch.randn(*shape, device="cuda") * 0.5).to(dtype)
def timeit(name, fn, bytes_):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1000
    print(f"{name:34s} {us:8.1f} us   {bytes_ / us / 1e6:6.2f} TB/s (algorithmic {bytes_ / 1e6:.0f} MB)")
st = lib.stream
Hn, Hd = t(B, N, hid), t(B, N, hid)
Xn = t(B, N, Cs)
stats = torch.zeros(B, hid // 16, 2, dtype=torch.int64, device="cuda"); chan = torch.zeros(B, hid, 2, dtype=torch.int64, device="cuda")   # crd_sum_t
gam, bet = torch.ones(hid, device="cuda"), torch.zeros(hid, device="cuda")
out = torch.zeros(B, N, hid, dtype=bf, device="cuda")
L.crd_gn_stats(Hn.data_ptr(), 0, hid, 0, B, N, hid, stats.data_ptr(), None, st())
hb = B * N * hid * 2
timeit("gn_stats hidden", lambda: L.crd_gn_stats(Hn.data_ptr(), 0, hid, 0, B, N, hid, stats.data_ptr(), None, st()), hb)
timeit("gn_apply hidden (+gelu)", lambda: L.crd_gn_apply(Hn.data_ptr(), 0, hid, 0, B, N, hid, stats.data_ptr(), hid // Cs, gam.data_ptr(), bet.data_ptr(), 1, None, out.data_ptr(), 0, hid, 0, st()), 2 * hb)
r = torch.zeros(B * hid * 2 + B * (Cs // 16) * 2 * 8, dtype=torch.int64, device="cuda")
scr = torch.zeros(1024 * 2 * 1024, device="cuda")
timeit("gn_bwd_reduce hidden", lambda: L.crd_gn_bwd_reduce(Hn.data_ptr(), 0, hid, 0, Hd.data_ptr(), 0, hid, 0, B, N, hid, stats.data_ptr(), hid // Cs, gam.data_ptr(), bet.data_ptr(), 1, None, r.data_ptr(), scr.data_ptr(), scr.numel(), st()), 2 * hb)
dg, db = torch.zeros(hid, device="cuda"), torch.zeros(hid, device="cuda")
timeit("gn_bwd_apply hidden (in place)", lambda: L.crd_gn_bwd_apply(Hn.data_ptr(), 0, hid, 0, Hd.data_ptr(), 0, hid, 0, B, N, hid, stats.data_ptr(), hid // Cs, gam.data_ptr(), bet.data_ptr(), 1, None, r.data_ptr(), dg.data_ptr(), db.data_ptr(), Hd.data_ptr(), 0, hid, 0, 0, None, 0, None, st()), 3 * hb)
w9, b9 = torch.randn(9, hid, device="cuda"), torch.randn(hid, device="cuda")
timeit("dwconv fwd (+bias,+stats)", lambda: L.crd_dwconv3x3(Hn.data_ptr(), B, H, W, hid, w9.data_ptr(), b9.data_ptr(), 0, out.data_ptr(), stats.data_ptr(), st()), 2 * hb)
timeit("dwconv dgrad (flip)", lambda: L.crd_dwconv3x3(Hn.data_ptr(), B, H, W, hid, w9.data_ptr(), None, 1, out.data_ptr(), None, st()), 2 * hb)
dw9 = torch.zeros(16, 10, hid, dtype=torch.int64, device="cuda")
timeit("dwconv wgrad", lambda: L.crd_dwconv3x3_wgrad(Hn.data_ptr(), Hd.data_ptr(), B, H, W, hid, dw9.data_ptr(), db.data_ptr(), st()), 2 * hb)
# GEMMs
def conv(x, xC, w, cout, y, yC, k=1, s=1, OH=H, OW=W, bias=None, stats_=None, gather=0, partial=None, f32=0, res=None):
    d = lib.ConvDesc()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), xC, 0, B, H if gather == 0 else OH, W if gather == 0 else OW, xC
    d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), cout, k, k, s, 0, OH, OW
    d.gather_mode = gather
    d.y, d.y_ld, d.y_coff, d.y_f32 = y.data_ptr(), yC, 0, f32
    d.bias = bias.data_ptr() if bias is not None else None
    d.stats = stats_.data_ptr() if stats_ is not None else None
    if partial is not None: d.stats_partial, d.stats_partial_capacity = partial.data_ptr(), partial.numel()
    if res is not None: d.res, d.res_ld = res.data_ptr(), yC
    return d
w1, w2 = t(hid, Cs), t(Cs, hid)
part = torch.zeros(B * (N // 64 + 1) * (hid // 16) * 2, device="cuda")
d1 = conv(Xn, Cs, w1, hid, out, hid, bias=b9, stats_=stats, partial=part)
timeit("fc1 GEMM (+bias,+stats)", lambda: L.crd_conv_igemm(C.byref(d1), st()), hb + B * N * Cs * 2)
yo = torch.zeros(B, N, Cs, device="cuda"); res = torch.zeros(B, N, Cs, device="cuda")
d2 = conv(Hn, hid, w2, Cs, yo, Cs, bias=b9, f32=1, res=res)
timeit("fc2 GEMM (+bias,+residual fp32)", lambda: L.crd_conv_igemm(C.byref(d2), st()), hb + B * N * Cs * 8)
dxn = torch.zeros(B, N, Cs, dtype=bf, device="cuda")
d3 = conv(Hd, hid, t(Cs, hid), Cs, dxn, Cs, gather=1)
timeit("fc1 dgrad (hid->C)", lambda: L.crd_conv_igemm(C.byref(d3), st()), hb + B * N * Cs * 2)
d4 = conv(dxn, Cs, t(hid, Cs), hid, out, hid, gather=1)
timeit("fc2 dgrad (C->hid)", lambda: L.crd_conv_igemm(C.byref(d4), st()), hb + B * N * Cs * 2)
def wg(x, xC, dy, dyC):
    d = lib.WgradDesc()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), xC, 0, B, H, W, xC
    d.dy, d.dy_ld, d.dy_coff, d.OH, d.OW, d.Cout = dy.data_ptr(), dyC, 0, H, W, dyC
    d.KH, d.KW, d.stride, d.pad = 1, 1, 1, 0
    return d
dwa, dba = torch.zeros(hid, Cs, dtype=torch.int64, device="cuda"), torch.zeros(hid, dtype=torch.int64, device="cuda")
dA = wg(Xn, Cs, Hd, hid); dA.dw, dA.dbias = dwa.data_ptr(), dba.data_ptr()
timeit("fc1 wgrad (+dbias)", lambda: L.crd_conv_wgrad(C.byref(dA), st()), hb + B * N * Cs * 2)
dwb = torch.zeros(Cs, hid, dtype=torch.int64, device="cuda"); dbb = torch.zeros(Cs, dtype=torch.int64, device="cuda")
dB = wg(Hn, hid, dxn, Cs); dB.dw, dB.dbias = dwb.data_ptr(), dbb.data_ptr()
timeit("fc2 wgrad (+dbias)", lambda: L.crd_conv_wgrad(C.byref(dB), st()), hb + B * N * Cs * 2)
