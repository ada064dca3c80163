"""Developer microbench: crd_gn_conv against the pair of launches it replaces (crd_gn_apply + crd_conv_igemm) on the
encoder's shapes, replayed from a HIP graph over rotating operand sets (operands from HBM / MALL as in the real step)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from camradepth_amd import lib
L = lib.load()
B = 8
# Cin, Cout, H, W, gmul, act, x_f32, residual epilogue, store xn
SHAPES = [(64, 64, 64, 104, 1, 0, 1, 0, 1), (64, 512, 64, 104, 1, 0, 1, 0, 1), (512, 64, 64, 104, 8, 1, 0, 1, 1),
          (128, 128, 32, 52, 1, 0, 1, 0, 1), (128, 1024, 32, 52, 1, 0, 1, 0, 1), (1024, 128, 32, 52, 8, 1, 0, 1, 1),
          (160, 160, 16, 26, 1, 0, 1, 0, 1), (160, 640, 16, 26, 1, 0, 1, 0, 1), (640, 160, 16, 26, 4, 1, 0, 1, 1),
          (256, 256, 8, 13, 1, 0, 1, 0, 1), (256, 1024, 8, 13, 1, 0, 1, 0, 1), (1024, 256, 8, 13, 4, 1, 0, 1, 1),
          (160, 160, 8, 13, 1, 0, 0, 0, 1)]
NSET, REPS = 8, 64
only = int(sys.argv[1]) if len(sys.argv) > 1 else None
for si, (Cin, Cout, H, W, gmul, act, xf, res, sx) in enumerate(SHAPES):
    if only is not None and si != only:
        continue
    P = H * W
    sets = []
    for i in range(NSET):
        x = torch.randn(B, P, Cin, device="cuda") * 0.7
        x = x if xf else x.to(torch.bfloat16)
        w = (torch.randn(Cout, 1, Cin, device="cuda") * 0.05).to(torch.bfloat16)
        xn = torch.zeros(B, P, Cin, dtype=torch.bfloat16, device="cuda")
        y = torch.zeros(B, P, Cout, dtype=torch.float32 if res else torch.bfloat16, device="cuda")
        r = torch.randn(B, P, Cout, device="cuda") if res else None
        stats = torch.zeros(B, Cin // 16, 2, device="cuda")
        L.crd_gn_stats(x.data_ptr(), xf, Cin, 0, B, P, Cin, stats.data_ptr(), None, lib.stream())
        ost = torch.zeros(B, Cout // 16, 2, dtype=torch.int64, device="cuda")
        gam, bet, bias = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda"), torch.zeros(Cout, device="cuda")
        d = lib.ConvDesc()
        d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = Cin, 0, B, H, W, Cin
        d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), Cout, 1, 1, 1, 0, H, W
        d.y, d.y_ld, d.y_coff, d.y_f32 = y.data_ptr(), Cout, 0, 1 if res else 0
        d.bias, d.stats = bias.data_ptr(), ost.data_ptr()
        if res:
            d.res, d.res_ld = r.data_ptr(), Cout
        d2 = lib.ConvDesc.from_buffer_copy(bytes(d))
        d.x, d2.x = x.data_ptr(), xn.data_ptr()
        n = lib.GnInput()
        n.x_f32, n.gmul, n.act, n.stats, n.gamma, n.beta = xf, gmul, act, stats.data_ptr(), gam.data_ptr(), bet.data_ptr()
        if sx:
            n.xn, n.xn_ld = xn.data_ptr(), Cin
        sets.append((x, w, xn, y, r, stats, ost, gam, bet, bias, d, d2, n))

    def fused(i):
        s = sets[i % NSET]
        lib.check(L.crd_gn_conv(C.byref(s[10]), C.byref(s[12]), lib.stream()), "gn_conv")

    def pair(i):
        s = sets[i % NSET]
        lib.check(L.crd_gn_apply(s[0].data_ptr(), xf, Cin, 0, B, P, Cin, s[5].data_ptr(), gmul, s[7].data_ptr(), s[8].data_ptr(), act, None,
                                 s[2].data_ptr(), 0, Cin, 0, lib.stream()), "gn_apply")
        lib.check(L.crd_conv_igemm(C.byref(s[11]), lib.stream()), "igemm")

    res_us = []
    for fn in (pair, fused):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for i in range(NSET):
                fn(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for i in range(REPS):
                    fn(i)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res_us.append(e0.elapsed_time(e1) * 1e3 / REPS)
    print(f"#{si:2d} Cin{Cin:5d} Cout{Cout:5d} {H}x{W} gmul{gmul} act{act} xf32={xf} res={res}: gn_apply+igemm {res_us[0]:7.2f} us   gn_conv {res_us[1]:7.2f} us")
