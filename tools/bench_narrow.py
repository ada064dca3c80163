"""Developer tool: the narrow streaming pointwise kernel (csrc/pw_narrow.hip) against the generic tiles it replaces, on the encoder's
stage-1 / stage-2 shapes at B = 8: Mlp.fc2 on activated rows (fp32 residual output + sums), Mlp.fc2 behind GroupNorm + GELU
(crd_gn_conv; the pair it replaces: crd_gn_apply + crd_conv_igemm), and the data gradient of Mlp.fc1 with the fused reduce of
Block.norm2's backward.  Graph-replayed over rotating operand sets (cold-ish caches), us per launch.   python tools/bench_narrow.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camradepth_amd import lib  # noqa: E402

L = lib.load()
ROT = 4


def timeit(fn, reps=20):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for i in range(ROT):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for r in range(reps):
                fn(r % ROT)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def case(B, H, W, K, N):
    P = H * W
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = [torch.randn(B, P, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(ROT)]
    w = (torch.randn(N, 1, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N + 1, device="cuda", generator=g)[1:]                  # 4-byte aligned only, like a parameter view
    res = [torch.randn(B, P, N, device="cuda", generator=g) for _ in range(ROT)]
    yf = [torch.zeros(B, P, N, device="cuda") for _ in range(ROT)]
    yb = [torch.zeros(B, P, N, dtype=torch.bfloat16, device="cuda") for _ in range(ROT)]
    xn = [torch.zeros(B, P, K, dtype=torch.bfloat16, device="cuda") for _ in range(ROT)]
    scale = torch.ones(B, device="cuda")
    stats = torch.zeros(B, N // 16, 2, dtype=torch.int64, device="cuda")
    chan = torch.zeros(B, N, 2, dtype=torch.int64, device="cuda")
    gstats = torch.zeros(B, K // 16, 2, dtype=torch.int64, device="cuda")
    lib.check(L.crd_gn_stats(xs[0].data_ptr(), 0, K, 0, B, P, K, gstats.data_ptr(), None, lib.stream()), "stats")
    gamma, beta = torch.ones(K + 1, device="cuda")[1:], torch.zeros(K + 1, device="cuda")[1:]
    rstats = torch.zeros(B, N // 16, 2, dtype=torch.int64, device="cuda")
    lib.check(L.crd_gn_stats(res[0].data_ptr(), 1, N, 0, B, P, N, rstats.data_ptr(), None, lib.stream()), "stats")
    rgam, rbet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    rr = torch.zeros(B * N * 2 + B * (N // 16) * 2, dtype=torch.int64, device="cuda")
    keep = []

    def desc(i, f32, dgrad, red):
        d = lib.ConvDesc()
        d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = xs[i].data_ptr(), K, 0, B, H, W, K
        d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), N, 1, 1, 1, 0, H, W
        d.gather_mode = 1 if dgrad else 0
        if f32:
            d.y, d.y_ld, d.y_f32 = yf[i].data_ptr(), N, 1
            d.res, d.res_ld, d.res_scale, d.bias = res[i].data_ptr(), N, scale.data_ptr(), bias.data_ptr()
            d.stats, d.chan_sums = stats.data_ptr(), chan.data_ptr()
        else:
            d.y, d.y_ld, d.y_f32 = yb[i].data_ptr(), N, 0
        if red:
            d.red_x, d.red_x_ld, d.red_gmul, d.red_act, d.red_x_f32 = res[i].data_ptr(), N, 1, 0, 1
            d.red_stats, d.red_gamma, d.red_beta, d.red_r = rstats.data_ptr(), rgam.data_ptr(), rbet.data_ptr(), rr.data_ptr()
        keep.append(d)
        return d

    def gn(i, with_xn):
        n = lib.GnInput()
        n.x_f32, n.gmul, n.act = 0, K // (N // 16) // 16, 1
        n.stats, n.gamma, n.beta = gstats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
        if with_xn:
            n.xn, n.xn_ld = xn[i].data_ptr(), K
        keep.append(n)
        return n
    out = {}
    for on in (1, 0):
        L.crd_tune_pw_narrow(2 if on else 0)         # 2: also the K = 1024 launches without a GroupNorm in front (off in the product)
        n0 = L.crd_tune_pw_narrow(-1)
        fwd = [desc(i, True, False, False) for i in range(ROT)]
        out[("fc2 fwd (activated rows in, fp32 residual + sums)", on)] = timeit(lambda i: lib.check(L.crd_conv_igemm(C.byref(fwd[i]), lib.stream())))
        dg = [desc(i, False, True, True) for i in range(ROT)]
        out[("fc1 dgrad + Block.norm2 reduce", on)] = timeit(lambda i: lib.check(L.crd_conv_igemm(C.byref(dg[i]), lib.stream())))
        dgp = [desc(i, False, True, False) for i in range(ROT)]
        out[("fc1 dgrad, plain", on)] = timeit(lambda i: lib.check(L.crd_conv_igemm(C.byref(dgp[i]), lib.stream())))
        for with_xn in (True, False):
            gns = [gn(i, with_xn) for i in range(ROT)]
            out[(f"fc2 behind GN+GELU (crd_gn_conv{', H3 stored' if with_xn else ''})", on)] = timeit(
                lambda i: lib.check(L.crd_gn_conv(C.byref(fwd[i]), C.byref(gns[i]), lib.stream())))
        used = L.crd_tune_pw_narrow(-1) - n0
        print(f"  narrow {'on ' if on else 'off'}: {used} launches took it")
    L.crd_tune_pw_narrow(1)
    for k in sorted({k for k, _ in out}):
        print(f"  {k:70s} narrow {out[(k, 1)]:6.1f} us   generic {out[(k, 0)]:6.1f} us")
    mb = B * P * K * 2 / 1e6
    print(f"  (hidden tensor {mb:.1f} MB: {mb / 6.3e3 * 1e3:.1f} us at 6.3 TB/s)")


for shape in ((8, 64, 104, 512, 64), (8, 32, 52, 1024, 128)):
    print(f"B {shape[0]}  {shape[1]} x {shape[2]}  K {shape[3]} -> N {shape[4]}")
    case(*shape)
