"""Per-phase shader-clock stamps of k_mlp_fwd (build with -DCRD_MLP_PROF: tools/prof_mlp.sh).  Runs one launch per stage shape."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camradepth_amd import lib
from tests.util import to_stat, zsum
L = lib.load()
raw = C.CDLL(lib.LIB_PATH)
for (B, H, W, Cs, hid) in ((8, 16, 26, 160, 640), (8, 8, 13, 256, 1024)):
    N = H * W
    slabs = hid // 64
    x1 = torch.randn(B, N, Cs, device="cuda")
    v = x1.double().reshape(B, N, Cs // 16, 16)
    st2 = to_stat(torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1).float()).cuda()
    f = lambda *s: torch.randn(*s, device="cuda") * 0.1
    w1, w2 = f(hid, Cs).to(torch.bfloat16), f(Cs, hid).to(torch.bfloat16)
    ts = dict(g0=f(Cs) + 1, b0=f(Cs), b1=f(hid), g1=f(hid) + 1, b1n=f(hid), w9=f(9, hid), bd=f(hid), g2=f(hid) + 1, b2n=f(hid))
    h = [torch.zeros(B, N, hid, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    xn = torch.zeros(B, N, Cs, dtype=torch.bfloat16, device="cuda")
    s1, s2 = zsum(B, hid // 16, 2), zsum(B, hid // 16, 2)
    part = torch.zeros(slabs, B, N, Cs, device="cuda")
    d = lib.MlpDesc()
    d.x1, d.x1_stats, d.norm_gamma, d.norm_beta = x1.data_ptr(), st2.data_ptr(), ts["g0"].data_ptr(), ts["b0"].data_ptr()
    d.w_fc1, d.b_fc1, d.norm1_gamma, d.norm1_beta = w1.data_ptr(), ts["b1"].data_ptr(), ts["g1"].data_ptr(), ts["b1n"].data_ptr()
    d.w9, d.b_dw, d.norm2_gamma, d.norm2_beta = ts["w9"].data_ptr(), ts["bd"].data_ptr(), ts["g2"].data_ptr(), ts["b2n"].data_ptr()
    d.w_fc2 = w2.data_ptr()
    d.xn, d.h1, d.h2, d.h3 = xn.data_ptr(), h[0].data_ptr(), h[1].data_ptr(), h[2].data_ptr()
    d.h1_stats, d.h2_stats, d.fc2_partials = s1.data_ptr(), s2.data_ptr(), part.data_ptr()
    d.B, d.H, d.W, d.C, d.hidden = B, H, W, Cs, hid
    for _ in range(3):
        lib.check(L.crd_mlp_fwd(C.byref(d), lib.stream()), "mlp")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L.crd_mlp_fwd(C.byref(d), lib.stream())
    e1.record()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    raw.crd_dbg_mlp_prof(out)
    st = [out[i] for i in range(8)]
    names = ["phase0 loads", "fc1 chunks", "h1 epilogue+stats", "norm1", "dwconv", "norm2+gelu", "fc2"]
    print(f"shape {B}x{H}x{W} C{Cs} hid{hid}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (back to back)")
    for n, a, b in zip(names, st, st[1:]):
        print(f"   {n:20s} {b - a:8d} cycles")
    print(f"   total                {st[7] - st[0]:8d} cycles (100 MHz counter ticks x ~21-24 = shader clocks if s_memrealtime; readcyclecounter = shader clocks)")
