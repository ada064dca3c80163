"""Developer microbench: cost of the epilogue variants of a small 1x1 conv (Mlp.fc2 of encoder stage 3: 640 -> 160 at 16x26)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from camradepth_amd import lib
L = lib.load()
B, H, W, Cin, Cout = 8, 16, 26, 640, 160
if len(sys.argv) > 1:
    H, W, Cin, Cout = (int(v) for v in sys.argv[1:5])
NSET, REPS = 8, 64
for name in ("bf16 out", "fp32 out + residual", "+ g16 stats", "+ channel sums"):
    descs, keep = [], []
    for i in range(NSET):
        x = (torch.randn(B, H * W, Cin, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(Cout, 1, Cin, device="cuda") * 0.05).to(torch.bfloat16)
        f32 = name != "bf16 out"
        y = torch.zeros(B, H * W, Cout, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        res = torch.randn(B, H * W, Cout, device="cuda")
        scale = torch.ones(B, device="cuda")
        st, ch = torch.zeros(B, Cout // 16, 2, dtype=torch.int64, device="cuda"), torch.zeros(B, Cout, 2, dtype=torch.int64, device="cuda")   # crd_sum_t
        d = lib.ConvDesc()
        d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), Cin, 0, B, H, W, Cin
        d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), Cout, 1, 1, 1, 0, H, W
        d.y, d.y_ld, d.y_coff, d.y_f32 = y.data_ptr(), Cout, 0, 1 if f32 else 0
        if f32:
            d.res, d.res_ld, d.res_scale = res.data_ptr(), Cout, scale.data_ptr()
        if name in ("+ g16 stats", "+ channel sums"):
            d.stats = st.data_ptr()
        if name == "+ channel sums":
            d.chan_sums = ch.data_ptr()
        descs.append(d); keep.append((x, w, y, res, scale, st, ch))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        def run():
            for r in range(REPS):
                lib.check(L.crd_conv_igemm(C.byref(descs[r % NSET]), lib.stream()), "conv")
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{Cin}->{Cout} @{H}x{W}  {name:22s} {e0.elapsed_time(e1) * 1e3 / REPS:7.2f} us")
