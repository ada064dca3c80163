"""Developer microbench: the encoder's small 1x1 / patch GEMMs through crd_conv_igemm, replayed from a HIP graph while
rotating over independent operand sets (so that operands come from HBM/MALL as in the real step, not from a hot L2)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from camradepth_amd import lib
import os
L = lib.load()
if os.environ.get("CRD_REGE") is not None:            # round 5: register epilogue of the 64 x 64 tiles off (0) / on (1)
    L.crd_tune_igemm_reg_epilogue(int(os.environ["CRD_REGE"]))
if os.environ.get("CRD_NARROW") is not None:
    L.crd_tune_pw_narrow(int(os.environ["CRD_NARROW"]))
B = 8
SHAPES = [  # Cin, Cout, H, W, k, stride, gather_mode   (the encoder's Mlp.fc1 / fc2 and their data gradients, base model)
    (64, 512, 64, 104, 1, 1, 0), (512, 64, 64, 104, 1, 1, 0), (512, 64, 64, 104, 1, 1, 1), (64, 512, 64, 104, 1, 1, 1),
    (128, 1024, 32, 52, 1, 1, 0), (1024, 128, 32, 52, 1, 1, 0), (1024, 128, 32, 52, 1, 1, 1), (128, 1024, 32, 52, 1, 1, 1),
    (160, 640, 16, 26, 1, 1, 0), (640, 160, 16, 26, 1, 1, 0), (640, 160, 16, 26, 1, 1, 1), (160, 640, 16, 26, 1, 1, 1),
    (256, 1024, 8, 13, 1, 1, 0), (1024, 256, 8, 13, 1, 1, 0), (128, 128, 32, 52, 1, 1, 0), (160, 160, 16, 26, 1, 1, 0),
    (160, 160, 16, 26, 2, 2, 0), (128, 128, 32, 52, 4, 4, 0)]
NSET, REPS = 12, 120
STATS = len(sys.argv) > 1 and sys.argv[1] == "stats"
for Cin, Cout, H, W, k, s, mode in SHAPES:
    OH, OW = H // s, W // s
    sets, descs = [], []
    for i in range(NSET):
        x = (torch.randn(B, H * W, Cin, device="cuda") * 0.5).to(torch.bfloat16)
        w = (torch.randn(Cout, k * k, Cin, device="cuda") * 0.05).to(torch.bfloat16)
        y = torch.zeros(B, OH * OW, Cout, dtype=torch.bfloat16, device="cuda")
        d = lib.ConvDesc()
        d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), Cin, 0, B, H, W, Cin
        d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), Cout, k, k, s, 0, OH, OW
        d.gather_mode = mode
        d.y, d.y_ld, d.y_coff, d.y_f32 = y.data_ptr(), Cout, 0, 0
        st_ = torch.zeros(B, Cout // 16, 2, dtype=torch.int64, device="cuda")
        if STATS and Cout % 16 == 0:
            d.stats = st_.data_ptr()
        sets.append((x, w, y, st_)); descs.append(d)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        def run():
            for r in range(REPS):
                lib.check(L.crd_conv_igemm(C.byref(descs[r % NSET]), lib.stream()), "conv")
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REPS
    byts = B * (H * W * Cin + OH * OW * Cout) * 2 + Cout * k * k * Cin * 2
    if os.environ.get("CRD_REGE_COUNT"):
        print("  register-epilogue launches so far:", L.crd_tune_igemm_reg_epilogue(-1))
    print(f"Cin{Cin:5d} Cout{Cout:5d} {H}x{W} k{k} s{s} mode{mode}: {us:7.2f} us  {2.0 * B * OH * OW * Cout * Cin * k * k / us / 1e6:6.1f} TF/s  {byts / us / 1e6:5.2f} TB/s")
