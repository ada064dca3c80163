"""Developer microbench: one conv layer through crd_conv_igemm, N launches (for rocprofv3 --pmc runs)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from camradepth_amd import lib
import os
B, H, W = int(os.environ.get("B", 8)), int(os.environ.get("H", 256)), int(os.environ.get("W", 416))
Cin, Cout = int(os.environ.get("CIN", 304)), int(os.environ.get("COUT", 128))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = lib.load()
x = (torch.randn(B, H * W, Cin, device="cuda") * 0.5).to(torch.bfloat16)
w = (torch.randn(Cout, 9, Cin, device="cuda") / (9 * Cin) ** 0.5).to(torch.bfloat16)
y = torch.zeros(B, H * W, Cout, dtype=torch.bfloat16, device="cuda")
stats = torch.zeros(B, Cout // 16, 2, dtype=torch.int64, device="cuda")
d = lib.ConvDesc()
d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), Cin, 0, B, H, W, Cin
d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), Cout, 3, 3, 1, 1, H, W
d.gather_mode = mode
d.y, d.y_ld, d.y_coff, d.y_f32 = y.data_ptr(), Cout, 0, 0
d.stats = stats.data_ptr() if len(sys.argv) <= 3 else None
partial = torch.zeros(B * (-(-H * W // 64)) * (Cout // 16) * 2, device="cuda")
if d.stats:
    d.stats_partial, d.stats_partial_capacity = partial.data_ptr(), partial.numel()
d.accumulate = int(os.environ.get("ACC", 0))
for _ in range(3):
    lib.check(L.crd_conv_igemm(C.byref(d), lib.stream()), "conv")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    L.crd_conv_igemm(C.byref(d), lib.stream())
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"conv {Cin}->{Cout} 3x3 @{H}x{W} B{B}: {ms:.3f} ms, {2.0 * B * H * W * Cout * Cin * 9 / ms / 1e9:.0f} TFLOP/s")

if hasattr(L, "crd_dbg_conv3_prof"):      # library built with -DCRD_CONV3_PROF (tools/build_prof.sh)
    buf = (C.c_ulonglong * 32)()
    L.crd_dbg_conv3_prof(buf)
    steps = 9 * ((Cin + 31) // 32)
    names = ["dma issue", "reads k0", "mfma k0", "reads k1", "mfma k1", "vm wait", "barrier"]
    for w in range(4):
        print(f"  wave {w}: " + "  ".join(f"{n} {buf[w * 8 + k] / steps:5.0f}" for k, n in enumerate(names)) + "  (cycles per step)")

if hasattr(L, "crd_dbg_conv3p_prof"):     # persistent kernel built with -DCRD_CONV3_PROF
    buf = (C.c_ulonglong * 4)()
    L.crd_dbg_conv3p_prof(buf, 1)
    L.crd_conv_igemm(C.byref(d), lib.stream()); torch.cuda.synchronize()
    L.crd_dbg_conv3p_prof(buf, 0)
    n = max(buf[2], 1)
    print(f"  persistent kernel, wave 0 of workgroup 0: {buf[2]} tiles, main loop {buf[0] / n:.0f} cycles / tile, epilogue {buf[1] / n:.0f} cycles / tile")
