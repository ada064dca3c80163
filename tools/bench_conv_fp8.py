"""Developer microbench: one decoder 3x3 ConvLayer through crd_conv3x3_fp8 beside the bf16 kernel (same shape)."""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import torch
from camradepth_amd import lib

B, H, W = int(os.environ.get("B", 8)), int(os.environ.get("H", 256)), int(os.environ.get("W", 416))
Cin, Cout = int(os.environ.get("CIN", 304)), int(os.environ.get("COUT", 128))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = lib.load()
x8 = torch.randint(0, 256, (B, H * W, Cin), dtype=torch.uint8, device="cuda") & 0x3F | 0x20     # finite e4m3 of modest size
w8 = torch.randint(0, 256, (Cout, 9, Cin), dtype=torch.uint8, device="cuda") & 0xBF | 0x20
ws = torch.ones(Cout, device="cuda") * 1e-3
y = torch.zeros(B, H * W, Cout, dtype=torch.bfloat16, device="cuda")
stats = torch.zeros(B, Cout // 16, 2, dtype=torch.int64, device="cuda")
partial = torch.zeros(B * (-(-W // 32)) * (-(-H // 16)) * 4 * (Cout // 16) * 2, device="cuda")
d = lib.ConvDesc()
d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x8.data_ptr(), Cin, 0, B, H, W, Cin
d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w8.data_ptr(), Cout, 3, 3, 1, 1, H, W
d.y, d.y_ld, d.y_coff = y.data_ptr(), Cout, 0
d.stats, d.stats_partial, d.stats_partial_capacity = stats.data_ptr(), partial.data_ptr(), partial.numel()
for _ in range(3):
    lib.check(L.crd_conv3x3_fp8(C.byref(d), ws.data_ptr(), 1.0, lib.stream()), "conv")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    L.crd_conv3x3_fp8(C.byref(d), ws.data_ptr(), 1.0, lib.stream())
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"fp8 conv {Cin}->{Cout} 3x3 @{H}x{W} B{B}: {ms:.3f} ms, {2.0 * B * H * W * Cout * Cin * 9 / ms / 1e9:.0f} TFLOP/s")
