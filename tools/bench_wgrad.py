"""Developer microbench: one weight-gradient launch (D4 layer 2 shape)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from camradepth_amd import lib
import os
B, H, W = 8, 256, 416
Cin, Cout = int(os.environ.get("CIN", 304)), int(os.environ.get("COUT", 128))
L = lib.load()
x = (torch.randn(B, H * W, Cin, device="cuda") * 0.5).to(torch.bfloat16)
dy = (torch.randn(B, H * W, Cout, device="cuda") * 0.5).to(torch.bfloat16)
dw = torch.zeros(Cout, 9, Cin, dtype=torch.int64, device="cuda")   # crd_sum_t
d = lib.WgradDesc()
d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), Cin, 0, B, H, W, Cin
d.dy, d.dy_ld, d.dy_coff, d.OH, d.OW, d.Cout = dy.data_ptr(), Cout, 0, H, W, Cout
d.KH, d.KW, d.stride, d.pad, d.dw, d.dbias = 3, 3, 1, 1, dw.data_ptr(), None
d.wg_budget = int(os.environ.get("BUDGET", 0))
S = L.crd_conv_wgrad_splits(C.byref(d))
if S > 0 and os.environ.get("PARTS", "1") == "1":          # per-split copies (what the training plan uses)
    parts = torch.empty(S, Cout, 9, Cin, device="cuda")
    d.dw_partials, d.dw_partial_capacity = parts.data_ptr(), S
for _ in range(3):
    lib.check(L.crd_conv_wgrad(C.byref(d), lib.stream()), "wgrad")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.crd_conv_wgrad(C.byref(d), lib.stream())
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"wgrad {Cin}->{Cout} budget {d.wg_budget} splits {S}: {ms:.3f} ms, {2.0 * B * H * W * Cout * Cin * 9 / ms / 1e9:.0f} TFLOP/s")
