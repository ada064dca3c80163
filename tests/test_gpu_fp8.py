"""GPU tests of the fp8 (OCP e4m3) inference path of the decoder's 3x3 ConvLayers (BASELINE.json config 5): quantisation
kernels bit-exact against torch's float8_e4m3fn conversion, the block-scaled-MFMA convolution against torch fp32 on the
de-quantised operands (so the only difference is fp32 accumulation order + the bf16 output rounding: rel-L2 < 4e-3, as for
the bf16 kernels), and the end-to-end model against the oracle's quant="fp8" mode."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_igemm import assert_close, bf, pack_w, to_pm

pytestmark = pytest.mark.gpu
E4M3 = torch.float8_e4m3fn


def _lib():
    from camradepth_amd import lib
    return lib


def dequant(u8):
    return u8.cpu().view(E4M3).float()


def test_quantisation_kernels_match_torch_e4m3():
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(0)
    rows, ld, coff, Cn = 1000, 72, 8, 40
    x = bf(torch.randn(rows, ld, generator=g) * 3)
    x[5, 9] = 1e4                       # saturates at 448 (e4m3fn has no infinity)
    x[6, 10] = -1e4
    x = bf(x)
    xd = x.to(torch.bfloat16).cuda()
    amax = torch.zeros(1, device="cuda")
    lib.check(L.crd_amax_bf16(xd.data_ptr(), rows, ld, coff, Cn, amax.data_ptr(), lib.stream()), "amax")
    assert float(amax) == float(x[:, coff:coff + Cn].abs().max())
    scale = 0.37
    y = torch.zeros(rows, 56, dtype=torch.uint8, device="cuda")
    lib.check(L.crd_quant_fp8(xd.data_ptr(), rows, ld, coff, Cn, y.data_ptr(), 56, 8, scale, lib.stream()), "quant")
    ref = (x[:, coff:coff + Cn] * (1.0 / np.float32(scale))).clamp(-448, 448).to(E4M3)
    got = y[:, 8:8 + Cn].cpu().view(E4M3)
    assert torch.equal(got.view(torch.uint8), ref.view(torch.uint8))
    assert int(y[:, :8].max()) == 0 and int(y[:, 8 + Cn:].max()) == 0
    # weights: per-output-channel scales, zero padding of the channel tail
    Co, taps, Ci, Ci16 = 21, 9, 136, 144
    w = bf(torch.randn(Co, taps, Ci, generator=g) * torch.rand(Co, 1, 1, generator=g))
    w[3] = 0
    wd = w.to(torch.bfloat16).cuda()
    w8 = torch.full((Co, taps, Ci16), 77, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(Co, device="cuda")
    lib.check(L.crd_weight_quant_fp8(wd.data_ptr(), Co, taps, Ci, Ci16, w8.data_ptr(), sc.data_ptr(), lib.stream()), "wquant")
    am = w.reshape(Co, -1).abs().max(1).values
    sref = torch.where(am > 0, am / 448.0, torch.ones_like(am))
    assert torch.allclose(sc.cpu(), sref, rtol=1e-6)
    wref = (w * (1.0 / sc.cpu()).view(Co, 1, 1)).clamp(-448, 448).to(E4M3)
    assert torch.equal(w8[:, :, :Ci].cpu(), wref.view(torch.uint8))
    assert int(w8[:, :, Ci:].max()) == 0


CASES = [
    # B, Cin(ref), H, W, Cout
    (8, 136, 90, 120, 96),      # 8-channel tail (padded to 144), 96-column tile, ragged borders
    (8, 296, 96, 128, 128),     # five 64-channel chunks, the last with 40 channels
    (8, 232, 94, 128, 64),
    (2, 48, 64, 96, 128),       # one chunk, few tiles (persistent loop shorter than the grid)
]


@pytest.mark.parametrize("case", CASES)
def test_conv3x3_fp8_matches_dequantised_reference(case):
    B, Ci, H, W, Co = case
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(sum(case))
    Ci16 = (Ci + 15) // 16 * 16
    ldx = Ci16 + 16
    x = bf(torch.randn(B, Ci, H, W, generator=g).abs() * 1.5 - 0.1)          # GELU-like: mostly positive
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5)
    xpm = to_pm(x, ld=ldx, coff=0)                                           # bf16 pixel-major [B,H,W,ldx]
    xs = float(x.abs().max()) / 448.0
    x8 = torch.zeros(B, H, W, ldx, dtype=torch.uint8, device="cuda")
    # the channels past Cin hold finite garbage in the real buffers (the next concat segment): emulate with 0x38 (= 1.0)
    x8[..., Ci:] = 0x38
    lib.check(L.crd_quant_fp8(xpm.data_ptr(), B * H * W, ldx, 0, (Ci + 7) // 8 * 8, x8.data_ptr(), ldx, 0, xs, lib.stream()), "quant")
    wp = pack_w(w, (Ci + 7) // 8 * 8)                                         # bf16 [Co][9][Cin8]
    w8 = torch.zeros(Co, 9, Ci16, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(Co, device="cuda")
    lib.check(L.crd_weight_quant_fp8(wp.data_ptr(), Co, 9, (Ci + 7) // 8 * 8, Ci16, w8.data_ptr(), ws.data_ptr(), lib.stream()), "wquant")
    y = torch.zeros(B, H, W, Co + 8, dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(B, Co // 16, 2, device="cuda")
    partial = torch.full((B * (-(-W // 32)) * (-(-H // 16)) * 4 * (Co // 16) * 2,), float("nan"), device="cuda")
    d = lib.ConvDesc()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x8.data_ptr(), ldx, 0, B, H, W, Ci16
    d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w8.data_ptr(), Co, 3, 3, 1, 1, H, W
    d.y, d.y_ld, d.y_coff = y.data_ptr(), Co + 8, 8
    d.stats, d.stats_partial, d.stats_partial_capacity = stats.data_ptr(), partial.data_ptr(), partial.numel()
    lib.check(L.crd_conv3x3_fp8(C.byref(d), ws.data_ptr(), xs, lib.stream()), "conv3x3_fp8")
    torch.cuda.synchronize()
    xq = dequant(x8[..., :Ci]).permute(0, 3, 1, 2) * xs                      # what the kernel multiplied
    wq = dequant(w8[:, :, :Ci]).reshape(Co, 3, 3, Ci).permute(0, 3, 1, 2) * ws.cpu().view(Co, 1, 1, 1)
    ref = F.conv2d(xq, wq, None, padding=1)
    got = y[..., 8:8 + Co].float().cpu().permute(0, 3, 1, 2)
    assert_close(got, ref, f"fp8 conv {case}")
    assert float(y[..., :8].float().abs().max()) == 0.0
    gq = got.reshape(B, Co // 16, 16, H * W)
    assert_close(stats.cpu(), torch.stack([gq.sum((2, 3)), (gq ** 2).sum((2, 3))], -1), "fp8 conv GroupNorm sums", rel=1e-3, elem=2e-3)
    # and the quantisation error itself against the un-quantised convolution: e4m3 has 3 mantissa bits
    full = F.conv2d(x, w, None, padding=1)
    rel = float((got - full).norm() / full.norm())
    print(f"fp8 vs bf16-operand convolution {case}: rel-L2 {rel:.4f}")
    assert rel < 0.06
