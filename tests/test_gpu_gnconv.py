"""GPU parity of crd_gn_conv -- GroupNorm (+ exact GELU) applied while the A operand of the pointwise / patch GEMM is
loaded -- against torch fp32: F.group_norm -> [F.gelu] -> bf16 rounding (the tensor the unfused path stores) -> F.conv2d on
bf16-rounded weights.  Shapes are the encoder's (simplified_attention.py:34-43,96-100,142-145): fc1 behind Block.norm2
(fp32 residual stream in), attn.sr patches behind Block.norm1, attn.k behind attn.norm, fc2 behind Mlp.norm2 (group =
gmul 16-channel slabs, Q1) + GELU with the residual / DropPath / statistics epilogue.  Tolerance as for crd_conv_igemm:
one bf16 rounding of the output, rel-L2 < 4e-3."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_igemm import assert_close, bf, pack_w
from tests.util import sval, to_stat, zsum

pytestmark = pytest.mark.gpu


def _lib():
    from camradepth_amd import lib
    return lib


def slab_sums(x):
    """[B,C,H,W] fp32 -> raw (sum, sumsq) per 16-channel slab, [B, C/16, 2] (what a producer's stats epilogue leaves)."""
    B, Cc = x.shape[:2]
    v = x.reshape(B, Cc // 16, -1).double()
    return torch.stack([v.sum(-1), (v * v).sum(-1)], -1).float()


# B, Cin, H, W, Cout, k (= stride), gmul, act, x_f32, epilogue
CASES = [
    (2, 64, 16, 24, 64, 1, 1, 0, 1, "plain"),             # q stage 1 behind Block.norm1 (fp32 stream in, XN stored)
    (2, 64, 16, 24, 512, 1, 1, 0, 1, "stats"),            # fc1 stage 1 behind Block.norm2: one K-slab, 4 column chunks
    (2, 160, 6, 10, 640, 1, 1, 0, 1, "stats"),            # fc1 stage 3, ragged K (2.5 slabs) and M
    (2, 16, 16, 24, 64, 2, 1, 0, 1, "stats"),             # 2x2 patches, resident (K = 64)
    (2, 64, 16, 24, 64, 8, 1, 0, 0, "stats"),             # 8x8 patches of a bf16 tensor: streaming, K = 4096
    (2, 160, 4, 13, 160, 1, 1, 0, 0, "plain"),            # attn.k behind attn.norm (bf16 in)
    (2, 256, 4, 13, 256, 1, 1, 0, 1, "plain"),            # q stage 4: four fp32 K-slabs resident
    (2, 512, 16, 24, 64, 1, 8, 1, 0, "res"),              # fc2 stage 1: Mlp.norm2 (gmul 8) + GELU, residual + sums; streaming
    (2, 640, 6, 10, 160, 1, 4, 1, 0, "res"),              # fc2 stage 3: two column tiles
    (2, 1024, 5, 7, 256, 1, 4, 1, 0, "res"),              # fc2 stage 4, 16 K-slabs
    (8, 128, 32, 52, 1024, 1, 1, 0, 1, "stats"),          # fc1 stage 2 at the benchmark size
    (8, 512, 64, 104, 64, 1, 8, 1, 0, "res"),             # fc2 stage 1 at the benchmark size
    (2, 320, 6, 10, 96, 1, 1, 0, 1, "stats"),             # fp32 input with five K-slabs (the register path has no K limit)
    (1, 64, 7, 9, 40, 1, 1, 0, 1, "plain"),               # ragged rows and columns on a single workgroup
    (8, 64, 64, 104, 512, 1, 1, 0, 1, "stats"),           # fc1 stage 1 at the benchmark size (the wide pointwise kernel, 4 tiles per workgroup)
    (3, 128, 9, 11, 1024, 1, 1, 0, 1, "plain"),           # wide kernel: ragged rows (99 pixels: two tiles, the second 35 rows), no sums
    (2, 160, 16, 26, 640, 1, 1, 0, 1, "stats"),           # fc1 stage 3 at the benchmark size: 128-column workgroups, 6.5 tiles
    (5, 64, 20, 30, 256, 1, 1, 0, 1, "stats"),            # wide kernel: one column block, odd batch, 600 pixels
    (8, 1024, 32, 52, 128, 1, 8, 1, 0, "res"),            # fc2 stage 2 at the benchmark size: the narrow streaming kernel (round 5), 2 tiles per workgroup
    (3, 1024, 8, 12, 128, 1, 8, 1, 0, "res"),             # ... three tiles per sample, odd batch
    (2, 512, 16, 24, 64, 1, 8, 1, 0, "plain"),            # ... its bf16-output form behind GroupNorm + GELU
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[1]}to{c[4]}k{c[5]}_{c[9]}_{c[2]}x{c[3]}" for c in CASES])
def test_gn_conv_matches_groupnorm_then_conv(case):
    B, Cin, H, W, Cout, k, gmul, act, x_f32, epi = case
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.3
    if not x_f32:
        x = bf(x)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(Cin, generator=g), 0.1 * torch.randn(Cin, generator=g)
    groups = Cin // (16 * gmul)
    xn = F.group_norm(x, groups, gamma, beta, 1e-5)
    if act:
        xn = F.gelu(xn)
    xn_b = bf(xn)
    ref = F.conv2d(xn_b, bf(w), bias, stride=k)
    OH, OW = H // k, W // k
    xpm = x.permute(0, 2, 3, 1).contiguous()
    xpm = (xpm if x_f32 else xpm.to(torch.bfloat16)).cuda()
    stats = to_stat(slab_sums(x)).cuda()
    wp = pack_w(w)
    xn_out = torch.zeros(B, H * W, Cin, dtype=torch.bfloat16, device="cuda")
    d, n = lib.ConvDesc(), lib.GnInput()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = xpm.data_ptr(), Cin, 0, B, H, W, Cin
    d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = wp.data_ptr(), Cout, k, k, k, 0, OH, OW
    bias_d = bias.cuda()
    d.bias = bias_d.data_ptr()
    n.x_f32, n.gmul, n.act = x_f32, gmul, act
    gam_d, bet_d = gamma.cuda(), beta.cuda()
    n.stats, n.gamma, n.beta = stats.data_ptr(), gam_d.data_ptr(), bet_d.data_ptr()
    n.xn, n.xn_ld = xn_out.data_ptr(), Cin
    ostats = zsum(B, Cout // 16, 2)
    if epi == "res":
        y = torch.zeros(B, OH * OW, Cout, device="cuda")
        res = torch.randn(B, OH * OW, Cout, generator=g)
        scale = torch.tensor([1.0 / 0.9, 0.0] * B)[:B]
        res_d, scale_d = res.cuda(), scale.cuda()
        chan = zsum(B, Cout, 2)
        d.y, d.y_ld, d.y_f32 = y.data_ptr(), Cout, 1
        d.res, d.res_ld, d.res_scale = res_d.data_ptr(), Cout, scale_d.data_ptr()
        d.stats, d.chan_sums = ostats.data_ptr(), chan.data_ptr()
        ref_y = res + scale.view(B, 1, 1) * bf(ref).permute(0, 2, 3, 1).reshape(B, OH * OW, Cout)
    else:
        y = torch.zeros(B, OH * OW, Cout, dtype=torch.bfloat16, device="cuda")
        d.y, d.y_ld, d.y_f32 = y.data_ptr(), Cout, 0
        if epi == "stats":
            d.stats = ostats.data_ptr()
        ref_y = ref.permute(0, 2, 3, 1).reshape(B, OH * OW, Cout)
    n_narrow = L.crd_tune_pw_narrow(-1)
    lib.check(L.crd_gn_conv(C.byref(d), C.byref(n), lib.stream()), "crd_gn_conv")
    torch.cuda.synchronize()
    # fc2 of encoder stages 1-2 behind Mlp.norm2 + GELU is the narrow streaming kernel's (csrc/pw_narrow.hip), not the generic tiles'
    narrow = (Cin, Cout) in ((512, 64), (1024, 128)) and act == 1 and k == 1 and (H * W) % 32 == 0 and epi in ("res", "plain")
    assert L.crd_tune_pw_narrow(-1) - n_narrow == (1 if narrow else 0), "kernel selection"
    assert_close(y.float().cpu(), ref_y, "gn_conv output")
    # the normalised operand is stored for the weight gradient: bf16 of the torch value, up to one ulp where the fp32
    # statistics (sum / sum of squares here, two-pass in torch) move a value across a rounding boundary
    xn_ref = xn_b.permute(0, 2, 3, 1).reshape(B, H * W, Cin)
    assert_close(xn_out.float().cpu(), xn_ref, "stored normalised operand", rel=3e-3, elem=1.6e-2)
    if epi in ("stats", "res"):
        stored = y.float().cpu().reshape(B, OH * OW, Cout).permute(0, 2, 1).reshape(B, Cout, OH, OW)
        assert_close(sval(ostats), slab_sums(stored), "output GroupNorm sums", rel=2e-3, elem=5e-3)
    if epi == "res":
        v = stored.reshape(B, Cout, -1).double()
        cref = torch.stack([v.sum(-1), (v * v).sum(-1)], -1).float()
        assert_close(sval(chan), cref, "output channel sums", rel=2e-3, elem=5e-3)


def test_gn_conv_rejects_what_it_does_not_cover():
    lib = _lib()
    L = lib.load()
    t = torch.zeros(64, device="cuda")
    d, n = lib.ConvDesc(), lib.GnInput()
    d.x = d.w = d.y = t.data_ptr()
    n.stats = n.gamma = n.beta = t.data_ptr()
    d.B, d.IH, d.IW, d.Cin, d.x_ld, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = 1, 8, 8, 64, 64, 64, 3, 3, 1, 1, 8, 8
    n.gmul = 1
    assert L.crd_gn_conv(C.byref(d), C.byref(n), lib.stream()) == -2          # overlapping taps: unsupported
    d.KH = d.KW = d.stride = 1
    d.pad = 0
    n.gmul = 3
    assert L.crd_gn_conv(C.byref(d), C.byref(n), lib.stream()) == -1          # 4 slabs do not split into groups of 3
