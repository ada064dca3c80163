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


# ---- crd_gn_bwd_conv (round 6): the backward twin -- GroupNorm-backward APPLY in the operand load of the consuming data-gradient GEMM
# Ci = the GroupNorm's channels = K of the GEMM, Co = output columns.  B, Ci, H, W, Co, gmul, act, gx_f32, mode
BWD_CASES = [
    (2, 512, 16, 24, 64, 1, 0, 0, "red"),        # Mlp.norm1 in front of fc1's data gradient, stage 1 (+ Block.norm2's fused reduce)
    (2, 640, 16, 26, 160, 1, 0, 0, "red"),       # ... stage 3 at the benchmark grid: ragged K (10 slabs), 160 columns on 64-wide tiles
    (8, 1024, 32, 52, 128, 1, 0, 0, "red"),      # ... stage 2 at the benchmark size
    (2, 1024, 8, 13, 256, 1, 0, 0, "plain"),     # ... stage 4
    (3, 1024, 7, 9, 256, 1, 0, 0, "acc"),        # ragged rows, odd batch, accumulating output
    (2, 64, 8, 13, 64, 1, 0, 0, "scatter8"),     # attn.norm in front of the sr patch scatter: stage 1 (8 x 8 patches)
    (2, 128, 8, 13, 128, 1, 0, 0, "scatter4"),   # stage 2
    (8, 160, 8, 13, 160, 1, 0, 0, "scatter2"),   # stage 3 at the benchmark size
    (2, 160, 8, 13, 160, 1, 0, 0, "scatter2acc"),
    (2, 512, 12, 20, 64, 8, 1, 0, "plain"),      # Mlp.norm2 + GELU (groups of 8 slabs)
    (2, 64, 9, 11, 512, 1, 0, 1, "plain"),       # fp32 GroupNorm input (the residual stream), 128-wide column tiles
    (8, 160, 16, 26, 640, 1, 0, 1, "stats"),     # ... with output sums
]


@pytest.mark.parametrize("case", BWD_CASES, ids=[f"{c[1]}to{c[4]}_g{c[5]}a{c[6]}f{c[7]}_{c[8]}_{c[0]}x{c[2]}x{c[3]}" for c in BWD_CASES])
def test_gn_bwd_conv_matches_apply_then_conv(case):
    """crd_gn_bwd_conv against the launches it replaces -- crd_gn_bwd_apply, then crd_conv_igemm on the stored gradient -- and
    against torch autograd through F.group_norm (+ F.gelu): same product (one bf16 rounding of dx may differ), same stored dx, same
    GroupNorm parameter gradients, same fused-reduce sums / output statistics."""
    from tests.test_gpu_igemm import run_conv, to_pm
    from tests.util import gval
    B, Ci, H, W, Co, gmul, act, gxf32, mode = case
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(Ci + 3 * Co + act)
    P = H * W
    xg = torch.randn(B, Ci, H, W, generator=g) * 1.4 + 0.25
    if not gxf32:
        xg = bf(xg)
    dy = bf(torch.randn(B, Ci, H, W, generator=g) * 0.7)
    gamma, beta = 1 + 0.2 * torch.randn(Ci, generator=g), 0.1 * torch.randn(Ci, generator=g)
    scatter = mode.startswith("scatter")
    pk = int(mode[7]) if scatter else 1
    N = Co * pk * pk
    w = bf(torch.randn(N, Ci, generator=g) / Ci ** 0.5)                    # rows = output columns, K contiguous
    # torch reference of dx
    xr = xg.clone().requires_grad_(True)
    yn = F.group_norm(xr, Ci // (16 * gmul), gamma, beta, 1e-5)
    if act:
        yn = F.gelu(yn)
    yn.backward(dy)
    dx_ref = xr.grad
    xpm = xg.permute(0, 2, 3, 1).reshape(B, P, Ci).contiguous()
    xpm = (xpm if gxf32 else xpm.to(torch.bfloat16)).cuda()
    dypm = to_pm(dy).reshape(B, P, Ci)
    gam_d, bet_d, wd = gamma.cuda(), beta.cuda(), w.to(torch.bfloat16).cuda()
    stats = zsum(B, Ci // 16, 2)
    lib.check(L.crd_gn_stats(xpm.data_ptr(), gxf32, Ci, 0, B, P, Ci, stats.data_ptr(), None, lib.stream()), "gn_stats")
    G = Ci // (16 * gmul)
    r = zsum(B * Ci * 2 + B * G * 2)
    lib.check(L.crd_gn_bwd_reduce(xpm.data_ptr(), gxf32, Ci, 0, dypm.data_ptr(), 0, Ci, 0, B, P, Ci, stats.data_ptr(), gmul,
                                  gam_d.data_ptr(), bet_d.data_ptr(), act, None, r.data_ptr(), None, 0, lib.stream()), "gn_bwd_reduce")
    # the pair it replaces
    dga0, dbe0 = torch.zeros(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    dx0 = torch.zeros(B, P, Ci, dtype=torch.bfloat16, device="cuda")
    lib.check(L.crd_gn_bwd_apply(xpm.data_ptr(), gxf32, Ci, 0, dypm.data_ptr(), 0, Ci, 0, B, P, Ci, stats.data_ptr(), gmul, gam_d.data_ptr(),
                                 bet_d.data_ptr(), act, None, r.data_ptr(), dga0.data_ptr(), dbe0.data_ptr(), dx0.data_ptr(), 0, Ci, 0, 0,
                                 None, 0, None, lib.stream()), "gn_bwd_apply")
    acc = mode.endswith("acc")
    YH, YW = H * pk, W * pk
    base = bf(torch.randn(B, YH * YW, Co, generator=g)) if acc else torch.zeros(B, YH * YW, Co)
    y0, y1 = base.to(torch.bfloat16).cuda(), base.to(torch.bfloat16).cuda()
    red0 = red1 = None
    ost0, ost1 = zsum(B, Co // 16, 2), zsum(B, Co // 16, 2)
    kw = {}
    if mode == "red":          # reduce phase of the NEXT GroupNorm's backward (input: an fp32 residual-stream tensor) on the product
        rx = (torch.randn(B, P, Co, generator=g) * 1.2).cuda()
        rst = zsum(B, Co // 16, 2)
        lib.check(L.crd_gn_stats(rx.data_ptr(), 1, Co, 0, B, P, Co, rst.data_ptr(), None, lib.stream()), "gn_stats")
        rg, rb = (1 + 0.1 * torch.randn(Co, generator=g)).cuda(), (0.1 * torch.randn(Co, generator=g)).cuda()
        red0, red1 = zsum(B * Co * 2 + B * (Co // 16) * 2), zsum(B * Co * 2 + B * (Co // 16) * 2)
    if scatter:
        kw = dict(out_mode=1, patch_k=pk, patch_c=Co)
    run_conv(dx0.view(B, H, W, Ci), Ci, 0, B, H, W, Ci, wd, N, 1, 1, 1, 0, H, W, y0.view(B, YH, YW, Co), Co, 0, accumulate=int(acc),
             stats=ost0 if mode == "stats" else None, red=(rx, rst, rg, rb, 1, 0, red0) if red0 is not None else None, **kw)
    # the fused launch
    d, n = lib.ConvDesc(), lib.GnBwdInput()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = dypm.data_ptr(), Ci, 0, B, H, W, Ci
    d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = wd.data_ptr(), N, 1, 1, 1, 0, H, W
    d.y, d.y_ld, d.y_coff, d.y_f32, d.accumulate = y1.data_ptr(), Co, 0, 0, int(acc)
    if scatter:
        d.out_mode, d.patch_k, d.patch_c = 1, pk, Co
    if mode == "stats":
        d.stats = ost1.data_ptr()
    if red1 is not None:
        d.red_x, d.red_x_ld, d.red_gmul, d.red_act, d.red_x_f32 = rx.data_ptr(), Co, 1, 0, 1
        d.red_stats, d.red_gamma, d.red_beta, d.red_r = rst.data_ptr(), rg.data_ptr(), rb.data_ptr(), red1.data_ptr()
    dga1, dbe1 = torch.zeros(Ci, device="cuda"), torch.zeros(Ci, device="cuda")
    dx1 = torch.zeros(B, P, Ci, dtype=torch.bfloat16, device="cuda")
    n.gx, n.gx_f32, n.gx_ld, n.gmul, n.act = xpm.data_ptr(), gxf32, Ci, gmul, act
    n.stats, n.gamma, n.beta, n.r = stats.data_ptr(), gam_d.data_ptr(), bet_d.data_ptr(), r.data_ptr()
    n.dx, n.dx_ld, n.dgamma, n.dbeta = dx1.data_ptr(), Ci, dga1.data_ptr(), dbe1.data_ptr()
    lib.check(L.crd_gn_bwd_conv(C.byref(d), C.byref(n), lib.stream()), "crd_gn_bwd_conv")
    torch.cuda.synchronize()
    # stored gradient: the apply kernel's, up to one bf16 ulp where the fp32 evaluation order moves a value across a rounding boundary
    assert_close(dx1.float().cpu(), dx0.float().cpu(), "stored dx vs crd_gn_bwd_apply", rel=2e-3, elem=1.6e-2)
    assert_close(dx1.float().cpu().reshape(B, H, W, Ci).permute(0, 3, 1, 2), dx_ref, "stored dx vs autograd", rel=6e-3, elem=2e-2)
    assert torch.equal(dga0, dga1) and torch.equal(dbe0, dbe1)
    assert_close(y1.float().cpu(), y0.float().cpu(), "product vs apply + conv_igemm", rel=4e-3, elem=1.5e-2)
    # against torch: conv of the bf16-rounded autograd gradient
    yt = torch.einsum("bpk,nk->bpn", bf(dx_ref).permute(0, 2, 3, 1).reshape(B, P, Ci), w)
    if scatter:
        yt = yt.reshape(B, H, W, pk, pk, Co).permute(0, 1, 3, 2, 4, 5).reshape(B, YH * YW, Co)
    assert_close(y1.float().cpu() - base, yt, "product vs torch", rel=8e-3, elem=2.5e-2)
    if mode == "stats":
        assert_close(sval(ost1), sval(ost0), "output sums", rel=2e-3, elem=5e-3)
    if red1 is not None:
        assert_close(gval(red1), gval(red0), "fused reduce of the next GroupNorm", rel=3e-3, elem=3e-3)


@pytest.mark.parametrize("B,C_,H,W,sr", [(8, 160, 16, 26, 2), (2, 128, 32, 52, 4), (3, 160, 6, 10, 2), (2, 64, 16, 24, 8)])
def test_gn_conv2_equals_two_gn_conv_launches(B, C_, H, W, sr):
    """crd_gn_conv2 -- attn.q and the attn.sr patch convolution of a Block, both behind Block.norm1 of the fp32 residual stream, in ONE
    launch -- against the two crd_gn_conv launches: the same kernel body on the same tiles, so outputs, the stored normalised operand and
    the output GroupNorm sums are bit-identical."""
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(C_ + sr)
    x = (torch.randn(B, H * W, C_, generator=g) * 1.3 + 0.2).cuda()
    stats = zsum(B, C_ // 16, 2)
    lib.check(L.crd_gn_stats(x.data_ptr(), 1, C_, 0, B, H * W, C_, stats.data_ptr(), None, lib.stream()), "gn_stats")
    gam, bet = (1 + 0.2 * torch.randn(C_, generator=g)).cuda(), (0.1 * torch.randn(C_, generator=g)).cuda()
    wq, wsr = pack_w(torch.randn(C_, C_, 1, 1, generator=g) / C_ ** 0.5), pack_w(torch.randn(C_, C_, sr, sr, generator=g) / (C_ * sr * sr) ** 0.5)
    bq, bsr = (0.1 * torch.randn(C_, generator=g)).cuda(), (0.1 * torch.randn(C_, generator=g)).cuda()
    OH, OW = H // sr, W // sr

    def descs(q_out, xn_out, kr_out, kr_stats):
        dq, nq, ds, ns = lib.ConvDesc(), lib.GnInput(), lib.ConvDesc(), lib.GnInput()
        for d, w, co, k, oh, ow, y, bias in ((dq, wq, C_, 1, H, W, q_out, bq), (ds, wsr, C_, sr, OH, OW, kr_out, bsr)):
            d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = x.data_ptr(), C_, 0, B, H, W, C_
            d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w.data_ptr(), co, k, k, k, 0, oh, ow
            d.y, d.y_ld, d.y_f32, d.bias = y.data_ptr(), C_, 0, bias.data_ptr()
        ds.stats = kr_stats.data_ptr()
        for n in (nq, ns):
            n.x_f32, n.gmul, n.act = 1, 1, 0
            n.stats, n.gamma, n.beta = stats.data_ptr(), gam.data_ptr(), bet.data_ptr()
        nq.xn, nq.xn_ld = xn_out.data_ptr(), C_
        return dq, nq, ds, ns
    outs = []
    for fused in (False, True):
        q_out = torch.zeros(B, H * W, C_, dtype=torch.bfloat16, device="cuda")
        xn_out = torch.zeros_like(q_out)
        kr_out = torch.zeros(B, OH * OW, C_, dtype=torch.bfloat16, device="cuda")
        kr_stats = zsum(B, C_ // 16, 2)
        dq, nq, ds, ns = descs(q_out, xn_out, kr_out, kr_stats)
        if fused:
            lib.check(L.crd_gn_conv2(C.byref(dq), C.byref(nq), C.byref(ds), C.byref(ns), lib.stream()), "crd_gn_conv2")
        else:
            lib.check(L.crd_gn_conv(C.byref(dq), C.byref(nq), lib.stream()), "crd_gn_conv q")
            lib.check(L.crd_gn_conv(C.byref(ds), C.byref(ns), lib.stream()), "crd_gn_conv sr")
        torch.cuda.synchronize()
        outs.append((q_out, xn_out, kr_out, kr_stats))
    for a, b_, what in zip(outs[0], outs[1], ("q", "stored norm1(x)", "sr output", "sr output sums")):
        assert torch.equal(a, b_), what
    assert float(outs[1][0].float().abs().max()) > 0 and float(outs[1][2].float().abs().max()) > 0
