"""GPU parity of the implicit-GEMM convolution kernel (crd_conv_igemm) against torch fp32 on the
same bf16-rounded operands.  Tolerance: fp32 accumulation-order noise + one bf16 output rounding
(rel 2^-8) -> |err| <= 1e-2 * max|ref| elementwise, rel-L2 < 4e-3."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import gval, sval, zsum  # noqa: E402

pytestmark = pytest.mark.gpu


def _lib():
    from camradepth_amd import lib
    return lib


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def to_pm(x, ld=None, coff=0):
    """NCHW fp32 cpu -> pixel-major bf16 cuda buffer [B,H,W,ld] with x at channel offset coff."""
    B, Cc, H, W = x.shape
    ld = ld or Cc
    buf = torch.zeros(B, H, W, ld, dtype=torch.bfloat16)
    buf[..., coff:coff + Cc] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return buf.cuda()


def pack_w(w, cin_pad=None):
    """[Cout,Cin,KH,KW] fp32 -> bf16 [Cout][KH*KW][Cin_pad] cuda."""
    Co, Ci, KH, KW = w.shape
    cin_pad = cin_pad or Ci
    p = torch.zeros(Co, KH * KW, cin_pad, dtype=torch.bfloat16)
    p[:, :, :Ci] = w.permute(0, 2, 3, 1).reshape(Co, KH * KW, Ci).to(torch.bfloat16)
    return p.cuda()


@pytest.fixture(autouse=True, params=[0, 512], ids=["wide-tiles", "small-grid-tiles"])
def _conv3x3_tile_choice(request):
    """The 3x3 halo kernel picks 64 / 32-column tiles for launches of fewer than 512 workgroups -- which is every case of
    this file.  Run everything under both settings so that the 96 / 128-column configurations stay covered."""
    L = _lib().load()
    old = L.crd_tune_conv3x3_small_grid(request.param)
    yield
    L.crd_tune_conv3x3_small_grid(old)


def run_conv(xpm, x_ld, x_coff, B, IH, IW, Cin, wp, Cout, KH, KW, stride, pad, OH, OW, y, y_ld, y_coff, y_f32=0,
             gather_mode=0, out_mode=0, patch_k=0, patch_c=0, bias=None, act=0, res=None, res_ld=0, res_scale=None,
             accumulate=0, stats=None, partial=None, red=None, chan=None):
    lib = _lib()
    L = lib.load()
    d = lib.ConvDesc()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = xpm.data_ptr(), x_ld, x_coff, B, IH, IW, Cin
    d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = wp.data_ptr(), Cout, KH, KW, stride, pad, OH, OW
    d.gather_mode = gather_mode
    d.y, d.y_ld, d.y_coff, d.y_f32 = y.data_ptr(), y_ld, y_coff, y_f32
    d.out_mode, d.patch_k, d.patch_c = out_mode, patch_k, patch_c
    d.bias = bias.data_ptr() if bias is not None else None
    d.act = act
    d.res = res.data_ptr() if res is not None else None
    d.res_ld = res_ld
    d.res_scale = res_scale.data_ptr() if res_scale is not None else None
    d.accumulate = accumulate
    d.stats = stats.data_ptr() if stats is not None else None
    d.chan_sums = chan.data_ptr() if chan is not None else None
    if partial is not None:
        d.stats_partial, d.stats_partial_capacity = partial.data_ptr(), partial.numel()
    if red is not None:
        rx, rstats, rgamma, rbeta, rgmul, ract, rr = red
        d.red_x, d.red_x_ld, d.red_gmul, d.red_act = rx.data_ptr(), rx.shape[-1], rgmul, ract
        d.red_x_f32 = 1 if rx.dtype == torch.float32 else 0
        d.red_stats, d.red_gamma, d.red_beta, d.red_r = rstats.data_ptr(), rgamma.data_ptr(), rbeta.data_ptr(), rr.data_ptr()
    lib.check(L.crd_conv_igemm(C.byref(d), lib.stream()), "crd_conv_igemm")
    torch.cuda.synchronize()


def assert_close(got, ref, what, rel=4e-3, elem=1e-2):
    got, ref = got.double(), ref.double()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    rl2 = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    assert err <= elem * scale and rl2 < rel, f"{what}: max err {err:.3e} (scale {scale:.3e}), rel-L2 {rl2:.3e}"


CASES = [
    # B, Cin(ref), Cin_pad, H, W, Cout, k, stride, pad
    (2, 136, 136, 19, 23, 96, 3, 1, 1),
    (1, 232, 232, 16, 20, 64, 3, 1, 1),
    (2, 296, 296, 9, 14, 128, 3, 1, 1),
    (1, 129, 136, 12, 13, 32, 3, 1, 1),
    (2, 128, 128, 10, 11, 21, 3, 1, 1),
    (1, 32, 32, 13, 9, 1, 3, 1, 1),
    (2, 7, 8, 32, 48, 64, 7, 4, 3),
    (2, 64, 64, 16, 24, 128, 3, 2, 1),
    (1, 128, 128, 9, 13, 160, 3, 2, 1),
    (2, 64, 64, 16, 24, 64, 8, 8, 0),
    (2, 160, 160, 8, 6, 160, 2, 2, 0),
    (3, 256, 256, 4, 7, 256, 1, 1, 0),
    (2, 64, 64, 20, 26, 512, 1, 1, 0),
    (1, 640, 640, 6, 7, 160, 1, 1, 0),
    # wide pointwise layers (k_gn_pw_wide, plain rows): ragged tiles, 256- and 128-column workgroups, atomic sums (odd pixel counts)
    (3, 128, 128, 9, 11, 1024, 1, 1, 0),
    (2, 64, 64, 21, 31, 512, 1, 1, 0),
    (2, 160, 160, 15, 27, 640, 1, 1, 0),
    (5, 64, 64, 13, 17, 384, 1, 1, 0),
    # 3x3 / stride 1 on grids >= 32 wide: halo-tile kernel (conv3x3.hip), all Cout tile variants, ragged borders
    (2, 136, 136, 19, 45, 96, 3, 1, 1),
    (1, 232, 232, 8, 64, 64, 3, 1, 1),
    (2, 296, 304, 33, 70, 128, 3, 1, 1),
    (1, 129, 136, 17, 40, 32, 3, 1, 1),
    (2, 128, 128, 9, 33, 21, 3, 1, 1),
    (1, 48, 48, 40, 32, 128, 3, 1, 1),
    # 160-column tile (3-slot weight ring): one and two tiles, ragged
    (2, 96, 96, 17, 40, 144, 3, 1, 1),
    (1, 128, 128, 9, 64, 304, 3, 1, 1),
    (1, 64, 64, 24, 33, 136, 3, 1, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_forward(case):
    B, Ci, Cp, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5)
    bias = torch.randn(Co, generator=g) * 0.1
    ref = F.conv2d(x, w, bias, stride=s, padding=p)
    OH, OW = ref.shape[2], ref.shape[3]
    xpm = to_pm(x, ld=Cp + 16, coff=8)
    wp = pack_w(w, Cp)
    ld_y = ((Co + 7) // 8) * 8 + 8
    y = torch.zeros(B, OH, OW, ld_y, dtype=torch.bfloat16, device="cuda")
    stats = zsum(B, Co // 16, 2) if Co % 16 == 0 else None
    # odd-sized cases use the atomic statistics path, the others the per-tile partials + finalize path
    partial = torch.full((B * (-(-OH * OW // 64)) * max(Co // 16, 1) * 2,), 7.0, device="cuda") if (H * W) % 2 == 0 else None
    run_conv(xpm, Cp + 16, 8, B, H, W, Cp, wp, Co, k, k, s, p, OH, OW, y, ld_y, 8 if Co % 8 == 0 else 0,
             bias=bias.cuda(), stats=stats, partial=partial)
    coff = 8 if Co % 8 == 0 else 0
    got = y[..., coff:coff + Co].float().cpu().permute(0, 3, 1, 2)
    assert_close(got, ref, f"conv {case}")
    # untouched channels of the wider buffer stay zero
    if coff:
        assert float(y[..., :coff].float().abs().max()) == 0.0
    if stats is not None:
        gq = got.reshape(B, Co // 16, 16, OH * OW)
        ref_s = torch.stack([gq.sum((2, 3)), (gq ** 2).sum((2, 3))], -1)
        assert_close(sval(stats), ref_s, f"stats {case}", rel=1e-3, elem=2e-3)


def test_epilogues_sigmoid_residual_accumulate():
    g = torch.Generator().manual_seed(3)
    B, Ci, H, W, Co = 2, 64, 7, 9, 64
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 1, 1, generator=g) / 8)
    bias = torch.randn(Co, generator=g) * 0.1
    conv = F.conv2d(x, w, bias)
    xpm, wp = to_pm(x), pack_w(w)
    # sigmoid
    y = torch.zeros(B, H, W, Co, dtype=torch.bfloat16, device="cuda")
    run_conv(xpm, Ci, 0, B, H, W, Ci, wp, Co, 1, 1, 1, 0, H, W, y, Co, 0, bias=bias.cuda(), act=1)
    assert_close(y.float().cpu().permute(0, 3, 1, 2), torch.sigmoid(conv), "sigmoid")
    # residual: y = res + scale[b] * bf16(conv)
    res = torch.randn(B, H, W, Co, generator=g)
    scale = torch.tensor([0.0, 1.25])
    yf = torch.zeros(B, H, W, Co, device="cuda")
    run_conv(xpm, Ci, 0, B, H, W, Ci, wp, Co, 1, 1, 1, 0, H, W, yf, Co, 0, y_f32=1, bias=bias.cuda(), res=res.cuda(),
             res_ld=Co, res_scale=scale.cuda())
    ref = res + scale.view(B, 1, 1, 1) * bf(conv).permute(0, 2, 3, 1)
    assert_close(yf.cpu(), ref, "residual", rel=2e-3, elem=6e-3)
    # the same with the GroupNorm sums of the stored fp32 output: g16 (sum, sumsq) and per-channel (sum, sumsq)
    if Co % 16 == 0:
        yf2 = torch.zeros(B, H, W, Co, device="cuda")
        st, ch = zsum(B, Co // 16, 2), zsum(B, Co, 2)
        run_conv(xpm, Ci, 0, B, H, W, Ci, wp, Co, 1, 1, 1, 0, H, W, yf2, Co, 0, y_f32=1, bias=bias.cuda(), res=res.cuda(),
                 res_ld=Co, res_scale=scale.cuda(), stats=st, chan=ch)
        assert torch.equal(yf2, yf)
        yd = yf2.double().cpu().view(B, H * W, Co)
        chr_ = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1)
        assert_close(sval(ch), chr_, "channel sums", rel=1e-5, elem=1e-5)
        assert_close(sval(st), chr_.view(B, Co // 16, 16, 2).sum(2), "g16 sums", rel=1e-5, elem=1e-5)
    # accumulate into bf16
    y0 = bf(torch.randn(B, H, W, Co, generator=g))
    y = y0.to(torch.bfloat16).cuda()
    run_conv(xpm, Ci, 0, B, H, W, Ci, wp, Co, 1, 1, 1, 0, H, W, y, Co, 0, accumulate=1)
    assert_close(y.float().cpu(), y0 + F.conv2d(x, w).permute(0, 2, 3, 1), "accumulate")


@pytest.mark.parametrize("Ci,Co,H,W,acc", [(640, 160, 16, 26, 0), (1024, 256, 8, 13, 0), (160, 160, 9, 7, 1), (128, 96, 12, 20, 0),
                                           # round 5: Mlp.fc2 of stages 1-2 on rows that are already activated -- the narrow streaming
                                           # kernel (csrc/pw_narrow.hip), 12 / 52 / 3 tiles of 32 rows per sample
                                           (512, 64, 16, 24, 0), (1024, 128, 32, 52, 0), (1024, 128, 8, 12, 0)])
def test_fp32_residual_epilogue_with_sums(Ci, Co, H, W, acc):
    """LDS-staged float4 epilogue: fp32 output = res + scale[b] * bf16(conv) (+ previous contents), GroupNorm sums of the
    stored values per 16-channel slab and per channel -- on the split-K (deep K, small grid) and plain small-tile kernels,
    with a partial last column tile."""
    B = 2
    g = torch.Generator().manual_seed(31)
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5)
    bias = torch.randn(Co, generator=g) * 0.1
    conv = F.conv2d(x, w, bias)
    xpm, wp = to_pm(x), pack_w(w)
    res = torch.randn(B, H, W, Co, generator=g)
    scale = torch.tensor([0.5, 1.25])
    y0 = torch.randn(B, H, W, Co, generator=g) if acc else torch.zeros(B, H, W, Co)
    yf = y0.clone().cuda()
    st, ch = zsum(B, Co // 16, 2), zsum(B, Co, 2)
    run_conv(xpm, Ci, 0, B, H, W, Ci, wp, Co, 1, 1, 1, 0, H, W, yf, Co, 0, y_f32=1, bias=bias.cuda(), res=res.cuda(),
             res_ld=Co, res_scale=scale.cuda(), accumulate=acc, stats=st, chan=ch)
    ref = res + scale.view(B, 1, 1, 1) * bf(conv).permute(0, 2, 3, 1) + y0
    assert_close(yf.cpu(), ref, "fp32 residual output", rel=2e-3, elem=6e-3)
    yd = yf.double().cpu().view(B, H * W, Co)
    chr_ = torch.stack([yd.sum(1), (yd * yd).sum(1)], -1)
    assert_close(sval(ch), chr_, "channel sums", rel=1e-5, elem=1e-5)
    assert_close(sval(st), chr_.view(B, Co // 16, 16, 2).sum(2), "g16 sums", rel=1e-5, elem=1e-5)


DGRAD_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad   (forward conv geometry; we test its data gradient)
    (2, 136, 11, 13, 96, 3, 1, 1),
    (1, 64, 16, 24, 128, 3, 2, 1),
    (2, 128, 9, 13, 160, 3, 2, 1),
    (2, 64, 10, 10, 64, 1, 1, 0),
    (2, 304, 19, 45, 128, 3, 1, 1),      # halo-tile kernel, data-gradient mode (mirrored taps), 3 N tiles
    (1, 144, 17, 40, 96, 3, 1, 1),
    (1, 128, 9, 64, 32, 3, 1, 1),
    (1, 296, 9, 33, 128, 3, 1, 1),       # 160-column tiles: N = 296 (two tiles), 136 (one)
    (2, 136, 17, 40, 96, 3, 1, 1),
    (2, 512, 20, 26, 64, 1, 1, 0),       # Mlp.fc2's data gradient (64 -> 512 columns): the wide pointwise kernel
    (2, 1024, 9, 11, 128, 1, 1, 0),
    (1, 640, 16, 26, 160, 1, 1, 0),
    (2, 64, 16, 24, 512, 1, 1, 0),       # Mlp.fc1's data gradient (512 -> 64 columns): the narrow streaming kernel, plain bf16 output
    (3, 128, 8, 12, 1024, 1, 1, 0),
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv_dgrad_gather_mode(case):
    B, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(11)
    w = bf(torch.randn(Co, Ci, k, k, generator=g) / (Co * k * k) ** 0.5)
    x = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
    yref = F.conv2d(x, w, None, stride=s, padding=p)
    OH, OW = yref.shape[2], yref.shape[3]
    dy = bf(torch.randn(B, Co, OH, OW, generator=g))
    yref.backward(dy)
    # dgrad weights: [Cin][tap][Cout]
    wd = w.permute(1, 2, 3, 0).contiguous().reshape(Ci, k * k, Co).to(torch.bfloat16).cuda()
    dypm = to_pm(dy)
    dx = torch.zeros(B, H, W, Ci, dtype=torch.bfloat16, device="cuda")
    run_conv(dypm, Co, 0, B, OH, OW, Co, wd, Ci, k, k, s, p, H, W, dx, Ci, 0, gather_mode=1)
    assert_close(dx.float().cpu().permute(0, 3, 1, 2), x.grad, f"dgrad {case}")


@pytest.mark.parametrize("Ci,Co,H,W,gmul,act,xf32", [(640, 160, 16, 26, 4, 1, 0), (512, 64, 24, 40, 8, 1, 0), (256, 64, 9, 7, 1, 0, 0),
                                                      (1024, 256, 2, 3, 4, 1, 0), (1024, 256, 8, 13, 4, 1, 0), (1024, 128, 19, 27, 8, 1, 0), (512, 64, 64, 104, 8, 1, 0),
                                                      # Mlp.fc1's data gradient feeding Block.norm2 (fp32 residual stream, no activation)
                                                      (160, 640, 16, 26, 1, 0, 1), (128, 1024, 32, 52, 1, 0, 1), (64, 512, 64, 104, 1, 0, 1)])
def test_conv_dgrad_with_fused_groupnorm_backward_reduce(Ci, Co, H, W, gmul, act, xf32):
    """Data gradient of a 1x1 conv (Mlp.fc2) that also runs the reduce phase of the GroupNorm(+GELU) backward on its own
    output: same dx, and r equal to crd_gn_bwd_reduce on (x_gn, dx).  Both tile configurations (64- and 128-wide)."""
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(13)
    B = 2
    w = bf(torch.randn(Co, Ci, 1, 1, generator=g) / Co ** 0.5)
    dy = bf(torch.randn(B, Co, H, W, generator=g))
    wd = w.permute(1, 2, 3, 0).contiguous().reshape(Ci, 1, Co).to(torch.bfloat16).cuda()
    dypm = to_pm(dy)
    xg = to_pm(bf(torch.randn(B, Ci, H, W, generator=g) * 1.3 + 0.2))       # the GroupNorm's raw input, [B, H*W, Ci]
    if xf32:
        xg = (xg.float() + 0.001 * torch.randn(xg.shape, generator=g).cuda()).contiguous()
    gam, bet = (1 + 0.1 * torch.randn(Ci, generator=g)).cuda(), (0.1 * torch.randn(Ci, generator=g)).cuda()
    stats = zsum(B, Ci // 16, 2)
    lib.check(L.crd_gn_stats(xg.data_ptr(), xf32, Ci, 0, B, H * W, Ci, stats.data_ptr(), None, lib.stream()), "gn_stats")
    G = Ci // (16 * gmul)
    dx0 = torch.zeros(B, H, W, Ci, dtype=torch.bfloat16, device="cuda")
    run_conv(dypm, Co, 0, B, H, W, Co, wd, Ci, 1, 1, 1, 0, H, W, dx0, Ci, 0, gather_mode=1)
    r_ref = zsum(B * Ci * 2 + B * G * 2)
    lib.check(L.crd_gn_bwd_reduce(xg.data_ptr(), xf32, Ci, 0, dx0.data_ptr(), 0, Ci, 0, B, H * W, Ci, stats.data_ptr(), gmul,
                                  gam.data_ptr(), bet.data_ptr(), act, None, r_ref.data_ptr(), None, 0, lib.stream()), "gn_bwd_reduce")
    dx1 = torch.zeros_like(dx0)
    r = torch.zeros_like(r_ref)
    run_conv(dypm, Co, 0, B, H, W, Co, wd, Ci, 1, 1, 1, 0, H, W, dx1, Ci, 0, gather_mode=1, red=(xg, stats, gam, bet, gmul, act, r))
    assert torch.equal(dx0, dx1)
    assert_close(gval(r), gval(r_ref), "fused gn-bwd reduce", rel=3e-4, elem=3e-4)


@pytest.mark.parametrize("k,C", [(8, 64), (4, 128), (2, 160)])
def test_conv_dgrad_patch_scatter(k, C):
    g = torch.Generator().manual_seed(5)
    B, PH, PW = 2, 3, 5
    H, W = PH * k, PW * k
    w = bf(torch.randn(C, C, k, k, generator=g) / (C * k * k) ** 0.5)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    yref = F.conv2d(x, w, None, stride=k)
    dy = bf(torch.randn(B, C, PH, PW, generator=g))
    yref.backward(dy)
    # scatter weights: [(tap, ci)][co]
    ws = w.permute(2, 3, 1, 0).contiguous().reshape(k * k * C, 1, C).to(torch.bfloat16).cuda()
    dypm = to_pm(dy)
    dx = torch.zeros(B, H, W, C, dtype=torch.bfloat16, device="cuda")
    run_conv(dypm, C, 0, B, PH, PW, C, ws, k * k * C, 1, 1, 1, 0, PH, PW, dx, C, 0, out_mode=1, patch_k=k, patch_c=C)
    assert_close(dx.float().cpu().permute(0, 3, 1, 2), x.grad, f"scatter k={k}")
    # accumulate into an existing bf16 gradient (how the sr path adds into d(XN)) -- the 16-byte read-modify-write epilogue
    base = bf(torch.randn(B, H, W, C, generator=g))
    dx2 = base.to(torch.bfloat16).cuda()
    run_conv(dypm, C, 0, B, PH, PW, C, ws, k * k * C, 1, 1, 1, 0, PH, PW, dx2, C, 0, out_mode=1, patch_k=k, patch_c=C, accumulate=1)
    assert_close(dx2.float().cpu(), base + dx.float().cpu(), f"scatter accumulate k={k}", rel=4e-3, elem=1e-2)


def test_bad_arguments_are_reported():
    lib = _lib()
    L = lib.load()
    d = lib.ConvDesc()
    rc = L.crd_conv_igemm(C.byref(d), lib.stream())
    assert rc == -1 and b"null" in L.crd_last_error()
    with pytest.raises(lib.CrdError):
        lib.check(rc, "crd_conv_igemm")


# ---- persistent one-wave-per-SIMD 3x3 kernel (conv3x3p.hip): taken for plain bf16 outputs on grids of >= 192 16x32 tiles
PERSIST_FWD = [
    # B, Cin(ref), Cin_pad, H, W, Cout
    (8, 136, 136, 90, 120, 96),      # 96-column tile, ragged right / bottom borders, 8-channel K tail
    (8, 296, 304, 96, 128, 128),     # 128-column tile, padded concat input (zero weight columns)
    (8, 232, 232, 94, 128, 64),      # 64-column tile
    (12, 48, 48, 64, 128, 128),      # two chunks, one of them half full
]


@pytest.mark.parametrize("case", PERSIST_FWD)
def test_conv3x3_persistent_forward(case):
    B, Ci, Cp, H, W, Co = case
    g = torch.Generator().manual_seed(sum(case))
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5)
    ref = F.conv2d(x, w, None, padding=1)
    xpm = to_pm(x, ld=Cp + 16, coff=8)
    wp = pack_w(w, Cp)
    ld_y = Co + 8
    y = torch.zeros(B, H, W, ld_y, dtype=torch.bfloat16, device="cuda")
    stats = zsum(B, Co // 16, 2)
    # the persistent kernel writes its GroupNorm sums as per-(16 x 32 tile, wave) partial rows (poisoned here: every row it
    # reads must have been written) that a finalize kernel folds into `stats`
    partial = torch.full((B * -(-W // 32) * -(-H // 16) * 8 * (Co // 16) * 2,), float("nan"), device="cuda")
    run_conv(xpm, Cp + 16, 8, B, H, W, Cp, wp, Co, 3, 3, 1, 1, H, W, y, ld_y, 8, stats=stats, partial=partial)
    got = y[..., 8:8 + Co].float().cpu().permute(0, 3, 1, 2)
    assert_close(got, ref, f"persistent conv {case}")
    assert float(y[..., :8].float().abs().max()) == 0.0
    gq = got.reshape(B, Co // 16, 16, H * W)
    ref_s = torch.stack([gq.sum((2, 3)), (gq ** 2).sum((2, 3))], -1)
    assert_close(sval(stats), ref_s, f"persistent stats {case}", rel=1e-3, elem=2e-3)
    # the same launch through the two-workgroup kernel (CRD_CONV3P=0 is read once per process: compare with the torch
    # reference only) -- and a second call must give the same result (the persistent loop leaves no state behind)
    y2 = torch.zeros_like(y)
    run_conv(xpm, Cp + 16, 8, B, H, W, Cp, wp, Co, 3, 3, 1, 1, H, W, y2, ld_y, 8)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("B,Ci,H,W,Co,acc", [(8, 304, 96, 128, 128, 0), (8, 240, 90, 120, 64, 1), (8, 144, 96, 128, 96, 0), (8, 136, 96, 128, 96, 1)])
def test_conv3x3_persistent_dgrad(B, Ci, H, W, Co, acc):
    """Data gradients towards 304 / 240 / 144 / 136 channels: 128-column tiles on the persistent kernel (a masked last tile
    for 240), the <= 64-column tail on the narrow two-workgroup tiles; store and accumulate."""
    g = torch.Generator().manual_seed(Ci + Co)
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / (Co * 9) ** 0.5)
    dy = bf(torch.randn(B, Co, H, W, generator=g))
    ref = F.conv_transpose2d(dy, w, None, padding=1)             # = d/dx of conv2d(x, w, padding=1)
    wd = w.permute(1, 2, 3, 0).contiguous().reshape(Ci, 9, Co).to(torch.bfloat16).cuda()
    dypm = to_pm(dy)
    base = bf(torch.randn(B, H, W, Ci, generator=g)) if acc else torch.zeros(B, H, W, Ci)
    dx = base.to(torch.bfloat16).cuda()
    run_conv(dypm, Co, 0, B, H, W, Co, wd, Ci, 3, 3, 1, 1, H, W, dx, Ci, 0, gather_mode=1, accumulate=acc)
    assert_close(dx.float().cpu(), base + ref.permute(0, 2, 3, 1), f"persistent dgrad {Ci}<-{Co}", rel=5e-3, elem=1.5e-2)


REGE_CASES = [
    # what, Cin, Cout, H, W  (B = 2; 64 x 64 tiles: the small-grid path of crd_conv_igemm)
    ("fwd_bias_stats", 160, 160, 16, 26), ("fwd_bias_stats", 128, 128, 9, 13), ("fwd_sigmoid", 64, 72, 10, 11),
    ("dgrad_acc", 160, 160, 16, 26), ("dgrad_acc_bias", 128, 128, 32, 52),
    # (K <= 448 here: from eight K-slabs on, a grid this small takes the split-K variant, which keeps the staged epilogue)
    ("dgrad_red_act", 640, 160, 16, 26), ("dgrad_red_f32", 160, 320, 16, 26), ("dgrad_red_f32_acc_bias", 160, 160, 16, 26),
    ("dgrad_red_f32", 128, 128, 8, 13),
]


@pytest.mark.parametrize("case", REGE_CASES, ids=[f"{c[0]}_{c[1]}to{c[2]}_{c[3]}x{c[4]}" for c in REGE_CASES])
def test_register_epilogue_matches_the_staged_epilogue(case):
    """Round 5: the 64 x 64 tiles' register epilogue (conv_common.h: conv_epilogue_reg) against the LDS-staged one it replaces, toggled with
    crd_tune_igemm_reg_epilogue: identical stored bf16 values (plain, sigmoid, accumulate, per-image bias), GroupNorm sums and the fused
    GroupNorm-backward reduce equal up to the summation order of fp32 partials -- and the register path really ran."""
    what, Ci, Co, H, W = case
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(sum(map(ord, what)) + Ci + Co)
    B = 2
    dgrad = what.startswith("dgrad")
    w = bf(torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5)
    x = bf(torch.randn(B, Co if dgrad else Ci, H, W, generator=g))
    xpm = to_pm(x)
    wp = w.permute(1, 2, 3, 0).contiguous().reshape(Ci, 1, Co).to(torch.bfloat16).cuda() if dgrad else pack_w(w)
    N = Ci if dgrad else Co                   # output channels of the launch
    K = Co if dgrad else Ci
    bias = (torch.randn(B, N, generator=g) * 0.1).cuda() if "bias" in what else None
    y0 = bf(torch.randn(B, H, W, N, generator=g)).to(torch.bfloat16).cuda()
    red_in = None
    if "red" in what:
        xf32 = "f32" in what
        gmul, act = (1, 0) if xf32 else (4, 1)
        xg = to_pm(bf(torch.randn(B, N, H, W, generator=g) * 1.3 + 0.2))
        if xf32:
            xg = (xg.float() + 0.001 * torch.randn(xg.shape, generator=g).cuda()).contiguous()
        gam, bet = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.1 * torch.randn(N, generator=g)).cuda()
        gst = zsum(B, N // 16, 2)
        lib.check(L.crd_gn_stats(xg.data_ptr(), 1 if xf32 else 0, N, 0, B, H * W, N, gst.data_ptr(), None, lib.stream()), "gn_stats")
        red_in = (xg, gst, gam, bet, gmul, act)
    outs = {}
    for on in (0, 1):
        prev = L.crd_tune_igemm_reg_epilogue(on)
        n0 = L.crd_tune_igemm_reg_epilogue(-1)
        y = y0.clone()
        st = zsum(B, N // 16, 2) if "stats" in what else None
        r = zsum(B * N * 2 + B * (N // (16 * red_in[4])) * 2) if red_in else None
        d = lib.ConvDesc()
        d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = xpm.data_ptr(), K, 0, B, H, W, K
        d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = wp.data_ptr(), N, 1, 1, 1, 0, H, W
        d.gather_mode = 1 if dgrad else 0
        d.y, d.y_ld, d.y_coff = y.data_ptr(), N, 0
        d.accumulate = 1 if "acc" in what else 0
        d.act = 1 if "sigmoid" in what else 0
        if bias is not None:
            d.bias, d.bias_bstride = bias.data_ptr(), N
        if st is not None:
            d.stats = st.data_ptr()
        if red_in:
            xg, gst, gam, bet, gmul, act = red_in
            d.red_x, d.red_x_ld, d.red_gmul, d.red_act, d.red_x_f32 = xg.data_ptr(), N, gmul, act, 1 if xg.dtype == torch.float32 else 0
            d.red_stats, d.red_gamma, d.red_beta, d.red_r = gst.data_ptr(), gam.data_ptr(), bet.data_ptr(), r.data_ptr()
        lib.check(L.crd_conv_igemm(C.byref(d), lib.stream()), "crd_conv_igemm")
        torch.cuda.synchronize()
        took = L.crd_tune_igemm_reg_epilogue(-1) - n0
        L.crd_tune_igemm_reg_epilogue(prev)
        assert took == on, f"register epilogue launches: {took} with the switch at {on}"
        outs[on] = (y, st, r)
    (ya, sa, ra), (yb, sb, rb) = outs[0], outs[1]
    assert torch.equal(ya, yb), f"stored values differ: max {float((ya.float() - yb.float()).abs().max()):.3e}"
    if sa is not None:
        assert_close(sval(sb), sval(sa), "GroupNorm sums", rel=1e-5, elem=1e-5)
    if ra is not None:
        assert_close(gval(rb), gval(ra), "fused gn-bwd reduce", rel=3e-4, elem=3e-4)
