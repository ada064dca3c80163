"""GPU parity of the non-GEMM kernels and the weight-gradient kernel against torch fp32 references
computed on the same bf16-rounded operands (through the C ABI, ctypes)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_igemm import assert_close, bf, to_pm
from tests.util import gval, sval, to_grad, to_stat, zsum

pytestmark = pytest.mark.gpu


def L():
    from camradepth_amd import lib
    return lib, lib.load()


def P(t):
    return None if t is None else t.data_ptr()


def ok(rc, what):
    lib, _ = L()
    lib.check(rc, what)
    torch.cuda.synchronize()


def _wgrad_problem(case, seed):
    lib, _ = L()
    B, Ci, Cp, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(seed)
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    bias = torch.zeros(Co, requires_grad=True)
    y = F.conv2d(x, w, bias, stride=s, padding=p)
    OH, OW = y.shape[2], y.shape[3]
    dy = bf(torch.randn(B, Co, OH, OW, generator=g))
    y.backward(dy)
    Cop = ((Co + 7) // 8) * 8
    xpm, dypm = to_pm(x, ld=Cp), to_pm(dy, ld=Cop)
    dw = zsum(Co, k * k, Cp)
    db = zsum(Co)
    return dict(x=xpm, dy=dypm, dw=dw, db=db, wg=w.grad, bg=bias.grad, dims=(B, Ci, Cp, H, W, Co, Cop, k, s, p, OH, OW))


def _fill_wgrad_desc(d, pr):
    B, Ci, Cp, H, W, Co, Cop, k, s, p, OH, OW = pr["dims"]
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = P(pr["x"]), Cp, 0, B, H, W, Cp
    d.dy, d.dy_ld, d.dy_coff, d.OH, d.OW, d.Cout = P(pr["dy"]), Cop, 0, OH, OW, Co
    d.KH, d.KW, d.stride, d.pad, d.dw, d.dbias = k, k, s, p, P(pr["dw"]), P(pr["db"])


def test_conv_wgrad_grouped():
    """One grouped dispatch over problems of every tile configuration (Cout 21/64/96/160/512), strided and 1x1, equals
    the per-problem references; the size query and the table build agree."""
    lib, lb = L()
    cases = [(3, 256, 256, 4, 7, 256, 1, 1, 0), (2, 64, 64, 40, 52, 512, 1, 1, 0), (1, 640, 640, 6, 7, 160, 1, 1, 0),
             (2, 64, 64, 16, 24, 64, 8, 8, 0), (2, 128, 128, 10, 11, 21, 3, 1, 1), (2, 136, 136, 19, 23, 96, 3, 1, 1),
             (2, 160, 160, 16, 26, 160, 2, 2, 0), (8, 160, 160, 1, 1, 160, 1, 1, 0)]
    probs = [_wgrad_problem(c, 11 + i) for i, c in enumerate(cases)]
    descs = (lib.WgradDesc * len(probs))()
    for d, pr in zip(descs, probs):
        _fill_wgrad_desc(d, pr)
    info = lib.WgradGroupInfo()
    lib.check(lb.crd_wgrad_group_build(descs, len(probs), None, 0, C.byref(info)), "size query")
    assert info.bytes > 0 and info.n_problems == len(probs) and sum(info.n_items) > len(probs)
    host = (C.c_uint8 * info.bytes)()
    lib.check(lb.crd_wgrad_group_build(descs, len(probs), host, info.bytes, C.byref(info)), "build")
    table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    ok(lb.crd_conv_wgrad_grouped(P(table), C.byref(info), lib.stream()), "crd_conv_wgrad_grouped")
    for c, pr in zip(cases, probs):
        B, Ci, Cp, H, W, Co, Cop, k, s, p, OH, OW = pr["dims"]
        got = gval(pr["dw"])[:, :, :Ci].reshape(Co, k, k, Ci).permute(0, 3, 1, 2)
        assert_close(got, pr["wg"], f"grouped wgrad {c}", rel=2e-3, elem=4e-3)
        assert_close(gval(pr["db"]), pr["bg"], f"grouped dbias {c}", rel=2e-3, elem=4e-3)


WG_CASES = [
    # B, Cin, Cin_pad, H, W, Cout, k, s, p
    (2, 136, 136, 19, 23, 96, 3, 1, 1),
    (1, 232, 232, 16, 20, 64, 3, 1, 1),
    (2, 296, 296, 9, 14, 128, 3, 1, 1),
    (2, 129, 136, 12, 13, 32, 3, 1, 1),
    (2, 128, 128, 10, 11, 21, 3, 1, 1),
    (1, 32, 32, 13, 9, 1, 3, 1, 1),
    (2, 7, 8, 32, 48, 64, 7, 4, 3),
    (2, 64, 64, 16, 24, 128, 3, 2, 1),
    (2, 64, 64, 16, 24, 64, 8, 8, 0),
    (3, 256, 256, 4, 7, 256, 1, 1, 0),
    (2, 64, 64, 40, 52, 512, 1, 1, 0),
    (1, 640, 640, 6, 7, 160, 1, 1, 0),
    # 3x3 / stride 1 on grids >= 32 wide: streaming halo-row kernel (wgrad3x3.hip); ragged widths, segments crossing
    # strip and image boundaries, all wave layouts
    (2, 136, 136, 19, 45, 96, 3, 1, 1),
    (1, 232, 232, 8, 64, 64, 3, 1, 1),
    (2, 296, 304, 33, 70, 128, 3, 1, 1),
    (2, 129, 136, 17, 40, 32, 3, 1, 1),
    (2, 128, 128, 9, 33, 21, 3, 1, 1),
    (3, 48, 48, 40, 32, 128, 3, 1, 1),
    (1, 64, 64, 12, 40, 80, 3, 1, 1),        # 96-channel dy rows (192-byte LDS rows), partial
]


@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad(case):
    lib, lb = L()
    B, Ci, Cp, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(7)
    x = bf(torch.randn(B, Ci, H, W, generator=g))
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    bias = torch.zeros(Co, requires_grad=True)
    y = F.conv2d(x, w, bias, stride=s, padding=p)
    OH, OW = y.shape[2], y.shape[3]
    dy = bf(torch.randn(B, Co, OH, OW, generator=g))
    y.backward(dy)
    Cop = ((Co + 7) // 8) * 8
    xpm, dypm = to_pm(x, ld=Cp), to_pm(dy, ld=Cop)
    dw = zsum(Co, k * k, Cp)
    db = zsum(Co)
    d = lib.WgradDesc()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = P(xpm), Cp, 0, B, H, W, Cp
    d.dy, d.dy_ld, d.dy_coff, d.OH, d.OW, d.Cout = P(dypm), Cop, 0, OH, OW, Co
    d.KH, d.KW, d.stride, d.pad, d.dw, d.dbias = k, k, s, p, P(dw), P(db)
    ok(lb.crd_conv_wgrad(C.byref(d), lib.stream()), "crd_conv_wgrad")
    got = gval(dw)[:, :, :Ci].reshape(Co, k, k, Ci).permute(0, 3, 1, 2)
    assert_close(got, w.grad, f"wgrad {case}", rel=2e-3, elem=4e-3)
    assert_close(gval(db), bias.grad, f"dbias {case}", rel=2e-3, elem=4e-3)
    if Cp > Ci:
        assert int(dw[:, :, Ci:].abs().max()) == 0
    # the accumulators are order-independent: a second run gives the same bits
    dw_b, db_b = zsum(Co, k * k, Cp), zsum(Co)
    d.dw, d.dbias = P(dw_b), P(db_b)
    ok(lb.crd_conv_wgrad(C.byref(d), lib.stream()), "crd_conv_wgrad (second run)")
    assert torch.equal(dw_b, dw) and torch.equal(db_b, db)
    d.dw, d.dbias = P(dw), P(db)
    # streaming 3x3 kernel: per-split copies (plain stores, contents don't-care on entry) summing to the same gradient
    S = lb.crd_conv_wgrad_splits(C.byref(d))
    assert (S > 0) == (k == 3 and s == 1 and W >= 32 and H >= 8)
    if S > 0:
        parts = torch.full((S + 1, Co, k * k, Cp), float("nan"), device="cuda")
        dw2 = torch.full_like(dw, 7)
        d.dw, d.dw_partials, d.dw_partial_capacity = P(dw2), P(parts), S + 1
        db.zero_()
        ok(lb.crd_conv_wgrad(C.byref(d), lib.stream()), "crd_conv_wgrad partials")
        assert int((dw2 - 7).abs().max()) == 0                  # dw untouched
        assert bool(torch.isnan(parts[S]).all())                # copies beyond S untouched
        got2 = parts[:S].sum(0).cpu()[:, :, :Ci].reshape(Co, k, k, Ci).permute(0, 3, 1, 2)
        assert_close(got2, w.grad, f"wgrad partials {case}", rel=2e-3, elem=4e-3)
        assert_close(gval(db), bias.grad, f"dbias partials {case}", rel=2e-3, elem=4e-3)
        # a smaller capacity caps the number of splits (how the two-stream step leaves CUs to the kernels running next to it)
        cap = max(1, S // 2)
        parts2 = torch.full((cap + 1, Co, k * k, Cp), float("nan"), device="cuda")
        d.dw_partials, d.dw_partial_capacity = P(parts2), cap
        S2 = lb.crd_conv_wgrad_splits(C.byref(d))
        assert 1 <= S2 <= cap
        db.zero_()
        ok(lb.crd_conv_wgrad(C.byref(d), lib.stream()), "crd_conv_wgrad capped partials")
        assert bool(torch.isnan(parts2[S2:]).all()) and not bool(torch.isnan(parts2[:S2]).any())
        got3 = parts2[:S2].sum(0).cpu()[:, :, :Ci].reshape(Co, k, k, Ci).permute(0, 3, 1, 2)
        assert_close(got3, w.grad, f"wgrad capped partials {case}", rel=2e-3, elem=4e-3)
        assert_close(gval(db), bias.grad, f"dbias capped partials {case}", rel=2e-3, elem=4e-3)
        # a workgroup budget (how the two-stream step leaves CUs to the kernels running next to it): the splits follow from it, and
        # a short last channel chunk (Cin = 136 here: 8 of 64 channels) gets fewer workgroups than copies -- its block of the
        # remaining copies must come out as zeros, not as whatever the buffer held
        for budget in (10, 25):
            d.wg_budget, d.dw_partial_capacity = budget, 0
            S3 = lb.crd_conv_wgrad_splits(C.byref(d))
            assert 1 <= S3 <= budget
            parts3 = torch.full((S3 + 1, Co, k * k, Cp), float("nan"), device="cuda")
            d.dw_partials, d.dw_partial_capacity = P(parts3), S3
            db.zero_()
            ok(lb.crd_conv_wgrad(C.byref(d), lib.stream()), "crd_conv_wgrad budget")
            assert bool(torch.isnan(parts3[S3:]).all()) and not bool(torch.isnan(parts3[:S3]).any())
            got4 = parts3[:S3].sum(0).cpu()[:, :, :Ci].reshape(Co, k, k, Ci).permute(0, 3, 1, 2)
            assert_close(got4, w.grad, f"wgrad budget {budget} {case}", rel=2e-3, elem=4e-3)
            assert_close(gval(db), bias.grad, f"dbias budget {budget} {case}", rel=2e-3, elem=4e-3)
        d.wg_budget = 0


@pytest.mark.parametrize("H,W,with_add", [(9, 13, False), (16, 40, True)])
def test_head_conv2_forward_backward(H, W, with_add):
    """Depth_Activation.conv_2 (utils.py:283,288): 3x3, 32 -> 1, as a stencil-reduce; backward fused with the sigmoid
    backward of conv_1's output; the data / weight-gradient halves also as calls of their own."""
    lib, lb = L()
    g = torch.Generator().manual_seed(21)
    B = 2
    a = bf(torch.sigmoid(torch.randn(B, 32, H, W, generator=g)))
    w = 0.2 * torch.randn(1, 32, 3, 3, generator=g)
    bias = torch.tensor([0.1])
    apm = to_pm(a)                                                        # [B, H*W, 32] bf16
    wd, bd = w.reshape(-1).contiguous().cuda(), bias.cuda()
    depth = torch.zeros(B, H * W, device="cuda")
    ok(lb.crd_head_conv2_fwd(P(apm), P(wd), P(bd), B, H, W, P(depth), None, 0, 0, lib.stream()), "head_conv2_fwd")
    ref = bf(F.conv2d(a, bf(w), padding=1) + bias)
    assert_close(depth.cpu().view(B, 1, H, W), ref, "depth", rel=4e-3, elem=1e-2)
    # backward
    gd = torch.randn(B, H * W, generator=g)
    add = bf(torch.randn(B, H * W, 8, generator=g)) if with_add else None
    dy = bf(gd + (add[..., 3] if with_add else 0.0)).view(B, 1, H, W)
    aa = a.clone().requires_grad_(True)
    ww = bf(w).clone().requires_grad_(True)
    bb = bias.clone().requires_grad_(True)
    F.conv2d(aa, ww, bb, padding=1).backward(dy)
    dz_ref = aa.grad * a * (1 - a)
    gdd = gd.cuda()
    addd = add.to(torch.bfloat16).cuda() if with_add else None
    rows = zsum(4, 289)
    dz = torch.zeros(B, H * W, 32, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_head_conv2_bwd(P(gdd), P(addd), 8 if with_add else 0, 3 if with_add else 0, P(apm), P(wd), B, H, W, P(dz), P(rows), 4,
                             lib.stream()), "head_conv2_bwd")
    assert_close(dz.float().cpu().view(B, H, W, 32).permute(0, 3, 1, 2), dz_ref, "dz", rel=6e-3, elem=2e-2)
    r = gval(rows.sum(0))
    assert_close(r[:288].view(32, 9), ww.grad.view(32, 9), "conv_2 dw", rel=2e-3, elem=4e-3)
    assert_close(r[288:], bb.grad, "conv_2 dbias", rel=2e-3, elem=4e-3)
    # the two halves on their own
    rows2, dz2 = torch.zeros_like(rows), torch.zeros_like(dz)
    ok(lb.crd_head_conv2_bwd_data(P(gdd), P(addd), 8 if with_add else 0, 3 if with_add else 0, P(apm), P(wd), B, H, W, P(dz2),
                                  lib.stream()), "head_conv2_bwd_data")
    ok(lb.crd_head_conv2_wgrad(P(gdd), P(addd), 8 if with_add else 0, 3 if with_add else 0, P(apm), B, H, W, P(rows2), 4, lib.stream()),
       "head_conv2_wgrad")
    assert torch.equal(dz2, dz)
    assert torch.equal(rows2.sum(0), rows.sum(0)), "rows (wgrad half): fixed-point sums are order-independent"


@pytest.mark.parametrize("C_,gmul,xf32,act", [(64, 1, 1, 0), (96, 1, 0, 1), (512, 8, 0, 1), (160, 1, 1, 0), (640, 4, 0, 1),
                                               (1024, 4, 0, 0)])
def test_groupnorm_forward_backward(C_, gmul, xf32, act):
    lib, lb = L()
    g = torch.Generator().manual_seed(1)
    B, Pn = 2, 77
    x = torch.randn(B, C_, Pn, generator=g) * 1.5 + 0.3
    if not xf32:
        x = bf(x)
    gamma = (1 + 0.1 * torch.randn(C_, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C_, generator=g)).requires_grad_(True)
    mask = (torch.rand(B, C_, generator=g) < 0.8).float() / 0.8
    xr = x.clone().requires_grad_(True)
    groups = C_ // (16 * gmul)
    yref = F.group_norm(xr, groups, gamma, beta, 1e-5)
    if act:
        yref = F.gelu(yref)
    yref = yref * mask.view(B, C_, 1)
    dy = bf(torch.randn(B, C_, Pn, generator=g))
    yref.backward(dy)
    xpm = x.permute(0, 2, 1).contiguous()
    xd = (xpm if xf32 else xpm.to(torch.bfloat16)).cuda()
    stats = zsum(B, C_ // 16, 2)
    chan = zsum(B, C_, 2)
    ok(lb.crd_gn_stats(P(xd), xf32, C_, 0, B, Pn, C_, P(stats), P(chan), lib.stream()), "gn_stats")
    ref_chan = torch.stack([x.sum(2), (x ** 2).sum(2)], -1)
    assert_close(sval(chan), ref_chan, "chan sums", rel=1e-4, elem=1e-4)
    assert_close(sval(stats), ref_chan.reshape(B, C_ // 16, 16, 2).sum(2), "g16 stats", rel=1e-4, elem=1e-4)
    y = torch.zeros(B, Pn, C_ + 8, dtype=torch.bfloat16, device="cuda")
    gc, bc, mc = gamma.detach().cuda(), beta.detach().cuda(), mask.cuda()
    ok(lb.crd_gn_apply(P(xd), xf32, C_, 0, B, Pn, C_, P(stats), gmul, P(gc), P(bc), act, P(mc), P(y), 0, C_ + 8, 8,
                       lib.stream()), "gn_apply")
    assert_close(y[..., 8:].float().cpu().permute(0, 2, 1), yref.detach(), "gn_apply", rel=4e-3, elem=1e-2)
    # fp32 output variant
    yf = torch.zeros(B, Pn, C_, device="cuda")
    ok(lb.crd_gn_apply(P(xd), xf32, C_, 0, B, Pn, C_, P(stats), gmul, P(gc), P(bc), act, P(mc), P(yf), 1, C_, 0,
                       lib.stream()), "gn_apply f32")
    assert_close(yf.cpu().permute(0, 2, 1), yref.detach(), "gn_apply f32", rel=1e-4, elem=1e-4)
    # backward
    dyd = dy.permute(0, 2, 1).contiguous().to(torch.bfloat16).cuda()
    r = zsum(B * C_ * 2 + B * groups * 2)
    scratch = torch.full((B * 64 * 2 * C_,), 3.0, device="cuda") if C_ % 32 == 0 else None   # both reduction paths
    ok(lb.crd_gn_bwd_reduce(P(xd), xf32, C_, 0, P(dyd), 0, C_, 0, B, Pn, C_, P(stats), gmul, P(gc), P(bc), act, P(mc), P(r),
                            P(scratch), scratch.numel() if scratch is not None else 0, lib.stream()), "gn_bwd_reduce")
    dgam, dbet = torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    dx = torch.zeros(B, Pn, C_, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_gn_bwd_apply(P(xd), xf32, C_, 0, P(dyd), 0, C_, 0, B, Pn, C_, P(stats), gmul, P(gc), P(bc), act, P(mc), P(r),
                           P(dgam), P(dbet), P(dx), 0, C_, 0, 0, None, 0, None, lib.stream()), "gn_bwd_apply")
    assert_close(dx.float().cpu().permute(0, 2, 1), xr.grad, "gn dx", rel=6e-3, elem=2e-2)
    assert_close(dgam.cpu(), gamma.grad, "dgamma", rel=2e-3, elem=4e-3)
    assert_close(dbet.cpu(), beta.grad, "dbeta", rel=2e-3, elem=4e-3)
    # fp32 accumulate variant
    base = torch.randn(B, Pn, C_, generator=g)
    dxf = base.clone().cuda()
    ok(lb.crd_gn_bwd_apply(P(xd), xf32, C_, 0, P(dyd), 0, C_, 0, B, Pn, C_, P(stats), gmul, P(gc), P(bc), act, P(mc), P(r),
                           None, None, P(dxf), 1, C_, 0, 1, None, 0, None, lib.stream()), "gn_bwd_apply f32")
    assert_close(dxf.cpu() - base, xr.grad.permute(0, 2, 1), "gn dx f32 acc", rel=1e-3, elem=2e-3)
    # ... with the scaled bf16 copy of the finished gradient
    dxf2, dx2 = base.clone().cuda(), torch.zeros(B, Pn, C_ + 8, dtype=torch.bfloat16, device="cuda")
    sc2 = torch.tensor([0.5 + 0.25 * i for i in range(B)]).cuda()
    ok(lb.crd_gn_bwd_apply(P(xd), xf32, C_, 0, P(dyd), 0, C_, 0, B, Pn, C_, P(stats), gmul, P(gc), P(bc), act, P(mc), P(r),
                           None, None, P(dxf2), 1, C_, 0, 1, P(dx2), C_ + 8, P(sc2), lib.stream()), "gn_bwd_apply f32 + copy")
    assert torch.equal(dxf2, dxf)
    assert torch.equal(dx2[..., :C_], (dxf * sc2.view(B, 1, 1)).to(torch.bfloat16))


@pytest.mark.parametrize("C_,H,W", [(64, 9, 13), (512, 8, 12), (160, 5, 7), (64, 11, 45), (80, 9, 33), (256, 17, 64)])
def test_dwconv(C_, H, W):
    lib, lb = L()
    g = torch.Generator().manual_seed(2)
    B = 2
    x = bf(torch.randn(B, C_, H, W, generator=g))
    w = (torch.randn(C_, 1, 3, 3, generator=g) / 3).requires_grad_(True)
    b = (torch.randn(C_, generator=g) * 0.1).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yref = F.conv2d(xr, w, b, padding=1, groups=C_)
    dy = bf(torch.randn(B, C_, H, W, generator=g))
    yref.backward(dy)
    w9 = w.detach().reshape(C_, 9).t().contiguous().cuda()
    xd = to_pm(x)
    y = torch.zeros(B, H, W, C_, dtype=torch.bfloat16, device="cuda")
    stats = zsum(B, C_ // 16, 2)
    bc = b.detach().cuda()
    ok(lb.crd_dwconv3x3(P(xd), B, H, W, C_, P(w9), P(bc), 0, P(y), P(stats), None, 1, None, None, None, None, None, None, lib.stream()), "dwconv")
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert_close(got, yref.detach(), "dwconv")
    gq = got.reshape(B, C_ // 16, 16, H * W)
    assert_close(sval(stats), torch.stack([gq.sum((2, 3)), (gq ** 2).sum((2, 3))], -1), "dwconv stats", rel=1e-3, elem=2e-3)
    dyd = to_pm(dy)
    dx = torch.zeros_like(y)
    ok(lb.crd_dwconv3x3(P(dyd), B, H, W, C_, P(w9), None, 1, P(dx), None, None, 1, None, None, None, None, None, None, lib.stream()), "dwconv dgrad")
    assert_close(dx.float().cpu().permute(0, 3, 1, 2), xr.grad, "dwconv dx")
    for R in (1, 5):       # accumulator copies the workgroups spread their atomics over; the gradient is their sum
        dw10 = zsum(R, 10, C_)
        ok(lb.crd_dwconv3x3_wgrad(P(xd), P(dyd), B, H, W, C_, P(dw10), R, None, 1, None, None, lib.stream()), "dwconv wgrad")
        tot = gval(dw10.sum(0))
        assert_close(tot[:9].t().reshape(C_, 1, 3, 3), w.grad, "dwconv dw", rel=2e-3, elem=4e-3)
        assert_close(tot[9], b.grad, "dwconv db", rel=2e-3, elem=4e-3)


@pytest.mark.parametrize("C_,H,W,gmul", [(64, 9, 13, 1), (256, 17, 40, 4), (80, 8, 33, 1)])
def test_dwconv_with_fused_input_groupnorm(C_, H, W, gmul):
    """in_stats != NULL: GroupNorm of the input applied while the halo is staged == crd_gn_apply followed by the plain
    kernel, bit for bit (forward, statistics and weight gradient)."""
    lib, lb = L()
    g = torch.Generator().manual_seed(8)
    B = 2
    if (C_ // 16) % gmul:
        gmul = 1
    xd = to_pm(bf(torch.randn(B, C_, H, W, generator=g) * 2 + 0.5))
    dyd = to_pm(bf(torch.randn(B, C_, H, W, generator=g)))
    w9 = (torch.randn(9, C_, generator=g) / 3).cuda()
    bc = (torch.randn(C_, generator=g) * 0.1).cuda()
    gam, bet = (1 + 0.1 * torch.randn(C_, generator=g)).cuda(), (0.1 * torch.randn(C_, generator=g)).cuda()
    st_in = zsum(B, C_ // 16, 2)
    ok(lb.crd_gn_stats(P(xd), 0, C_, 0, B, H * W, C_, P(st_in), None, lib.stream()), "gn_stats")
    xn = torch.zeros_like(xd)
    ok(lb.crd_gn_apply(P(xd), 0, C_, 0, B, H * W, C_, P(st_in), gmul, P(gam), P(bet), 0, None, P(xn), 0, C_, 0, lib.stream()), "gn_apply")
    outs = []
    for src, nrm in ((xn, (None, 1, None, None)), (xd, (P(st_in), gmul, P(gam), P(bet)))):
        y = torch.zeros(B, H, W, C_, dtype=torch.bfloat16, device="cuda")
        st = zsum(B, C_ // 16, 2)
        ok(lb.crd_dwconv3x3(P(src), B, H, W, C_, P(w9), P(bc), 0, P(y), P(st), *nrm, None, None, None, None, lib.stream()), "dwconv")
        dw10 = zsum(1, 10, C_)
        ok(lb.crd_dwconv3x3_wgrad(P(src), P(dyd), B, H, W, C_, P(dw10), 1, *nrm, lib.stream()), "dwconv wgrad")
        outs.append((y.float().cpu(), st.cpu(), dw10.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]), "fused input norm changes the forward result"
    assert torch.equal(outs[1][1], outs[0][1]), "stats"
    assert torch.equal(outs[1][2], outs[0][2]), "dw10"
    # data gradient with the reduce phase of the following GroupNorm backward fused in == plain kernel + crd_gn_bwd_reduce
    G = C_ // 16
    dxa = torch.zeros(B, H, W, C_, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_dwconv3x3(P(dyd), B, H, W, C_, P(w9), None, 1, P(dxa), None, None, 1, None, None, None, None, None, None, lib.stream()), "dgrad")
    r_ref = zsum(B * C_ * 2 + B * G * 2)
    ok(lb.crd_gn_bwd_reduce(P(xd), 0, C_, 0, P(dxa), 0, C_, 0, B, H * W, C_, P(st_in), 1, P(gam), P(bet), 0, None, P(r_ref), None, 0,
                            lib.stream()), "gn_bwd_reduce")
    dxb = torch.zeros_like(dxa)
    r_fused = torch.zeros_like(r_ref)
    ok(lb.crd_dwconv3x3(P(dyd), B, H, W, C_, P(w9), None, 1, P(dxb), None, None, 1, None, None, P(xd), P(st_in), P(gam), P(r_fused),
                        lib.stream()), "dgrad + fused reduce")
    assert torch.equal(dxa, dxb)
    assert_close(gval(r_fused), gval(r_ref), "fused gn-bwd reduce", rel=2e-4, elem=2e-4)


@pytest.mark.parametrize("N,M,heads,d", [(200, 104, 1, 64), (130, 104, 2, 64), (70, 35, 4, 40), (104, 104, 8, 32), (300, 1450, 8, 64), (97, 325, 5, 64),
                                         (64, 129, 1, 24)])
def test_attention_scores_and_backward(N, M, heads, d):
    lib, lb = L()
    g = torch.Generator().manual_seed(4)
    B, C_ = 2, heads * d
    scale = d ** -0.5
    q = bf(torch.randn(B, N, C_, generator=g))
    k = bf(torch.randn(B, M, C_, generator=g))
    qh = q.reshape(B, N, heads, d).permute(0, 2, 1, 3)
    kh = k.reshape(B, M, heads, d).permute(0, 2, 3, 1)
    att = bf(bf(qh @ kh) * scale)                       # [B,h,N,M]
    smax, imax = att.max(-1)
    S_ref = smax.sum(1)                                 # [B,N]
    qd, kd = q.to(torch.bfloat16).cuda(), k.to(torch.bfloat16).cuda()
    S = torch.zeros(B, N, device="cuda")
    idx = torch.zeros(B, N, heads, dtype=torch.int16, device="cuda")
    ok(lb.crd_attn_scores(P(qd), P(kd), B, N, M, heads, d, scale, P(S), P(idx), lib.stream()), "attn_scores")
    assert_close(S.cpu(), S_ref, "S", rel=2e-3, elem=1e-2)
    # argmax may differ only where two scores tie after bf16 rounding: check the chosen score is the max
    chosen = torch.gather(att, 3, idx.cpu().long().permute(0, 2, 1).unsqueeze(-1)).squeeze(-1)
    assert float((chosen - smax).abs().max()) <= 1e-2 * float(smax.abs().max())
    if C_ % 16 == 0:      # fused launch: scores + the rank-one value path (crd_attn_xbar_proj) in cdiv(C, 64) extra workgroups per sample
        chan = to_stat(torch.randn(B, C_, 2, generator=g)).cuda()
        st = to_stat(torch.stack([torch.randn(B, C_ // 16, generator=g), 20.0 + torch.rand(B, C_ // 16, generator=g)], -1) * N).cuda()
        gam, bet = (1 + 0.1 * torch.randn(C_, generator=g)).cuda(), (0.1 * torch.randn(C_, generator=g)).cuda()
        wf = (0.2 * torch.randn(C_, C_, generator=g)).to(torch.bfloat16).cuda()
        xb0, u0 = torch.zeros(B, C_, dtype=torch.bfloat16, device="cuda"), torch.zeros(B, C_, device="cuda")
        ok(lb.crd_attn_xbar_proj(P(chan), P(st), P(gam), P(bet), P(wf), B, N, C_, P(xb0), P(u0), lib.stream()), "xbar_proj")
        S2, idx2, xb1, u1 = torch.zeros_like(S), torch.zeros_like(idx), torch.zeros_like(xb0), torch.zeros_like(u0)
        ok(lb.crd_attn_fwd(P(qd), P(kd), B, N, M, heads, d, scale, P(S2), P(idx2), P(chan), P(st), P(gam), P(bet), P(wf),
                           P(xb1), P(u1), lib.stream()), "attn_fwd (fused)")
        assert torch.equal(S2, S) and torch.equal(idx2, idx) and torch.equal(xb1, xb0) and torch.equal(u1, u0)
        assert bool(torch.isfinite(u0).all())
    # backward of the score path
    dS = torch.randn(B, N, generator=g)
    dq = torch.zeros(B, N, C_, dtype=torch.bfloat16, device="cuda")
    dk = zsum(B, M, C_)
    dSc = dS.cuda()
    ok(lb.crd_attn_scores_bwd(P(qd), P(kd), P(dSc), P(idx), B, N, M, heads, d, scale, P(dq), P(dk), None, lib.stream()),
       "attn_scores_bwd")
    ii = idx.cpu().long()
    dq_ref = torch.zeros(B, N, heads, d)
    dk_ref = torch.zeros(B, M, heads, d)
    k4, q4 = k.reshape(B, M, heads, d), q.reshape(B, N, heads, d)
    for b in range(B):
        for h in range(heads):
            dq_ref[b, :, h] = scale * dS[b].unsqueeze(1) * k4[b, ii[b, :, h], h]
            dk_ref[b, :, h].index_add_(0, ii[b, :, h], scale * dS[b].unsqueeze(1) * q4[b, :, h])
    assert_close(dq.float().cpu(), dq_ref.reshape(B, N, C_), "dq")
    assert_close(gval(dk), dk_ref.reshape(B, M, C_), "dk", rel=1e-4, elem=1e-4)
    dkb0 = torch.zeros(B, M, C_, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_gsum_to_bf16(P(dk), P(dkb0), B * M * C_, lib.stream()), "gsum_to_bf16")
    assert torch.equal(dkb0.cpu(), gval(dk).to(torch.bfloat16))
    # partial-accumulator variant: per-workgroup stores, folded together with the bf16 conversion
    nparts = lb.crd_attn_scores_bwd_partials(B, N, M, heads, d)
    assert nparts >= 1
    parts = torch.full((nparts, B, M, C_), float("nan"), device="cuda")
    dq2 = torch.zeros_like(dq)
    ok(lb.crd_attn_scores_bwd(P(qd), P(kd), P(dSc), P(idx), B, N, M, heads, d, scale, P(dq2), None, P(parts), lib.stream()),
       "attn_scores_bwd partials")
    assert torch.equal(dq2, dq)
    dkb = torch.zeros(B, M, C_, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_sum_partials_bf16(P(parts), nparts, B * M * C_, P(dkb), B * M * C_, lib.stream()), "sum_partials")
    assert_close(dkb.float().cpu(), bf(dk_ref.reshape(B, M, C_)), "dk from partials (bf16)", rel=4e-3, elem=1e-2)
    # fused launch: the same plus the rank-one vector path (crd_attn_vec_bwd) in one extra workgroup per sample
    t = to_grad(torch.randn(B, C_, generator=g)).cuda()
    wd = torch.zeros(C_, C_ + 8, dtype=torch.bfloat16, device="cuda")
    wd[:, :C_] = (0.2 * torch.randn(C_, C_, generator=g)).to(torch.bfloat16)
    tb0, es0 = torch.zeros(B, C_, dtype=torch.bfloat16, device="cuda"), torch.zeros(B, C_, device="cuda")
    ok(lb.crd_attn_vec_bwd(P(t), P(wd), B, C_, C_ + 8, 1.0 / N, P(tb0), P(es0), lib.stream()), "attn_vec_bwd")
    parts3 = torch.full((nparts, B, M, C_), float("nan"), device="cuda")
    dq3, tb1, es1 = torch.zeros_like(dq), torch.zeros_like(tb0), torch.zeros_like(es0)
    ok(lb.crd_attn_bwd(P(qd), P(kd), P(dSc), P(idx), B, N, M, heads, d, scale, P(dq3), None, P(parts3), P(t), P(wd), C_ + 8,
                       1.0 / N, P(tb1), P(es1), lib.stream()), "attn_bwd (fused)")
    assert torch.equal(dq3, dq) and torch.equal(tb1, tb0) and torch.equal(es1, es0)
    # (the order of a key's pixel list comes from LDS atomics, but its products are added in fixed point: bit for bit)
    assert torch.equal(parts3, parts), "dk partials (fused launch)"


def test_attention_output_path():
    lib, lb = L()
    g = torch.Generator().manual_seed(6)
    B, N, C_ = 2, 150, 160
    x = torch.randn(B, N, C_, generator=g)
    u, S, bp = torch.randn(B, C_, generator=g), torch.randn(B, N, generator=g), torch.randn(C_, generator=g)
    dp = torch.tensor([1.0 / 0.9, 0.0])
    x1 = torch.zeros(B, N, C_, device="cuda")
    xc, uc, Sc, bpc, dpc = x.cuda(), u.cuda(), S.cuda(), bp.cuda(), dp.cuda()
    ok(lb.crd_attn_out_residual(P(xc), P(uc), P(Sc), P(bpc), P(dpc), B, N, C_, P(x1), lib.stream()), "attn_out_residual")
    ref = x + dp.view(B, 1, 1) * bf(u.unsqueeze(1) * S.unsqueeze(2) + bp)
    assert_close(x1.cpu(), ref, "x1", rel=1e-5, elem=1e-5)
    if C_ % 16 == 0:      # fused variant: same x1 plus the g16 sums crd_gn_stats(x1) would produce
        x1b, st = torch.zeros(B, N, C_, device="cuda"), zsum(B, C_ // 16, 2)
        ok(lb.crd_attn_out_residual_stats(P(xc), P(uc), P(Sc), P(bpc), P(dpc), B, N, C_, P(x1b), P(st), lib.stream()), "attn_out_residual_stats")
        assert torch.equal(x1b, x1)
        xd_ = x1.double().cpu().view(B, N, C_ // 16, 16)
        assert_close(sval(st), torch.stack([xd_.sum((1, 3)), (xd_ * xd_).sum((1, 3))], -1), "norm2 sums", rel=1e-5, elem=1e-5)
    dx1 = torch.randn(B, N, C_, generator=g)
    t, dbp, dS = zsum(B, C_), zsum(B, C_), torch.zeros(B, N, device="cuda")
    dx1c = dx1.cuda()
    ok(lb.crd_attn_out_bwd(P(dx1c), P(uc), P(Sc), P(dpc), B, N, C_, P(t), P(dbp), P(dS), lib.stream()), "attn_out_bwd")
    dy = dp.view(B, 1, 1) * dx1
    assert_close(gval(t), (dy * S.unsqueeze(2)).sum(1), "t", rel=1e-4, elem=1e-4)
    assert_close(gval(dbp), dy.sum(1), "dbp rows (one per sample)", rel=1e-4, elem=1e-4)
    assert_close(dS.cpu(), (dy * u.unsqueeze(1)).sum(2), "dS", rel=1e-4, elem=1e-4)
    # xbar
    gamma, beta = 1 + 0.1 * torch.randn(C_, generator=g), 0.1 * torch.randn(C_, generator=g)
    xd = x.cuda()
    stats, chan = zsum(B, C_ // 16, 2), zsum(B, C_, 2)
    ok(lb.crd_gn_stats(P(xd), 1, C_, 0, B, N, C_, P(stats), P(chan), lib.stream()), "gn_stats")
    # crd_attn_out_bwd_gn = crd_gn_bwd_apply (Block.norm2: fp32 input, accumulating fp32 output) + crd_attn_out_bwd in one launch:
    # the same dx1, t, dbp, dS and the GroupNorm's parameter gradients
    dxn = torch.randn(B, N, C_, generator=g).to(torch.bfloat16).cuda()
    gc_, bc_ = gamma.cuda(), beta.cuda()
    r = zsum(B * C_ * 2 + B * (C_ // 16) * 2)
    ok(lb.crd_gn_bwd_reduce(P(xd), 1, C_, 0, P(dxn), 0, C_, 0, B, N, C_, P(stats), 1, P(gc_), P(bc_), 0, None, P(r), None, 0, lib.stream()), "reduce")
    dxa, dga, dba = dx1.clone().cuda(), torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    ok(lb.crd_gn_bwd_apply(P(xd), 1, C_, 0, P(dxn), 0, C_, 0, B, N, C_, P(stats), 1, P(gc_), P(bc_), 0, None, P(r), P(dga), P(dba),
                           P(dxa), 1, C_, 0, 1, None, 0, None, lib.stream()), "apply")
    ta, dbpa, dSa = zsum(B, C_), zsum(B, C_), torch.zeros(B, N, device="cuda")
    ok(lb.crd_attn_out_bwd(P(dxa), P(uc), P(Sc), P(dpc), B, N, C_, P(ta), P(dbpa), P(dSa), lib.stream()), "attn_out_bwd")
    dxb, dgb, dbb = dx1.clone().cuda(), torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    tb, dbpb, dSb = zsum(B, C_), zsum(B, C_), torch.zeros(B, N, device="cuda")
    ok(lb.crd_attn_out_bwd_gn(P(dxb), P(uc), P(Sc), P(dpc), B, N, C_, P(tb), P(dbpb), P(dSb), P(xd), P(dxn), P(stats), P(gc_), P(r),
                              P(dgb), P(dbb), lib.stream()), "attn_out_bwd_gn")
    # (to fp32 rounding: the two kernels contract a*b+c differently; the parameter gradients are the same integer sums)
    assert_close(dxb.cpu(), dxa.cpu(), "fused dx1", rel=1e-6, elem=2e-6)
    assert_close(gval(tb), gval(ta), "fused t", rel=1e-5, elem=1e-5)
    assert_close(gval(dbpb), gval(dbpa), "fused dbp", rel=1e-5, elem=1e-5)
    assert_close(dSb.cpu(), dSa.cpu(), "fused dS", rel=1e-5, elem=1e-5)
    assert torch.equal(dgb, dga) and torch.equal(dbb, dba) and float(dga.abs().sum()) > 0
    assert not torch.equal(dxa.cpu(), dx1)
    xbar = torch.zeros(B, C_, dtype=torch.bfloat16, device="cuda")
    gac, bec = gamma.cuda(), beta.cuda()
    ok(lb.crd_attn_xbar(P(chan), P(stats), P(gac), P(bec), B, N, C_, P(xbar), lib.stream()), "xbar")
    xn = F.group_norm(x.permute(0, 2, 1), C_ // 16, gamma, beta, 1e-5)
    assert_close(xbar.float().cpu(), xn.mean(2), "xbar", rel=4e-3, elem=1e-2)
    # fused xbar -> proj, and its backward (packed weights: forward form [co][ci], data-gradient form [ci][co_pad])
    wp = bf(0.2 * torch.randn(C_, C_, generator=g))
    wf = wp.to(torch.bfloat16).cuda().contiguous()
    pad = 8
    wd = torch.zeros(C_, C_ + pad, dtype=torch.bfloat16, device="cuda")
    wd[:, :C_] = wp.t().to(torch.bfloat16)
    xbar2, u2 = torch.zeros_like(xbar), torch.zeros(B, C_, device="cuda")
    ok(lb.crd_attn_xbar_proj(P(chan), P(stats), P(gac), P(bec), P(wf), B, N, C_, P(xbar2), P(u2), lib.stream()), "xbar_proj")
    assert torch.equal(xbar2, xbar)
    assert_close(u2.cpu(), xbar.float().cpu() @ wp.t(), "u = Wp xbar", rel=1e-5, elem=1e-4)
    tb, es = torch.zeros(B, C_, dtype=torch.bfloat16, device="cuda"), torch.zeros(B, C_, device="cuda")
    ok(lb.crd_attn_vec_bwd(P(t), P(wd), B, C_, C_ + pad, 1.0 / N, P(tb), P(es), lib.stream()), "attn_vec_bwd")
    assert torch.equal(tb.cpu(), gval(t).to(torch.bfloat16))
    assert_close(es.cpu(), (tb.float().cpu() @ wp) / N, "es = Wp^T tb / N", rel=1e-5, elem=1e-4)


@pytest.mark.parametrize("H,W,C_", [(5, 7, 16), (8, 13, 136), (3, 4, 8), (16, 26, 128), (9, 16, 40), (19, 37, 72)])
def test_bicubic(H, W, C_):
    lib, lb = L()
    g = torch.Generator().manual_seed(8)
    B = 2
    x = bf(torch.randn(B, C_, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    yref = F.interpolate(xr, scale_factor=2, mode="bicubic")
    dy = bf(torch.randn(B, C_, 2 * H, 2 * W, generator=g))
    yref.backward(dy)
    xd = to_pm(x, ld=C_ + 8, coff=8)
    y = torch.zeros(B, 2 * H, 2 * W, C_ + 16, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_bicubic2x(P(xd), C_ + 8, 8, B, H, W, C_, P(y), C_ + 16, 0, lib.stream()), "bicubic")
    assert_close(y[..., :C_].float().cpu().permute(0, 3, 1, 2), yref.detach(), "bicubic", rel=3e-3, elem=8e-3)
    dyd = to_pm(dy)
    dx = torch.zeros(B, H, W, C_, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_bicubic2x_bwd(P(dyd), C_, 0, B, H, W, C_, P(dx), C_, 0, 0, lib.stream()), "bicubic_bwd")
    assert_close(dx.float().cpu().permute(0, 3, 1, 2), xr.grad, "bicubic bwd", rel=3e-3, elem=8e-3)


def test_layout_and_small_ops():
    lib, lb = L()
    g = torch.Generator().manual_seed(9)
    B, C_, H, W = 2, 7, 6, 10
    x = torch.randn(B, C_, H, W, generator=g)
    y = torch.full((B, H, W, 16), 7.0, dtype=torch.bfloat16, device="cuda")
    xc = x.cuda()
    ok(lb.crd_nchw_to_pm(P(xc), B, C_, H, W, P(y), 16, 8, 8, lib.stream()), "nchw_to_pm")
    assert torch.equal(y[..., 8:15].float().cpu(), bf(x).permute(0, 2, 3, 1))
    assert float(y[..., 15].float().abs().max()) == 0 and float((y[..., :8].float() - 7).abs().max()) == 0
    # pm -> nchw (fp32 logits with ld 24)
    lg = torch.randn(B, H * W, 24, generator=g)
    out = torch.zeros(B, 21, H, W, device="cuda")
    lgc = lg.cuda()
    ok(lb.crd_pm_to_nchw(P(lgc), 1, 24, 0, B, 21, H, W, P(out), lib.stream()), "pm_to_nchw")
    assert torch.equal(out.cpu(), lg[..., :21].reshape(B, H, W, 21).permute(0, 3, 1, 2))
    # seg argmax
    sm = torch.zeros(B, H * W, 8, dtype=torch.bfloat16, device="cuda")
    ok(lb.crd_seg_argmax(P(lgc), 24, B, H * W, 21, 21, P(sm), 0, 8, 3, lib.stream()), "seg_argmax")
    ref = (lg[..., :21].argmax(-1) / 21)
    assert torch.equal(sm[..., 3].float().cpu(), bf(ref))
    # slice copy + accumulate
    a = bf(torch.randn(30, 24, generator=g))
    dst = torch.zeros(30, 40, dtype=torch.bfloat16, device="cuda")
    ad = a.to(torch.bfloat16).cuda()
    ok(lb.crd_slice_copy(P(ad), 24, 8, P(dst), 40, 16, 30, 16, 0, lib.stream()), "slice_copy")
    ok(lb.crd_slice_copy(P(ad), 24, 8, P(dst), 40, 16, 30, 16, 1, lib.stream()), "slice_copy acc")
    assert torch.equal(dst[:, 16:32].float().cpu(), bf(2 * a[:, 8:24]))
    # f32 -> bf16 rows with per-sample scale
    src = torch.randn(2 * 15, 1, generator=g)
    sc = torch.tensor([0.5, 2.0])
    d2 = torch.zeros(30, 8, dtype=torch.bfloat16, device="cuda")
    srcc, scc = src.cuda(), sc.cuda()
    ok(lb.crd_f32_to_bf16_rows(P(srcc), 1, P(d2), 8, 0, 30, 1, P(scc), 15, None, 0, 0, lib.stream()), "f32_to_bf16_rows")
    assert torch.equal(d2[:, 0].float().cpu(), bf(src[:, 0] * sc.repeat_interleave(15)))
    # sigmoid backward
    av = bf(torch.rand(64, generator=g))
    dv = bf(torch.randn(64, generator=g))
    dd = dv.to(torch.bfloat16).cuda()
    avc = av.to(torch.bfloat16).cuda()
    ok(lb.crd_sigmoid_bwd(P(avc), P(dd), 64, lib.stream()), "sigmoid_bwd")
    assert_close(dd.float().cpu(), dv * av * (1 - av), "sigmoid bwd")


def test_losses():
    from camradepth_amd import synth
    lib, lb = L()
    g = torch.Generator().manual_seed(10)
    b = synth.make_batch(2, 24, 40, seed=3)
    pred = (torch.rand(2, 1, 24, 40, generator=g) * 3.4 - 1.2).requires_grad_(True)
    tgt = b["gt_full"]
    m = tgt > 0
    l1 = F.smooth_l1_loss(pred[m], tgt[m])
    mse = ((tgt - pred)[m] ** 2).mean()
    l1.backward()
    acc = zsum(4)
    pd, td = pred.detach().cuda(), tgt.cuda()
    ok(lb.crd_masked_l1_fwd(P(pd), P(td), pd.numel(), P(acc), lib.stream()), "l1 fwd")
    a = sval(acc)
    acc_b = zsum(4)         # order-independent sums: the same bits on a second run
    ok(lb.crd_masked_l1_fwd(P(pd), P(td), pd.numel(), P(acc_b), lib.stream()), "l1 fwd (second run)")
    assert torch.equal(acc_b, acc)
    np.testing.assert_allclose(float(a[0] / a[1]), float(l1), rtol=1e-5)
    np.testing.assert_allclose(float(a[2] / a[1]), float(mse), rtol=1e-5)
    assert int(a[1]) == int(m.sum())
    dpred = torch.zeros_like(pd)
    gout = torch.tensor([2.0], device="cuda")
    ok(lb.crd_masked_l1_bwd(P(pd), P(td), pd.numel(), P(acc), P(gout), 0.5, P(dpred), lib.stream()), "l1 bwd")
    assert_close(dpred.cpu(), pred.grad, "dpred", rel=1e-5, elem=1e-5)
    # focal CE
    logits = (torch.randn(2, 21, 24, 40, generator=g) * 2).requires_grad_(True)
    ce = F.cross_entropy(logits, b["seg"], ignore_index=255)
    focal = (1 - torch.exp(-ce)) ** 2 * ce
    focal.backward()
    acc2 = zsum(4)
    ld, lab = logits.detach().cuda(), b["seg"].cuda()
    ok(lb.crd_ce_fwd(P(ld), P(lab), 2, 21, 24 * 40, P(acc2), lib.stream()), "ce fwd")
    a2 = sval(acc2)
    np.testing.assert_allclose(float(a2[0] / a2[1]), float(ce), rtol=1e-5)
    dl = torch.zeros_like(ld)
    ok(lb.crd_ce_focal_bwd(P(ld), P(lab), 2, 21, 24 * 40, P(acc2), None, 1.0, P(dl), lib.stream()), "ce bwd")
    assert_close(dl.cpu(), logits.grad, "dlogits", rel=1e-4, elem=1e-4)


def test_diffgradnorm_matches_golden_trajectory():
    """40 steps of the reference optimizer (tests/golden/diffgradnorm_40steps.npz), flat-buffer kernel."""
    from tests.util import load_npz
    lib, lb = L()
    gd = load_npz("diffgradnorm_40steps.npz")
    ps = [torch.from_numpy(gd[f"p{j}_init"]).reshape(-1) for j in range(3)]
    # add a large tensor that spans several 4096-element chunks, checked against the CPU oracle
    from oracle import optim as ooptim
    g = torch.Generator().manual_seed(0)
    big = torch.randn(10000, generator=g)
    big_ref = big.clone()
    big_state = ooptim.new_state(big_ref)
    sizes = [p.numel() for p in ps] + [big.numel()]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    flat = torch.cat(ps + [big]).cuda()
    n = flat.numel()
    m, v, pg = (torch.zeros(n, device="cuda") for _ in range(3))
    egn, fac = (torch.zeros(4, device="cuda") for _ in range(2))
    b2s, b2c = [], []
    for t, sz in enumerate(sizes):
        for c in range((sz + 4095) // 4096):
            b2s.append(t)
            b2c.append(c)
    seg = torch.from_numpy(np.stack([off[:-1], off[1:]], 1).copy()).cuda()
    b2s_d, b2c_d = torch.tensor(b2s, dtype=torch.int32).cuda(), torch.tensor(b2c, dtype=torch.int32).cuda()
    nsq = torch.full((len(b2s),), float("nan"), device="cuda")     # scratch: per-workgroup parts of ||g||^2, contents don't-care
    for it in range(40):
        lr, b1, b2 = (float(z) for z in gd["hp"][it])
        gb = torch.randn(10000, generator=g) * (0.02 if 10 <= it < 14 else 1.0)
        grads = torch.cat([torch.from_numpy(gd[f"p{j}_grads"][it]).reshape(-1) for j in range(3)] + [gb]).cuda()
        ok(lb.crd_diffgradnorm_step(P(flat), P(grads), P(m), P(v), P(pg), P(egn), P(nsq), P(fac), P(seg), P(b2s_d), P(b2c_d),
                                    4, len(b2s), None, lr, b1, b2, 1e-8, 0.0, it + 1, None, lib.stream()), "dgn")
        ooptim.step_tensor(big_ref, gb, big_state, lr, b1, b2)
        fc = flat.cpu()
        for j in range(3):
            np.testing.assert_allclose(fc[off[j]:off[j + 1]].numpy(), gd[f"p{j}_traj"][it].reshape(-1), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(fc[off[3]:off[4]].numpy(), big_ref.numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(egn.cpu().numpy()[:3], gd["exp_grad_norm"][it], rtol=1e-5)


def test_weight_pack_and_unpack():
    lib, lb = L()
    g = torch.Generator().manual_seed(12)
    Co, Ci, k, Cp = 21, 129, 3, 136
    w = torch.randn(Co, Ci, k, k, generator=g)
    cmap = torch.full((Cp,), -1, dtype=torch.int32)
    cmap[:Ci] = torch.arange(Ci, dtype=torch.int32)
    cmap[130] = 5  # arbitrary remap of a pad slot
    Cop = 24
    wd = w.cuda()
    fwd = torch.zeros(Co, 9, Cp, dtype=torch.bfloat16, device="cuda")
    dg = torch.zeros(Cp, 9, Cop, dtype=torch.bfloat16, device="cuda")
    sc = torch.zeros(9, Cp, Cop, dtype=torch.bfloat16, device="cuda")
    cm = cmap.cuda()
    e = lib.PackEntry()
    e.src, e.dst_fwd, e.dst_dgrad, e.dst_scatter, e.cmap = P(wd), P(fwd), P(dg), P(sc), P(cm)
    e.Cout, e.Cin_ref, e.taps, e.Cin_pad, e.Cout_pad = Co, Ci, 9, Cp, Cop
    tab = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).cuda()
    ok(lb.crd_weight_pack(P(tab), 1, Cp * 9 * Cop, lib.stream()), "weight_pack")
    ref = torch.zeros(Co, 9, Cp)
    wr = w.reshape(Co, Ci, 9)
    for cp in range(Cp):
        if cmap[cp] >= 0:
            ref[:, :, cp] = wr[:, cmap[cp], :]
    assert torch.equal(fwd.float().cpu(), bf(ref))
    refd = torch.zeros(Cp, 9, Cop)
    refd[:, :, :Co] = ref.permute(2, 1, 0)
    assert torch.equal(dg.float().cpu(), bf(refd))
    assert torch.equal(sc.float().cpu(), bf(refd.permute(1, 0, 2)))
    # unpack: identity map back to the reference layout
    cmap2 = torch.full((Cp,), -1, dtype=torch.int32)
    cmap2[:Ci] = torch.arange(Ci, dtype=torch.int32)
    src = torch.randn(Co, 9, Cp, generator=g)
    dst = torch.ones(Co, Ci, 9)
    u = lib.UnpackEntry()
    sd, dd, c2 = src.cuda(), dst.cuda(), cmap2.cuda()
    u.src, u.dst, u.cmap, u.Cout, u.Cin_ref, u.taps, u.Cin_pad = P(sd), P(dd), P(c2), Co, Ci, 9, Cp
    tab2 = torch.frombuffer(bytearray(bytes(u)), dtype=torch.uint8).cuda()
    ok(lb.crd_wgrad_unpack(P(tab2), 1, Co * 9 * Cp, 1, lib.stream()), "wgrad_unpack")
    assert_close(dd.cpu(), 1 + src[:, :, :Ci].permute(0, 2, 1), "unpack", rel=1e-6, elem=1e-6)
    # ... from three copies of a fixed-point (crd_sum_t) accumulator
    src3 = torch.randn(3, Co, 9, Cp, generator=g)
    s3, dd3 = to_grad(src3).cuda(), torch.ones(Co, Ci, 9).cuda()
    u.src, u.dst, u.replicas, u.replica_stride, u.src_sum = P(s3), P(dd3), 3, Co * 9 * Cp, 1
    tab3 = torch.frombuffer(bytearray(bytes(u)), dtype=torch.uint8).cuda()
    ok(lb.crd_wgrad_unpack(P(tab3), 1, Co * 9 * Cp, 1, lib.stream()), "wgrad_unpack (sums)")
    assert_close(dd3.cpu(), 1 + src3.sum(0)[:, :, :Ci].permute(0, 2, 1), "unpack of crd_sum_t copies", rel=1e-6, elem=1e-6)


@pytest.mark.parametrize("Co,Ci,Cp,k,f32", [(64, 64, 64, 8, 0), (64, 7, 8, 7, 0), (1024, 128, 128, 1, 0), (128, 289, 304, 3, 0), (1, 640, 640, 3, 1),
                                            (160, 160, 160, 2, 0), (21, 128, 128, 3, 0)])
def test_weight_pack_shapes(Co, Ci, Cp, k, f32):
    """Every tap count / padding / layout the plan requests in ONE table launch: 8x8 and 7x7 patches, pointwise, a padded
    3x3 concat input, the depthwise fp32 [tap][channel] form; two entries per launch (table indexing)."""
    lib, lb = L()
    g = torch.Generator().manual_seed(Co + Ci)
    taps = k * k
    Cop = (Co + 7) // 8 * 8
    entries, keep, refs = [], [], []
    for rep in range(2):
        if f32:
            w = torch.randn(Ci, 1, k, k, generator=g)          # depthwise: [hid][1][3][3], one "output channel" in the table
            wr = w.reshape(1, Ci, taps)
        else:
            w = torch.randn(Co, Ci, k, k, generator=g)
            wr = w.reshape(Co, Ci, taps)
        ref = torch.zeros(wr.shape[0], taps, Cp)
        ref[:, :, :Ci] = wr.permute(0, 2, 1)
        wd = w.cuda()
        fwd = torch.full((wr.shape[0], taps, Cp), 7.0, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        dg = torch.full((Cp, taps, Cop), 7.0, dtype=torch.bfloat16, device="cuda")
        sc = torch.full((taps, Cp, Cop), 7.0, dtype=torch.bfloat16, device="cuda")
        e = lib.PackEntry()
        e.src, e.dst_fwd = P(wd), P(fwd)
        if not f32:
            e.dst_dgrad, e.dst_scatter = P(dg), P(sc)
        e.Cout, e.Cin_ref, e.taps, e.Cin_pad, e.Cout_pad, e.dst_f32 = wr.shape[0], Ci, taps, Cp, Cop, f32
        entries.append(e)
        keep.append((wd, fwd, dg, sc))
        refs.append(ref)
    tab = torch.frombuffer(bytearray(b"".join(bytes(e) for e in entries)), dtype=torch.uint8).cuda()
    ok(lb.crd_weight_pack(P(tab), 2, max(Co * taps * Cp, Cp * taps * Cop), lib.stream()), "weight_pack")
    torch.cuda.synchronize()
    for (wd, fwd, dg, sc), ref in zip(keep, refs):
        assert torch.equal(fwd.float().cpu(), ref if f32 else bf(ref))
        if not f32:
            refd = torch.zeros(Cp, taps, Cop)
            refd[:, :, :Co] = ref.permute(2, 1, 0)
            assert torch.equal(dg.float().cpu(), bf(refd))
            assert torch.equal(sc.float().cpu(), bf(refd.permute(1, 0, 2)))


def test_depth_metrics_match_reference_golden():
    """crd_test_metrics (device-side Trainer.test metrics) against the reference's golden values and the CPU oracle."""
    from camradepth_amd import synth
    from camradepth_amd.metrics import DepthMetrics
    from oracle import losses as ol
    from tests.util import load_npz
    g = load_npz("losses_metrics.npz")
    b = synth.make_batch(2, 24, 40, seed=3)
    pred = torch.from_numpy(g["pred"])
    dm = DepthMetrics()
    dm.update(pred.cuda(), b["gt_full"].cuda())
    pf = dm.per_frame()
    np.testing.assert_allclose([pf[0]["MAE"], pf[0]["RMSE"], pf[0]["REL"]], g["metrics"], rtol=2e-5)
    for f in range(2):
        m = ol.test_metrics(pred[f], b["gt_full"][f])
        np.testing.assert_allclose([pf[f]["MAE"], pf[f]["RMSE"], pf[f]["REL"]], [m["MAE"], m["RMSE"], m["REL"]], rtol=2e-5)
    # range-limited variant (<= 50 m) and a frame without valid ground truth
    dm50 = DepthMetrics(max_distance=50.0)
    gt0 = b["gt_full"].clone(); gt0[1] = 0
    dm50.update(pred.cuda(), gt0.cuda())
    m = ol.test_metrics(pred[0], gt0[0], max_distance=50.0)
    pf = dm50.per_frame()
    np.testing.assert_allclose([pf[0]["MAE"], pf[0]["RMSE"], pf[0]["REL"]], [m["MAE"], m["RMSE"], m["REL"]], rtol=2e-5)
    assert pf[1] is None and dm50.result() is not None


def test_batch_assembly_matches_dataloader_arithmetic():
    """crd_assemble_input / crd_gt_pyramid against the reference dataloader's own torch / numpy arithmetic
    (dataloader.py:213-222, 226-257, 300-318), restated here: ToTensor + Normalize, clip / scale, MaxPool2d-based minpool."""
    from camradepth_amd.batch import assemble_batch
    g = torch.Generator().manual_seed(21)
    B, H, W, max_depth = 2, 34, 50, 100.0
    img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    radar = torch.zeros(B, H, W, 3)
    m = torch.rand(B, H, W, generator=g) < 0.05
    radar[..., 0][m] = torch.rand(int(m.sum()), generator=g) * 130 - 5          # also below 0 and above max_depth
    radar[..., 1:][m] = torch.randn(int(m.sum()), 2, generator=g)
    rad_vel = torch.randn(B, H, W, generator=g) * m
    depth = torch.rand(B, H, W, generator=g) * 140 * (torch.rand(B, H, W, generator=g) < 0.2)
    out = assemble_batch(img.cuda(), radar.cuda(), rad_vel.cuda(), depth.cuda(), max_depth)
    # reference arithmetic
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    image = (img.permute(0, 3, 1, 2).float() / 255 - mean) / std
    rd = radar[..., 0].clamp(0, max_depth) / max_depth
    ref_x = torch.cat([image, rd.unsqueeze(1), radar[..., 1:].permute(0, 3, 1, 2), rad_vel.unsqueeze(1)], 1)
    assert_close(out["image"].cpu(), ref_x, "assembled input", rel=1e-6, elem=1e-5)
    gt = depth.clamp(0, max_depth)
    gt = torch.where(gt > 0, (max_depth - gt) * (1 / max_depth), gt).unsqueeze(1)

    def minpool(t):
        x = t.clone()
        x[t == 0] = 255
        x = -F.max_pool2d(-x, kernel_size=3, stride=2, padding=1)
        x[x == 255] = 0
        return x
    refs = [gt, minpool(gt), minpool(minpool(gt)), minpool(minpool(minpool(gt)))]
    for name, ref in zip(("gt_full", "gt_half", "gt_quarter", "gt_eighth"), refs):
        got = out[name].cpu()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert torch.equal(got, ref) or float((got - ref).abs().max()) < 1e-7, name


def test_nonfinite_partials_raise_the_sticky_flag():
    """A crd_sum_t accumulator cannot hold NaN / infinity: such a partial adds nothing and raises a sticky flag instead
    (include/camradepth_hip.h: crd_nonfinite_status), which TrainStep.losses() turns back into NaN (the fp32 atomics
    of round 2 propagated NaN, the integer sums of round 3 silently dropped it)."""
    from camradepth_amd import lib as L
    lib = L.load()
    L.nonfinite()                                        # clear
    n = 4096
    pred = torch.rand(n, device="cuda")
    tgt = torch.rand(n, device="cuda") + 0.1
    acc = zsum(4)
    L.check(lib.crd_masked_l1_fwd(pred.data_ptr(), tgt.data_ptr(), n, acc.data_ptr(), L.stream()))
    torch.cuda.synchronize()
    assert not L.nonfinite()
    pred[17] = float("nan")
    acc.zero_()
    L.check(lib.crd_masked_l1_fwd(pred.data_ptr(), tgt.data_ptr(), n, acc.data_ptr(), L.stream()))
    torch.cuda.synchronize()
    assert L.nonfinite(reset=False) and L.nonfinite() and not L.nonfinite()       # sticky until cleared
    pred[17] = float("inf")
    acc.zero_()
    L.check(lib.crd_masked_l1_fwd(pred.data_ptr(), tgt.data_ptr(), n, acc.data_ptr(), L.stream()))
    torch.cuda.synchronize()
    assert L.nonfinite()


def test_diffgradnorm_refuses_a_switched_active_set_without_touching_state():
    """the kernel's bias corrections use one step count per group.  A frozen for good after n steps and B unfrozen then is
    the silent mismatch (B would be corrected with step n + 1 instead of 1, diffGradNorm.py:66,76-77): the step is refused, and the
    refused call changes nothing -- not the `active` mask, not the step counts, not a parameter."""
    from camradepth_amd import lib as L
    from camradepth_amd.optim import diffGradNorm
    g = torch.Generator().manual_seed(3)
    a = torch.nn.Parameter(torch.randn(300, generator=g).cuda())
    b = torch.nn.Parameter(torch.randn(70, generator=g).cuda())
    opt = diffGradNorm([a, b], lr=1e-2)
    for _ in range(3):                                  # A trains alone (B has no gradient: skipped like the reference's `grad is None`)
        a.grad = torch.randn(300, generator=g).cuda()
        b.grad = None
        opt.step()
    st = opt._groups[0]
    assert st["step"] == 3 and opt.state[a]["step"] == 3 and opt.state[b]["step"] == 0
    act_before, host_before = st["active"].clone(), st.get("act_host")
    a0, b0 = a.detach().clone(), b.detach().clone()
    a.grad, b.grad = None, torch.randn(70, generator=g).cuda()      # the whole active set switches
    with pytest.raises(L.CrdError, match="step counts"):
        opt.step()
    assert st["step"] == 3 and opt.state[b]["step"] == 0
    assert torch.equal(st["active"], act_before) and st.get("act_host") == host_before
    assert torch.equal(a.detach(), a0) and torch.equal(b.detach(), b0)
    a.grad, b.grad = torch.randn(300, generator=g).cuda(), None     # ... and the run it interrupted can go on
    opt.step()
    assert st["step"] == 4 and opt.state[a]["step"] == 4


def test_diffgradnorm_never_reads_the_segment_of_a_tensor_without_gradient():
    """Round 6: with separately allocated tensors the optimizer uses the gradients in place -- g = the first active gradient's pointer minus
    its offset -- so a tensor WITHOUT a gradient (`p.grad is None`, diffGradNorm.py:54-55) has no storage behind its segment; k_dgn_norm read
    it anyway (past the end of another allocation: the full GPU suite took a memory access fault there once).  The norm kernel now skips
    such tensors like k_dgn_update always did: their workgroups' parts of ||g||^2 are written as zeros, whatever lies behind the segment."""
    from camradepth_amd.optim import diffGradNorm
    g = torch.Generator().manual_seed(11)
    a = torch.nn.Parameter(torch.randn(5000, generator=g).cuda())
    b = torch.nn.Parameter(torch.randn(9000, generator=g).cuda())
    opt = diffGradNorm([a, b], lr=1e-2)
    b0 = b.detach().clone()
    for _ in range(2):
        a.grad, b.grad = torch.randn(5000, generator=g).cuda(), None
        opt.step()
    torch.cuda.synchronize()
    st = opt._groups[0]
    blocks_of_b = (st["b2s"] == 1).nonzero().flatten()
    assert len(blocks_of_b) >= 2 and float(st["nsq"][blocks_of_b].abs().max()) == 0.0
    assert float(st["nsq"][(st["b2s"] == 0).nonzero().flatten()].sum()) > 0
    assert torch.equal(b.detach(), b0) and float(st["fac"][1]) == 1.0
