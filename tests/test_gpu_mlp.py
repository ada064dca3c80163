"""GPU parity of the fused Mlp of an encoder Block (crd_mlp_fwd + crd_mlp_reduce, csrc/mlp_fused.hip) against torch fp32 with
the reference's rounding points: Block.norm2 -> fc1 -> Mlp.norm1 -> depthwise 3x3 -> Mlp.norm2 -> GELU -> fc2 -> DropPath ->
residual (src/models/simplified_attention.py:34-43,141-145).  Tolerance as for the unfused kernels: one bf16 rounding per
stored tensor (|err| <= 1e-2 max|ref|, rel-L2 <= 4e-3); the GroupNorm sums must equal the sums of the stored tensors."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_igemm import assert_close, bf
from tests.util import sval, to_stat, zsum

pytestmark = pytest.mark.gpu


def slab_sums(x):          # [B, N, C] -> [B, C/16, 2]
    B, N, Cc = x.shape
    v = x.double().reshape(B, N, Cc // 16, 16)
    return torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1).float()


@pytest.mark.parametrize("B,H,W,Cs,hid", [(2, 16, 26, 160, 640), (3, 8, 13, 256, 1024), (2, 5, 7, 32, 128), (1, 7, 9, 64, 256)])
def test_fused_mlp_forward(B, H, W, Cs, hid):
    from camradepth_amd import lib
    L = lib.load()
    N = H * W
    slabs = L.crd_mlp_fused_supported(H, W, Cs, hid)
    assert slabs == hid // 64
    assert L.crd_mlp_fused_supported(64, 104, 64, 512) == 0 and L.crd_mlp_fused_supported(32, 52, 128, 1024) == 0
    g = torch.Generator().manual_seed(B * 1000 + Cs)
    x1 = torch.randn(B, N, Cs, generator=g) * 1.7 + 0.3
    w1 = bf(torch.randn(hid, Cs, generator=g) / Cs ** 0.5)
    b1 = 0.2 * torch.randn(hid, generator=g)
    w2 = bf(torch.randn(Cs, hid, generator=g) / hid ** 0.5)
    b2 = 0.2 * torch.randn(Cs, generator=g)
    wd = torch.randn(hid, 1, 3, 3, generator=g) / 3
    bd = 0.1 * torch.randn(hid, generator=g)
    gam = [1 + 0.1 * torch.randn(n, generator=g) for n in (Cs, hid, hid)]
    bet = [0.1 * torch.randn(n, generator=g) for n in (Cs, hid, hid)]
    dp = torch.tensor([1.0 / 0.9, 0.0, 1.0][:B])
    # torch reference ([B, C, N] / [B, C, H, W] layouts)
    xc = x1.permute(0, 2, 1)
    xn = bf(F.group_norm(xc, Cs // 16, gam[0], bet[0], 1e-5))
    h1 = bf(F.conv1d(xn, w1.unsqueeze(-1), b1))
    h1n = bf(F.group_norm(h1, hid // 16, gam[1], bet[1], 1e-5))
    h2 = bf(F.conv2d(h1n.reshape(B, hid, H, W), wd, bd, padding=1, groups=hid)).reshape(B, hid, N)
    h3 = bf(F.gelu(F.group_norm(h2, hid // 64, gam[2], bet[2], 1e-5)))
    o = bf(F.conv1d(h3, w2.unsqueeze(-1), b2))
    x2 = xc + dp.view(B, 1, 1) * o
    # device side
    dev = "cuda"
    x1d = x1.to(dev)
    st2 = to_stat(slab_sums(x1)).to(dev)
    t = {k: v.to(dev).contiguous() for k, v in dict(g0=gam[0], b0=bet[0], g1=gam[1], b1n=bet[1], g2=gam[2], b2n=bet[2], b1=b1, b2=b2, bd=bd,
                                                  w1=w1.to(torch.bfloat16), w2=w2.to(torch.bfloat16),
                                                  w9=wd.reshape(hid, 9).t().contiguous(), dp=dp).items()}
    outs = {k: torch.full((B, N, hid), float("nan"), dtype=torch.bfloat16, device=dev) for k in ("h1", "h2", "h3")}
    xn_o = torch.full((B, N, Cs), float("nan"), dtype=torch.bfloat16, device=dev)
    sth1, sth2 = zsum(B, hid // 16, 2), zsum(B, hid // 16, 2)
    part = torch.full((slabs, B, N, Cs), float("nan"), device=dev)
    d = lib.MlpDesc()
    d.x1, d.x1_stats, d.norm_gamma, d.norm_beta = x1d.data_ptr(), st2.data_ptr(), t["g0"].data_ptr(), t["b0"].data_ptr()
    d.w_fc1, d.b_fc1, d.norm1_gamma, d.norm1_beta = t["w1"].data_ptr(), t["b1"].data_ptr(), t["g1"].data_ptr(), t["b1n"].data_ptr()
    d.w9, d.b_dw, d.norm2_gamma, d.norm2_beta = t["w9"].data_ptr(), t["bd"].data_ptr(), t["g2"].data_ptr(), t["b2n"].data_ptr()
    d.w_fc2 = t["w2"].data_ptr()
    d.xn, d.h1, d.h2, d.h3 = xn_o.data_ptr(), outs["h1"].data_ptr(), outs["h2"].data_ptr(), outs["h3"].data_ptr()
    d.h1_stats, d.h2_stats, d.fc2_partials = sth1.data_ptr(), sth2.data_ptr(), part.data_ptr()
    d.B, d.H, d.W, d.C, d.hidden = B, H, W, Cs, hid
    lib.check(L.crd_mlp_fwd(C.byref(d), lib.stream()), "crd_mlp_fwd")
    x2d = torch.zeros(B, N, Cs, device=dev)
    nst, nch = zsum(B, Cs // 16, 2), zsum(B, Cs, 2)
    lib.check(L.crd_mlp_reduce(part.data_ptr(), slabs, x1d.data_ptr(), t["b2"].data_ptr(), t["dp"].data_ptr(), B, N, Cs, x2d.data_ptr(),
                               nst.data_ptr(), nch.data_ptr(), lib.stream()), "crd_mlp_reduce")
    torch.cuda.synchronize()
    assert not bool(torch.isnan(part).any())
    assert_close(xn_o.float().cpu(), xn.permute(0, 2, 1), "Block.norm2(x1)")
    assert_close(outs["h1"].float().cpu(), h1.permute(0, 2, 1), "h1 = fc1")
    assert_close(outs["h2"].float().cpu(), h2.permute(0, 2, 1), "h2 = depthwise", rel=8e-3, elem=2e-2)
    assert_close(outs["h3"].float().cpu(), h3.permute(0, 2, 1), "h3 = GELU(norm2)", rel=8e-3, elem=3e-2)
    assert_close(x2d.cpu(), x2.permute(0, 2, 1), "x2", rel=4e-3, elem=2e-2)
    # the GroupNorm sums are those of the STORED (rounded) tensors
    assert_close(sval(sth1), slab_sums(outs["h1"].float().cpu()), "h1 sums", rel=1e-5, elem=1e-5)
    assert_close(sval(sth2), slab_sums(outs["h2"].float().cpu()), "h2 sums", rel=1e-5, elem=1e-5)
    assert_close(sval(nst), slab_sums(x2d.cpu()), "x2 g16 sums", rel=1e-5, elem=1e-5)
    xd = x2d.double().cpu()
    assert_close(sval(nch), torch.stack([xd.sum(1), (xd * xd).sum(1)], -1).float(), "x2 channel sums", rel=1e-5, elem=1e-5)
    # bit-reproducible: a second run gives the same bits everywhere
    part2, x2b = torch.zeros_like(part), torch.zeros_like(x2d)
    d.fc2_partials = part2.data_ptr()
    lib.check(L.crd_mlp_fwd(C.byref(d), lib.stream()), "crd_mlp_fwd (second run)")
    lib.check(L.crd_mlp_reduce(part2.data_ptr(), slabs, x1d.data_ptr(), t["b2"].data_ptr(), t["dp"].data_ptr(), B, N, Cs, x2b.data_ptr(),
                               None, None, lib.stream()), "crd_mlp_reduce (no sums)")
    torch.cuda.synchronize()
    assert torch.equal(part2, part) and torch.equal(x2b, x2d)
