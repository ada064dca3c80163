"""GPU tests of the fp8 (OCP e4m3) inference path of the decoder's 3x3 ConvLayers (BASELINE.json config 5): quantisation
kernels bit-exact against torch's float8_e4m3fn conversion, the block-scaled-MFMA convolution against torch fp32 on the
de-quantised operands (so the only difference is fp32 accumulation order + the bf16 output rounding: rel-L2 < 4e-3, as for
the bf16 kernels), and the end-to-end model against the oracle's quant="fp8" mode."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_igemm import assert_close, bf, pack_w, to_pm
from tests.util import sval, to_stat, zsum

pytestmark = pytest.mark.gpu
E4M3 = torch.float8_e4m3fn
FP8G_SHALLOW_MED, FP8G_SHALLOW_DEC, FP8G_SHALLOW_WORST = 0.022, 0.006, 0.35        # 2 x measured: 0.0109 / 0.0028 / 0.17


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
    stats = zsum(B, Co // 16, 2)
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
    assert_close(sval(stats), torch.stack([gq.sum((2, 3)), (gq ** 2).sum((2, 3))], -1), "fp8 conv GroupNorm sums", rel=1e-3, elem=2e-3)
    # and the quantisation error itself against the un-quantised convolution: e4m3 has 3 mantissa bits
    full = F.conv2d(x, w, None, padding=1)
    rel = float((got - full).norm() / full.norm())
    print(f"fp8 vs bf16-operand convolution {case}: rel-L2 {rel:.4f}")
    assert rel < 0.06


def test_fused_fp8_producers_equal_quantised_bf16_outputs():
    """crd_gn_apply_fp8 / crd_bicubic2x_fp8 write exactly what crd_quant_fp8 makes of the bf16 kernels' outputs."""
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(5)
    B, H, W, Cc = 2, 12, 20, 96
    x = bf(torch.randn(B, H * W, Cc, generator=g) * 2).to(torch.bfloat16).cuda()
    v = x.float().reshape(B, -1, Cc // 16, 16)
    stats = to_stat(torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)).contiguous()
    gamma, beta = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.1 * torch.randn(Cc, generator=g)).cuda()
    scale = 0.013
    y16 = torch.zeros(B, H * W, Cc, dtype=torch.bfloat16, device="cuda")
    lib.check(L.crd_gn_apply(x.data_ptr(), 0, Cc, 0, B, H * W, Cc, stats.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), 1, None,
                             y16.data_ptr(), 0, Cc, 0, lib.stream()), "gn_apply")
    ref8 = torch.zeros(B, H * W, 128, dtype=torch.uint8, device="cuda")
    lib.check(L.crd_quant_fp8(y16.data_ptr(), B * H * W, Cc, 0, Cc, ref8.data_ptr(), 128, 16, scale, lib.stream()), "quant")
    got8 = torch.zeros_like(ref8)
    lib.check(L.crd_gn_apply_fp8(x.data_ptr(), 0, Cc, 0, B, H * W, Cc, stats.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), 1, None,
                                 got8.data_ptr(), 128, 16, scale, None, 0, 0, lib.stream()), "gn_apply_fp8")
    assert torch.equal(got8, ref8) and int(ref8[..., 16:16 + Cc].max()) > 0
    # with the bf16 output beside it (training plans keep it for the backward pass): both identical to the separate kernels
    got8.zero_()
    both16 = torch.zeros(B, H * W, Cc + 8, dtype=torch.bfloat16, device="cuda")
    lib.check(L.crd_gn_apply_fp8(x.data_ptr(), 0, Cc, 0, B, H * W, Cc, stats.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), 1, None,
                                 got8.data_ptr(), 128, 16, scale, both16.data_ptr(), Cc + 8, 8, lib.stream()), "gn_apply_fp8 dual")
    assert torch.equal(got8, ref8) and torch.equal(both16[..., 8:], y16) and float(both16[..., :8].float().abs().max()) == 0.0
    up16 = torch.zeros(B, 4 * H * W, Cc, dtype=torch.bfloat16, device="cuda")
    lib.check(L.crd_bicubic2x(x.data_ptr(), Cc, 0, B, H, W, Cc, up16.data_ptr(), Cc, 0, lib.stream()), "bicubic")
    ref8 = torch.zeros(B, 4 * H * W, 128, dtype=torch.uint8, device="cuda")
    lib.check(L.crd_quant_fp8(up16.data_ptr(), B * 4 * H * W, Cc, 0, Cc, ref8.data_ptr(), 128, 8, scale, lib.stream()), "quant")
    got8 = torch.zeros_like(ref8)
    lib.check(L.crd_bicubic2x_fp8(x.data_ptr(), Cc, 0, B, H, W, Cc, got8.data_ptr(), 128, 8, scale, None, 0, 0, lib.stream()), "bicubic_fp8")
    assert torch.equal(got8, ref8)
    got8.zero_()
    both16 = torch.zeros(B, 4 * H * W, Cc + 8, dtype=torch.bfloat16, device="cuda")
    lib.check(L.crd_bicubic2x_fp8(x.data_ptr(), Cc, 0, B, H, W, Cc, got8.data_ptr(), 128, 8, scale, both16.data_ptr(), Cc + 8, 0,
                                  lib.stream()), "bicubic_fp8 dual")
    assert torch.equal(got8, ref8) and torch.equal(both16[..., :Cc], up16)


def test_fp8_inference_model_matches_oracle_fp8_mode():
    """BASELINE config 5 (inference form): the ConvLayers of the two largest decoder stages in fp8 against the oracle's fp8 mode
    (same quantisation points and scales), against the bf16 path (what the quantisation costs), and the north-star RMSE."""
    from camradepth_amd import losses as hl, synth
    from camradepth_amd.config import ModelConfig
    from camradepth_amd.inference import InferenceGraph
    from oracle import losses as ol
    from oracle import model as om
    from tests.test_gpu_model import build, rel
    cfg = ModelConfig.variant("base")
    model = build(cfg)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = synth.make_batch(4, 256, 416, seed=1234)
    x = batch["image"].cuda()
    with torch.no_grad():
        out16 = model(x)["depth"]["final_depth"].clone()
    scales = model.calibrate_fp8(x)
    assert set(scales) == {"depth_upsample.3", "depth_upsample.4"} and all(s > 0 for s in scales.values())
    with torch.no_grad():
        res = model(x)
        out8, half8 = res["depth"]["final_depth"].clone(), res["depth"]["intermediate_depths"][3].clone()
    plan = model._plans[model._plan_key(x)]
    assert sum(op.name == "crd_conv3x3_fp8" for op in plan.fwd) == 6          # the native fp8 kernels ran, not a bf16 fallback
    assert sum(op.name == "crd_gn_apply_fp8" for op in plan.fwd) == 4
    # the graph-replayed inference path takes the same route
    ig = InferenceGraph(model, 4, 256, 416)
    assert sum(op.name == "crd_conv3x3_fp8" for op in ig.plan.fwd) == 6
    outg = ig.run(x)["depth"]["final_depth"]
    # (run-to-run: the encoder's GroupNorm sums are fp32 atomics, and an activation that lands on the other side of an e4m3
    # rounding boundary moves by 6 % -- measured 1.1e-2 between two runs of the same plan)
    r_graph = rel(outg, out8)
    assert r_graph < 4e-2
    o8 = om.forward(sd, batch["image"], cfg, quant="bf16", fp8_scales=scales)
    o32 = om.forward(sd, batch["image"], cfg)
    r_par = rel(out8, o8["depth"]["final_depth"])
    r_half = rel(half8, o8["depth"]["intermediate_depths"][3])
    r_q = rel(out8, out16)
    rmse8 = float(torch.sqrt(hl.MaskedMSELoss()(out8, batch["gt_full"].cuda())))
    rmse32 = float(torch.sqrt(ol.masked_mse(o32["depth"]["final_depth"], batch["gt_full"])))
    print(f"fp8 inference: graph replay vs eager {r_graph:.4f}, vs oracle fp8 mode {r_par:.4f} (half-res {r_half:.4f}), vs bf16 path {r_q:.4f}, "
          f"RMSE {rmse8:.6f} vs fp32 oracle {rmse32:.6f} (gap {abs(rmse8 - rmse32):.2e}), scales {scales}")
    assert r_par < 2e-2 and r_half < 2e-2          # same arithmetic: bf16-level agreement (measured, see profiles/r02_gpu_tests.log)
    assert r_q < 5e-2                               # e4m3 operands: 3 mantissa bits, averaged over K >= 1296 products
    assert abs(rmse8 - rmse32) < 1e-3               # the north-star accuracy gate holds in fp8 as well
    model.calibrate_fp8(None)
    with torch.no_grad():
        again = model(x)["depth"]["final_depth"]
    assert rel(again, out16) < 1.5e-2               # switched off: back on the bf16 plan (bf16 run-to-run: ~4e-3)


def test_fp8_forward_in_training_matches_oracle_ste():
    """fp8 forward convolutions inside a training step (calibrate_fp8(train=True)): forward value of the e4m3 convolution,
    gradients of the bf16 one (straight-through), against the oracle doing the same.  Shallow encoder (depths 1,1,1,1) so that the
    decoder -- where the fp8 layers are -- dominates and the chaotic arg-max routing of a deep encoder does not."""
    import dataclasses
    from camradepth_amd import losses as hl, synth
    from camradepth_amd.config import ModelConfig
    from camradepth_amd.params import param_specs
    from oracle import losses as ol
    from oracle import model as om
    from tests.test_gpu_model import build, rel
    cfg = dataclasses.replace(ModelConfig.variant("base"), depths=(1, 1, 1, 1))
    sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
    model = build(cfg, sd, train=True)
    batch = synth.make_batch(4, 256, 416, seed=77)
    masks = synth.make_masks(cfg, 4, seed=4321)
    x = batch["image"].cuda()
    scales = model.calibrate_fp8(x, train=True)
    assert model.training                                    # calibration restores the mode
    out = model(x, masks=masks)
    plan = model._plans[model._plan_key(x)]
    assert plan.training and sum(op.name == "crd_conv3x3_fp8" for op in plan.fwd) == 6
    loss, _ = hl.total_loss(out, {k: v.cuda() for k, v in batch.items()}, False)
    loss.backward()
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = om.forward(sdo, batch["image"], cfg, quant="bf16", masks=masks, fp8_scales=scales)
    lo, _ = ol.total_loss(o, batch, False)
    lo.backward()
    r_out = rel(out["depth"]["final_depth"], o["depth"]["final_depth"])
    named = dict(model.named_parameters())
    errs = []
    for n, _ in param_specs(cfg):
        go, g = sdo[n].grad, named[n].grad
        if go is None:
            continue
        errs.append((rel(g, go), n))
    dec = [e for e, n in errs if n.startswith("depth_upsample.3") or n.startswith("depth_upsample.4")]
    med, worst_dec = float(np.median([e for e, _ in errs])), max(dec)
    loss, lo = loss.detach(), lo.detach()
    print(f"fp8 forward in training: loss {float(loss):.6f} vs oracle {float(lo):.6f}, final depth rel-L2 {r_out:.4f}, "
          f"gradient rel-L2 median {med:.4f}, worst of the fp8 stages' parameters {worst_dec:.4f}")
    assert abs(float(loss) - float(lo)) <= 5e-3 * abs(float(lo))
    assert r_out < 3e-2
    assert med < 0.03 and worst_dec < 0.02          # measured 9.1e-3 / 3.6e-3 (profiles/r02_gpu_tests.log)
    model.calibrate_fp8(None)


def test_fp8_route_selection_by_grid_and_variant():
    """Which stages take the fp8 route is decided per plan: a stage needs >= 192 tiles of 16 x 32 pixels (a single 416 x 800 frame
    qualifies with its full-resolution stages only), and the segmentation branch's stages are calibrated on their own concat
    buffers.  The supervised-seg model agrees with the oracle's fp8 mode restricted to the stages the plan chose."""
    from camradepth_amd import synth
    from camradepth_amd.config import ModelConfig
    from camradepth_amd.inference import InferenceGraph
    from oracle import model as om
    from tests.test_gpu_model import build, rel
    cfg = ModelConfig.variant("supervised_seg")
    model = build(cfg)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = synth.make_batch(1, 416, 800, seed=5)
    x = batch["image"].cuda()
    with torch.no_grad():
        ref = model(x)
    d16, s16 = ref["depth"]["final_depth"].clone(), ref["seg"]["final_seg"].clone()
    scales = model.calibrate_fp8(x)
    assert set(scales) == {"depth_upsample.3", "depth_upsample.4", "seg_upsample.0", "seg_upsample.1"}
    ig = InferenceGraph(model, 1, 416, 800)
    n8 = sum(op.name == "crd_conv3x3_fp8" for op in ig.plan.fwd)
    assert n8 == 6, n8       # 13 x 25 tiles at full resolution (both branches), 13 x 13 = 169 < 192 at half resolution
    out = ig.run(x)
    used = {k: v for k, v in scales.items() if k in ("depth_upsample.4", "seg_upsample.1")}
    o8 = om.forward(sd, batch["image"], cfg, quant="bf16", fp8_scales=used)
    r_d, r_s = rel(out["depth"]["final_depth"], d16), rel(out["seg"]["final_seg"], s16)
    p_d, p_s = rel(out["depth"]["final_depth"], o8["depth"]["final_depth"]), rel(out["seg"]["final_seg"], o8["seg"]["final_seg"])
    print(f"fp8 (full-resolution stages) 1x416x800 supervised_seg: depth vs bf16 {r_d:.4f} / vs oracle fp8 {p_d:.4f}, "
          f"seg logits vs bf16 {r_s:.4f} / vs oracle fp8 {p_s:.4f}, scales {scales}")
    # (seg logits at the reference initialisation are small and zero-mean: their bf16-vs-oracle floor is ~2e-2 already;
    # measured here 0.012 / 0.009 depth, 0.059 / 0.046 seg -- bounds = 2x)
    assert r_d < 3e-2 and p_d < 2e-2 and r_s < 0.12 and p_s < 0.09
    model.calibrate_fp8(None)


# ---------------------------------------------------------------------------------------------------------------------------
# Round 5: e4m3 DATA GRADIENTS of the same ConvLayers (config 5 as a training step)
# ---------------------------------------------------------------------------------------------------------------------------
def pack_w_dgrad(w, cout_pad=None):
    """[Cout,Cin,3,3] fp32 -> the packed data-gradient weights bf16 [Cin][9][Cout_pad] (crd_pack_entry.dst_dgrad: tap = ky*3+kx, un-mirrored)."""
    Co, Ci = w.shape[:2]
    cout_pad = cout_pad or Co
    p = torch.zeros(Ci, 9, cout_pad, dtype=torch.bfloat16)
    p[:, :, :Co] = w.permute(1, 2, 3, 0).reshape(Ci, 9, Co).to(torch.bfloat16)
    return p.cuda()


DGRAD_CASES = [
    # B, Cout (channels of dy = K), H, W, N (channels of dx to produce), accumulate
    (8, 128, 96, 128, 304, 0),      # 256 columns on 128-wide tiles + 48 on a 64-wide one
    (8, 64, 90, 120, 240, 1),       # one 64-channel chunk, two 128-wide tiles (the second 112 wide), read-modify-write, ragged borders
    (8, 96, 94, 128, 144, 1),       # 64 + 32-channel chunks, 128 + 16 columns
    (2, 128, 64, 96, 136, 0),       # few tiles
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv3x3_fp8_dgrad_matches_dequantised_reference(case):
    B, Co, H, W, N, acc = case
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(sum(case))
    C16 = (Co + 15) // 16 * 16
    dy = bf(torch.randn(B, Co, H, W, generator=g) * 3e-4 * torch.rand(B, Co, 1, 1, generator=g))       # gradient-sized values
    w = bf(torch.randn(Co, N, 3, 3, generator=g) / (N * 9) ** 0.5)
    w[:, 5] = 0                                                                                         # a pad channel of the concat buffer
    dypm = to_pm(dy, ld=C16, coff=0)
    scale = torch.tensor([float(dy.abs().max()) / 448.0], device="cuda")                                # device-resident scale
    dy8 = torch.zeros(B, H, W, C16, dtype=torch.uint8, device="cuda")
    lib.check(L.crd_quant_fp8_dev(dypm.data_ptr(), B * H * W, C16, 0, Co, dy8.data_ptr(), C16, 0, scale.data_ptr(), lib.stream()), "quant_dev")
    ref8 = torch.zeros_like(dy8)
    lib.check(L.crd_quant_fp8(dypm.data_ptr(), B * H * W, C16, 0, Co, ref8.data_ptr(), C16, 0, float(scale), lib.stream()), "quant")
    assert torch.equal(dy8, ref8)                                                                       # same quantiser, scale from memory
    wd = pack_w_dgrad(w)                                                                                # bf16 [N][9][Co]
    w8 = torch.zeros(N, 9, C16, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(N, device="cuda")
    lib.check(L.crd_weight_quant_fp8(wd.data_ptr(), N, 9, Co, C16, w8.data_ptr(), ws.data_ptr(), lib.stream()), "wquant")
    assert float(ws[5]) == 1.0
    ld = N + 16
    old = bf(torch.randn(B, H, W, ld, generator=g) * 1e-4).to(torch.bfloat16).cuda()
    y = old.clone()
    d = lib.ConvDesc()
    d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = dy8.data_ptr(), C16, 0, B, H, W, C16
    d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = w8.data_ptr(), N, 3, 3, 1, 1, H, W
    d.gather_mode, d.y, d.y_ld, d.y_coff, d.accumulate = 1, y.data_ptr(), ld, 8, acc
    lib.check(L.crd_conv3x3_fp8_dgrad(C.byref(d), ws.data_ptr(), scale.data_ptr(), lib.stream()), "conv3x3_fp8_dgrad")
    torch.cuda.synchronize()
    dq = dequant(dy8[..., :Co]).permute(0, 3, 1, 2) * float(scale)
    wq = dequant(w8[:, :, :Co]).reshape(N, 3, 3, Co).permute(3, 0, 1, 2) * ws.cpu().view(1, N, 1, 1)     # back to [Co][N][3][3]
    ref = F.conv_transpose2d(dq, wq, padding=1)
    got = y[..., 8:8 + N].float().cpu().permute(0, 3, 1, 2)
    if acc:
        ref = bf(bf(ref) + old[..., 8:8 + N].float().cpu().permute(0, 3, 1, 2))
    assert_close(got, ref, f"fp8 dgrad {case}")
    assert torch.equal(y[..., :8], old[..., :8]) and torch.equal(y[..., 8 + N:], old[..., 8 + N:])         # neighbours of the slice untouched
    assert float(got[:, 5].abs().max()) == (float(old[..., 8 + 5].float().abs().max()) if acc else 0.0)
    full = F.conv_transpose2d(dy, w, padding=1)
    if acc:
        full = full + old[..., 8:8 + N].float().cpu().permute(0, 3, 1, 2)
    rel = float((got - full).norm() / full.norm())
    print(f"fp8 data gradient vs the bf16-operand one {case}: rel-L2 {rel:.4f}")
    assert rel < 0.06


def test_gn_bwd_apply_fp8_equals_the_bf16_kernel_plus_quantisation():
    """crd_gn_bwd_apply_fp8 writes the same bf16 dx as crd_gn_bwd_apply, its e4m3 copy = crd_quant_fp8_dev of that tensor with the
    same device scale, the amax slots hold max |bf16 dx|; crd_fp8_scale_update turns them into amax / 448 and zeroes them."""
    lib = _lib()
    L = lib.load()
    g = torch.Generator().manual_seed(11)
    B, P, Cc = 3, 50 * 37, 96
    x = bf(torch.randn(B, P, Cc, generator=g) * 2).to(torch.bfloat16).cuda()
    dy = bf(torch.randn(B, P, Cc + 8, generator=g) * 1e-3).to(torch.bfloat16).cuda()
    v = x.float().reshape(B, P, Cc // 16, 16)
    stats = to_stat(torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)).contiguous()
    gamma, beta = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.1 * torch.randn(Cc, generator=g)).cuda()
    common = [x.data_ptr(), 0, Cc, 0, dy.data_ptr(), 0, Cc + 8, 8, B, P, Cc, stats.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), 1, None]
    res = []
    for fp8 in (False, True):
        r = torch.zeros(B * Cc * 2 + B * (Cc // 16) * 2, dtype=torch.int64, device="cuda")
        lib.check(L.crd_gn_bwd_reduce(*common, r.data_ptr(), None, 0, lib.stream()), "reduce")
        dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        dx = torch.zeros(B, P, Cc, dtype=torch.bfloat16, device="cuda")
        if not fp8:
            lib.check(L.crd_gn_bwd_apply(*common, r.data_ptr(), dg.data_ptr(), db.data_ptr(), dx.data_ptr(), 0, Cc, 0, 0, None, 0, None, lib.stream()), "apply")
            res.append((dx, dg, db))
            continue
        scale = torch.tensor([3.1e-6], device="cuda")
        slots = torch.zeros(64, dtype=torch.int32, device="cuda")
        dx8 = torch.zeros(B, P, 112, dtype=torch.uint8, device="cuda")
        lib.check(L.crd_gn_bwd_apply_fp8(*common, r.data_ptr(), dg.data_ptr(), db.data_ptr(), dx.data_ptr(), Cc, 0, dx8.data_ptr(), 112, 16,
                                         scale.data_ptr(), slots.data_ptr(), lib.stream()), "apply_fp8")
        res.append((dx, dg, db))
        ref8 = torch.zeros_like(dx8)
        lib.check(L.crd_quant_fp8_dev(dx.data_ptr(), B * P, Cc, 0, Cc, ref8.data_ptr(), 112, 16, scale.data_ptr(), lib.stream()), "quant_dev")
        torch.cuda.synchronize()
        assert torch.equal(dx8, ref8) and int(dx8[..., 16:16 + Cc].max()) > 0
        amax = float(slots.view(torch.float32).max())
        assert amax == float(dx.float().abs().max()) and int((slots != 0).sum()) > 1                  # spread over several slots
        scales = torch.tensor([7.0, 9.0], device="cuda")
        two = torch.stack([slots, torch.zeros_like(slots)]).contiguous()
        lib.check(L.crd_fp8_scale_update(two.data_ptr(), scales.data_ptr(), 2, 1.0, 0, lib.stream()), "scale_update")
        torch.cuda.synchronize()
        assert abs(float(scales[0]) - amax / 448.0) <= 1e-7 * amax and float(scales[1]) == 9.0          # nothing recorded: the scale is kept
        assert int(two.abs().max()) == 0
    (dx0, dg0, db0), (dx1, dg1, db1) = res
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)


def _fp8_grad_step_vs_oracle(depths, B, H, W, seed=77, reference_init=False):
    import dataclasses
    from camradepth_amd import losses as hl, synth
    from camradepth_amd.config import ModelConfig
    from camradepth_amd.params import param_specs
    from oracle import losses as ol
    from oracle import model as om
    from tests.test_gpu_model import build, rel
    cfg = dataclasses.replace(ModelConfig.variant("base"), depths=depths)
    if reference_init:       # the reference's initialisation (as tests/test_gpu_train.py::_train_step_vs_oracle): the deliberately
        model = build(cfg, None, train=True)      # ill-conditioned fill_state_dict weights are chaotic through 34 blocks
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    else:
        sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
        model = build(cfg, sd, train=True)
    batch = synth.make_batch(B, H, W, seed=seed)
    masks = synth.make_masks(cfg, B, seed=4321)
    x = batch["image"].cuda()
    scales = model.calibrate_fp8(x, train=True, grads=True)
    out = model(x, masks=masks)
    plan = model._plans[model._plan_key(x)]
    n8 = sum(op.name == "crd_conv3x3_fp8_dgrad" for op in plan.bwd)
    # stages with >= 192 tiles of 16 x 32 pixels take the fp8 route (B = 2: the full-resolution stage only); round 6: per stage ALL three
    # data gradients are e4m3 -- the write-once launches over the K-concatenated gradient buffer -- with ONE dy scale per stage
    stages = 2 if B * (H // 32) * (W // 64) >= 192 else 1
    assert plan.training and plan.fp8_jit and n8 == 3 * stages and len(plan.fp8_grad_layers) == stages      # the native e4m3 data gradients ran
    loss, _ = hl.total_loss(out, {k: v.cuda() for k, v in batch.items()}, False)
    loss.backward()
    torch.cuda.synchronize()
    # just-in-time: THIS step's max |dy| over the stage's three slices / 448 (the launches behind layers 2 and 1 ran with the running max of
    # the slices produced until then: a difference of one e4m3 quantisation grid for those two launches, inside the tolerances below)
    gsc = {f"{n}.conv.layers.{li}": float(plan.g8_scales[i]) for i, n in enumerate(plan.fp8_grad_layers) for li in range(3)}
    assert all(0 < s < 1 for s in gsc.values()) and int(plan.g8_amax.abs().max()) == 0
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = om.forward(sdo, batch["image"], cfg, quant="bf16", masks=masks, fp8_scales=scales, fp8_grad_scales=gsc)
    lo, _ = ol.total_loss(o, batch, False)
    lo.backward()
    sdb = {k: v.clone().requires_grad_(True) for k, v in sd.items()}                                   # the bf16-backward (straight-through) oracle
    ob = om.forward(sdb, batch["image"], cfg, quant="bf16", masks=masks, fp8_scales=scales)
    lb, _ = ol.total_loss(ob, batch, False)
    lb.backward()
    named = dict(model.named_parameters())
    errs, q_errs, ratios = [], [], []
    for n, _ in param_specs(cfg):
        go, g, gb = sdo[n].grad, named[n].grad, sdb[n].grad
        if go is None:
            continue
        errs.append((rel(g, go), n))
        q_errs.append(rel(go, gb))
        ratios.append(float(g.double().norm() / (go.double().norm() + 1e-30)))
    model.calibrate_fp8(None)
    return float(loss.detach()), float(lo.detach()), errs, q_errs, np.array(ratios), gsc


def test_fp8_data_gradients_in_training_match_oracle_fp8_mode_shallow():
    """Config 5 as a training iteration (calibrate_fp8(train=True, grads=True)): e4m3 forward AND e4m3 data gradients in decoder stages
    3-4 against the oracle doing the same (oracle/model.py::_Fp8ConvTrain) with the scales the HIP plan used.  Shallow encoder: the
    decoder -- where the fp8 layers are -- dominates, per-ELEMENT gradients are comparable."""
    loss, lo, errs, q_errs, ratios, gsc = _fp8_grad_step_vs_oracle((1, 1, 1, 1), 4, 256, 416)
    med, worst = float(np.median([e for e, _ in errs])), max(errs)
    up = [e for e, n in errs if not n.startswith(("depth_upsample.3", "depth_upsample.4", "depth_activation"))]   # everything UPSTREAM of the e4m3 dgrads
    print(f"fp8 forward + data gradients: loss {loss:.6f} vs oracle {lo:.6f}; gradient rel-L2 vs the oracle's fp8 mode: median {med:.4f}, "
          f"worst {worst}, upstream-of-fp8 median {float(np.median(up)):.4f}; what e4m3 dy costs (oracle fp8-grad vs oracle bf16-grad): "
          f"median {float(np.median(q_errs)):.4f}; scales {gsc}")
    dec = max(e for e, n in errs if n.startswith(("depth_upsample.3", "depth_upsample.4")))
    print(f"worst parameter of the fp8 stages {dec:.4f}")
    assert abs(loss - lo) <= 5e-3 * abs(lo)
    # measured (round 5): median 0.0109 (the fp8-forward-only test above: 0.0091), the fp8 stages' own parameters -- see the print --,
    # worst overall 0.17 on dest_encoder.block2.0.attn.sr.weight (an arg-max-routed gradient two stages upstream of the e4m3 layers:
    # one flipped key moves it; 0.12-0.2 in the bf16-only shallow tests as well); bounds = 2x
    assert med < FP8G_SHALLOW_MED and dec < FP8G_SHALLOW_DEC and worst[0] < FP8G_SHALLOW_WORST, (med, dec, worst)
    assert 0.97 < float(np.median(ratios)) < 1.03


def test_fp8_data_gradients_train_step_at_full_depth_vs_oracle_fp8_mode():
    """VERDICT r4 item 1 'done' criterion: train-step parity vs the oracle's fp8 mode at full depth, per-parameter NORM ratios as in
    tests/test_gpu_train.py::_train_step_vs_oracle (per-element gradients of the 34-block model are chaotic in bf16)."""
    loss, lo, errs, q_errs, r, gsc = _fp8_grad_step_vs_oracle((3, 10, 16, 5), 2, 256, 416, seed=2024, reference_init=True)
    print(f"fp8 train step at full depth: loss {loss:.6f} / {lo:.6f}; per-parameter norm ratio median {np.median(r):.3f}, "
          f"5-95 % {np.percentile(r, 5):.3f}-{np.percentile(r, 95):.3f}")
    assert abs(loss - lo) < 3e-3 * abs(lo)
    assert 0.9 < float(np.median(r)) < 1.1
    assert float(np.percentile(r, 5)) > 0.6 and float(np.percentile(r, 95)) < 1.6


def test_fp8_graph_step_with_delayed_scaling_equals_the_just_in_time_step_on_the_calibration_batch():
    """TrainStep captures the delayed-scaling variant after one just-in-time calibration iteration on the batch in its buffers: the
    first graph step quantises with exactly the scales that iteration left (equal scales, equal losses; gradients equal to e4m3 rounding:
    see the comment at the assertion); a second step on the same batch uses step 1's amax (delayed)."""
    from camradepth_amd import synth
    from camradepth_amd.config import ModelConfig
    from camradepth_amd.trainer import TrainStep
    from tests.test_gpu_model import build, rel
    from tests.test_gpu_train import fix_masks
    cfg = ModelConfig.variant("base")
    B = 8
    batch = {k: v.cuda() for k, v in synth.make_batch(B, 256, 416, seed=1234).items()}
    masks = synth.make_masks(cfg, B, seed=4321)
    m0 = build(cfg)
    sd = {k: v.detach().cpu().clone() for k, v in m0.state_dict().items()}
    del m0
    res = {}
    for use_graph in (True, False):
        m = build(cfg, sd, train=True)
        m.calibrate_fp8(batch["image"], train=True, grads=True)
        ts = TrainStep(m, B, 256, 416, lr=6e-5, use_graph=use_graph)
        fix_masks(ts, masks)
        ts.set_batch(batch)
        ts.step()
        torch.cuda.synchronize()
        assert len(ts.plan.fp8_grad_layers) == 2 and ts.plan.fp8_jit == (not use_graph)
        res[use_graph] = (ts.losses(), m.flat_grad.clone(), ts.plan.g8_scales.clone())
        if use_graph:
            ts.step()
            torch.cuda.synchronize()
            assert float(ts.plan.g8_scales[:2].min()) > 0                     # delayed update at the head of the second backward
        del ts, m
        torch.cuda.empty_cache()
    (lg, gg, sg), (le, ge, se) = res[True], res[False]
    r = rel(gg, ge)
    print(f"fp8 graph step (delayed scaling, calibrated on this batch) vs eager just-in-time step: loss {lg['loss']:.6f} / {le['loss']:.6f}, "
          f"gradient rel-L2 {r:.3e}, scales {sg[:2].tolist()} / {se[:2].tolist()}")
    assert lg == le
    assert torch.equal(sg[:2], se[:2])
    # Round 6: one scale per STAGE.  The graph step quantises all three slices with the calibrated stage scale; the just-in-time step ran the
    # launches behind layers 2 and 1 with the running max of the slices produced until then -- the same values on another e4m3 grid for two
    # of the three launches of a stage: agreement to e4m3 rounding, no longer bit for bit (round 5, per-layer scales: 3.5e-8).
    assert r < 2e-2


# ---- round 6: the e4m3 attention scores (BASELINE.json configs[4] "fp8 MFMA attention"): an experiment with a verdict, not a plan path
def _quant_heads(lib, L, x, heads, d, scale):
    """bf16 [B, P, heads * d] -> e4m3 [B, P, heads, 64] (head dimension zero-padded), per-tensor scale: crd_quant_fp8 per head."""
    B, P, C = x.shape
    y = torch.zeros(B, P, heads, 64, dtype=torch.uint8, device="cuda")
    for h in range(heads):
        lib.check(L.crd_quant_fp8(x.data_ptr(), B * P, C, h * d, d, y.data_ptr(), heads * 64, h * 64, float(scale), lib.stream()), "crd_quant_fp8")
    return y


@pytest.mark.parametrize("B,N,M,heads,d", [(2, 6656, 104, 1, 64), (2, 1664, 104, 2, 64), (2, 416, 104, 4, 40), (2, 104, 104, 8, 32), (1, 77, 45, 4, 40)])
def test_attention_scores_e4m3_variant_and_argmax_flip_rate(B, N, M, heads, d):
    """crd_attn_scores_fp8 against torch on the de-quantised operands (same rounding points as the bf16 kernel behind the product), and
    what e4m3 operands do to the ARG-MAX the max-pool attention routes its gradient through (simplified_attention.py:104-106): the
    flip rate against the bf16 kernel on the same q, k.  The numbers printed here are the ones DESIGN.md quotes for the decision."""
    from camradepth_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(N + 7 * heads)
    C = heads * d
    q = (torch.randn(B, N, C, generator=g) * 0.8).to(torch.bfloat16).cuda()
    k = (torch.randn(B, M, C, generator=g) * 0.8).to(torch.bfloat16).cuda()
    scale = d ** -0.5
    qs, ks = float(q.float().abs().max()) / 448.0, float(k.float().abs().max()) / 448.0
    q8, k8 = _quant_heads(lib, L, q, heads, d, qs), _quant_heads(lib, L, k, heads, d, ks)
    S8, i8 = torch.zeros(B, N, device="cuda"), torch.zeros(B, N, heads, dtype=torch.int16, device="cuda")
    lib.check(L.crd_attn_scores_fp8(q8.data_ptr(), k8.data_ptr(), B, N, M, heads, qs * ks, scale, S8.data_ptr(), i8.data_ptr(), lib.stream()), "scores fp8")
    S16, i16 = torch.zeros(B, N, device="cuda"), torch.zeros(B, N, heads, dtype=torch.int16, device="cuda")
    lib.check(L.crd_attn_scores(q.data_ptr(), k.data_ptr(), B, N, M, heads, d, scale, S16.data_ptr(), i16.data_ptr(), lib.stream()), "scores bf16")
    torch.cuda.synchronize()
    # torch on the de-quantised operands
    deq = lambda t8, sc: t8.view(torch.float8_e4m3fn).float()[..., :d] * sc                   # [B, P, heads, d]
    qd, kd = deq(q8, 1.0).cpu(), deq(k8, 1.0).cpu()
    prod = torch.einsum("bnhd,bmhd->bhnm", qd, kd) * (qs * ks)
    sc_ = (prod.to(torch.bfloat16).float() * scale).to(torch.bfloat16).float()
    best, arg = sc_.max(-1)                                                                    # [B, heads, N]
    S_ref = best.sum(1)
    same = (arg.permute(0, 2, 1) == i8.cpu().long()).float().mean().item()
    err = float((S8.cpu() - S_ref).abs().max() / (S_ref.abs().max() + 1e-9))
    flips = (i8 != i16).float().mean().item()
    dS = float((S8 - S16).norm() / (S16.norm() + 1e-9))
    print(f"e4m3 scores B{B} N{N} M{M} heads{heads} d{d}: arg-max equal to torch's on the de-quantised operands {100 * same:.2f} %, max |S - S_ref| "
          f"{err:.2e}; against the bf16 kernel: {100 * flips:.1f} % of the (query, head) arg-maxes flip, S rel-L2 {dS:.3e}")
    assert same > 0.99 and err < 2e-2          # (ties between equal bf16 scores may resolve differently; the values agree)
    assert flips < 0.6
