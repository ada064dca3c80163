"""CPU oracle for the CamRaDepth forward pass. TEST INFRASTRUCTURE ONLY.

This is a plain-PyTorch (CPU, fp32) restatement of the reference model written as pure
functions over a `state_dict` with the reference's key names. Only tests/, bench.py's
`cpu_baseline` leg and `__graft_entry__.smoke()` may import it; the product path
(camradepth_amd/) never does.

Parity status: PINNED. `tests/golden/make_golden.py` imports the real reference
(/root/reference, CPU) and stores its outputs; `tests/test_oracle_golden.py` checks this file
against those fixtures.  Third-party arithmetic (torch conv/group_norm/gelu/upsample_bicubic2d,
timm DropPath) is unpinned by the reference itself (it has no tests) and is pinned here only
through those fixtures (torch 2.10 CPU).

`quant="bf16"` rounds tensors to bf16 at the points where CUDA autocast does in the reference's
training loop (conv / matmul inputs, weights and outputs; src/main/runner.py:191) while
accumulating in fp32. That mode exists to compare against the bf16 HIP path at a tight tolerance.
"""
import math
import torch
import torch.nn.functional as F

GN_DIV = 16  # reference: src/utils/args.py:38 (groupnorm_divisor)


def _q(t, quant):
    if quant == "bf16" and t is not None:
        return t.to(torch.bfloat16).to(torch.float32)
    return t


def _conv2d(x, w, b, quant, **kw):
    """Conv2d under autocast: low-precision operands and result, fp32 accumulation."""
    return _q(F.conv2d(_q(x, quant), _q(w, quant), _q(b, quant), **kw), quant)


def _conv1d(x, w, b, quant):
    return _q(F.conv1d(_q(x, quant), _q(w, quant), _q(b, quant)), quant)


def _gn(x, sd, name, groups):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def patch_embed(sd, name, x, k, stride, quant=None):
    """OverlapPatchEmbed.forward (reference: src/models/simplified_attention.py:183-188)."""
    y = _conv2d(x, sd[name + ".proj.weight"], sd[name + ".proj.bias"], quant, stride=stride, padding=k // 2)
    H, W = y.shape[2], y.shape[3]
    y = _gn(y, sd, name + ".norm", y.shape[1] // GN_DIV)
    return y.flatten(2), H, W


def attention_maxpool(sd, name, x, H, W, heads, sr, quant=None, taps=None):
    """Attention_MaxPool.forward (reference: src/models/simplified_attention.py:90-109)."""
    B, C, N = x.shape
    d = C // heads
    scale = d ** -0.5
    q = _conv1d(x, sd[name + ".q.weight"], sd[name + ".q.bias"], quant)
    q = q.reshape(B, heads, d, N).permute(0, 1, 3, 2)
    if sr > 1:
        x_ = x.reshape(B, C, H, W)
        x_ = _conv2d(x_, sd[name + ".sr.weight"], sd[name + ".sr.bias"], quant, stride=sr).reshape(B, C, -1)
        x_ = _gn(x_, sd, name + ".norm", C // GN_DIV)
        k = _conv1d(x_, sd[name + ".k.weight"], sd[name + ".k.bias"], quant).reshape(B, heads, d, -1)
    else:
        k = _conv1d(x, sd[name + ".k.weight"], sd[name + ".k.bias"], quant).reshape(B, heads, d, -1)
    v = torch.mean(x, 2, True).repeat(1, 1, heads).transpose(-2, -1)      # [B, heads, C]
    attn = _q(_q(q @ k, quant) * scale, quant)                            # [B, heads, N, M]
    attn, idx = torch.max(attn, -1)                                       # [B, heads, N]
    if taps is not None:
        taps[name + ".rowmax"] = attn
        taps[name + ".argmax"] = idx            # [B, heads, N]: the key torch.max routes the gradient to
    out = _q(_q(attn.transpose(-2, -1), quant) @ _q(v, quant), quant)     # [B, N, C]
    out = out.transpose(-2, -1)
    return _conv1d(out, sd[name + ".proj.weight"], sd[name + ".proj.bias"], quant)


def mlp(sd, name, x, H, W, dim, quant=None):
    """Mlp.forward + DWConv.forward (reference: simplified_attention.py:34-43, 318-323)."""
    B = x.shape[0]
    h = _conv1d(x, sd[name + ".fc1.weight"], sd[name + ".fc1.bias"], quant)
    hid = h.shape[1]
    h = _gn(h, sd, name + ".norm1", hid // GN_DIV)
    h = _conv2d(h.reshape(B, hid, H, W), sd[name + ".dwconv.dwconv.weight"], sd[name + ".dwconv.dwconv.bias"],
                quant, padding=1, groups=hid).flatten(2)
    h = _gn(h, sd, name + ".norm2", dim // GN_DIV)        # groups from out_features (:24)
    h = F.gelu(h)
    return _conv1d(h, sd[name + ".fc2.weight"], sd[name + ".fc2.bias"], quant)


def block(sd, name, x, H, W, heads, sr, quant=None, drop_path=None, taps=None):
    """Block.forward (reference: simplified_attention.py:141-145). drop_path: [B] scaled mask or None."""
    C = x.shape[1]
    dp = (lambda t: t) if drop_path is None else (lambda t: t * drop_path.view(-1, 1, 1))
    if taps is not None:
        taps[name + ".in"] = x.detach()
    xn = _gn(x, sd, name + ".norm1", C // GN_DIV)
    x = x + dp(attention_maxpool(sd, name + ".attn", xn, H, W, heads, sr, quant, taps))
    x = x + dp(mlp(sd, name + ".mlp1", _gn(x, sd, name + ".norm2", C // GN_DIV), H, W, C, quant))
    if taps is not None:
        taps[name + ".out"] = x.detach()
    return x


def encoder(sd, x, cfg, quant=None, masks=None, taps=None):
    """SimplifiedTransformer.forward_features (reference: simplified_attention.py:265-306)."""
    B = x.shape[0]
    outs, bi = [], 0
    for s in range(4):
        k, stride = (7, 4) if s == 0 else (3, 2)
        x, H, W = patch_embed(sd, f"dest_encoder.patch_embed{s + 1}", x, k, stride, quant)
        for i in range(cfg.depths[s]):
            dp = None if masks is None else masks["drop_path"][bi]
            x = block(sd, f"dest_encoder.block{s + 1}.{i}", x, H, W, cfg.heads[s], cfg.reduction_ratio[s],
                      quant, dp, taps)
            bi += 1
        x = x.reshape(B, -1, H, W).contiguous()
        outs.append(x)
    return outs


E4M3_MAX = 448.0


def _e4m3(t):
    """Values representable in OCP fp8 e4m3 (gfx950's fp8): saturating, round to nearest even."""
    return t.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def _conv2d_fp8(x, w, x_scale, **kw):
    """The fp8 inference convolution of BASELINE.json config 5 (not in the reference, which runs fp16 autocast): the bf16
    activation tensor quantised per tensor (x / x_scale -> e4m3), the bf16 weights per output channel (scale = max|w| / 448,
    1 for an all-zero row), fp32 accumulation of the e4m3 products, result * x_scale * w_scale[c] rounded to bf16."""
    x_scale = torch.tensor(x_scale, dtype=torch.float32)
    xq = _e4m3(_q(x, "bf16") * (1.0 / x_scale))
    wb = _q(w, "bf16")
    am = wb.abs().amax(dim=(1, 2, 3))
    ws = torch.where(am > 0, am / E4M3_MAX, torch.ones_like(am))
    wq = _e4m3(wb * (1.0 / ws).view(-1, 1, 1, 1))
    return _q(F.conv2d(xq, wq, None, **kw) * (x_scale * ws).view(1, -1, 1, 1), "bf16")


class _Fp8ConvTrain(torch.autograd.Function):
    """The e4m3 convolution of config 5 WITH its e4m3 data gradient (camradepth_amd: crd_conv3x3_fp8 / crd_conv3x3_fp8_dgrad; no
    reference counterpart -- the reference trains under fp16 autocast, runner.py:191: this is a BUILDER-DEFINED emulation of what the
    HIP kernels do, not a restatement of the reference).  Forward: _conv2d_fp8.  Backward: dy rounded to bf16 (what the GroupNorm
    backward stores), quantised per tensor with g_scale, the bf16 weights quantised per INPUT channel with the scales `ws` (round 6: the
    three layers of a ShortResBlock share them -- kcat_weight_scales -- as the K-concatenated write-once data gradients of the HIP path
    quantise one matrix row per concat channel; ws None: this layer's own max over (cout, taps) / 448, 1 for an all-zero channel),
    dx = the transposed convolution of the de-quantised pair rounded to bf16; dw = the bf16 weight gradient of the bf16 activations and
    the bf16 dy."""

    @staticmethod
    def forward(ctx, x, w, x_scale, g_scale, pad, ws=None):
        ctx.save_for_backward(x, w)
        ctx.g_scale, ctx.pad, ctx.ws = g_scale, pad, ws
        return _conv2d_fp8(x, w, x_scale, padding=pad)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gb = _q(gy, "bf16")
        gs = torch.tensor(ctx.g_scale, dtype=torch.float32)
        gq = _e4m3(gb * (1.0 / gs)) * gs
        wb = _q(w, "bf16")
        if ctx.ws is None:
            am = wb.abs().amax(dim=(0, 2, 3))
            ws = torch.where(am > 0, am / E4M3_MAX, torch.ones_like(am))
        else:
            ws = ctx.ws[:w.shape[1]]
        wq = _e4m3(wb * (1.0 / ws).view(1, -1, 1, 1)) * ws.view(1, -1, 1, 1)
        dx = _q(F.conv_transpose2d(gq, wq, padding=ctx.pad), "bf16") if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(_q(x, "bf16"), w.shape, gb, padding=ctx.pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None, None


def kcat_weight_scales(ws_list):
    """Per concat channel n: max over the ShortResBlock layers that read channel n of max_{cout, tap} |bf16(w)[cout, n, tap]| / 448 (1 for a
    channel that is zero everywhere) -- the per-row scales of the HIP path's K-concatenated data-gradient matrices (crd_weight_quant_fp8 on
    [rows][9][K])."""
    n = max(w.shape[1] for w in ws_list)
    am = torch.zeros(n)
    for w in ws_list:
        a = _q(w.detach(), "bf16").abs().amax(dim=(0, 2, 3))
        am[:a.numel()] = torch.maximum(am[:a.numel()], a)
    return torch.where(am > 0, am / E4M3_MAX, torch.ones_like(am))


def conv_layer(sd, name, x, k, quant=None, fp8_scale=None, fp8_gscales=None, fp8_ws=None):
    """ConvLayer.forward: conv(no bias) -> GroupNorm(Cout/16) -> GELU (reference: src/utils/utils.py:210-228).
    fp8_gscales: {layer name: e4m3 scale of its dy} -- the layers listed there also take their DATA gradient in e4m3 (fp8_ws: the
    per-input-channel weight scales shared by the block's layers)."""
    w = sd[name + ".model.0.weight"]
    if fp8_scale is None:
        y = _conv2d(x, w, None, quant, padding=k // 2)
    elif torch.is_grad_enabled() and (x.requires_grad or w.requires_grad) and fp8_gscales and name in fp8_gscales:
        y = _Fp8ConvTrain.apply(x, w, fp8_scale, float(fp8_gscales[name]), k // 2, fp8_ws)
    elif torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        # fp8 forward inside a training step: the value of the fp8 convolution, the gradient of the bf16 one (what the HIP
        # path does: data and weight gradients are the bf16 kernels on the bf16 activations / weights)
        y16 = _conv2d(x, w, None, "bf16", padding=k // 2)
        with torch.no_grad():
            y8 = _conv2d_fp8(x, w, fp8_scale, padding=k // 2)
        y = y16 + (y8 - y16).detach()
    else:
        y = _conv2d_fp8(x, w, fp8_scale, padding=k // 2)
    return F.gelu(_gn(y, sd, name + ".model.1", w.shape[0] // GN_DIV))


def short_res_block(sd, name, x, quant=None, fp8_scale=None, fp8_gscales=None):
    """ShortResBlock.forward (reference: src/utils/utils.py:127-135)."""
    ws = None
    if fp8_gscales and all(f"{name}.layers.{li}" in fp8_gscales for li in range(3)):
        ws = kcat_weight_scales([sd[f"{name}.layers.{li}.model.0.weight"] for li in range(3)])
    for li in range(2):
        out = conv_layer(sd, f"{name}.layers.{li}", x, 3, quant, fp8_scale, fp8_gscales, ws)
        x = torch.cat((x, out), dim=1)
    return conv_layer(sd, f"{name}.layers.2", x, 3, quant, fp8_scale, fp8_gscales, ws)


def bicubic2x(x):
    """nn.Upsample(scale_factor=2, mode='bicubic') (reference: src/utils/utils.py:241)."""
    return F.interpolate(x, scale_factor=2, mode="bicubic")


def decoder_stage(sd, name, x, skip=None, quant=None, fp8_scale=None, fp8_gscales=None):
    """Decoder.forward (reference: src/utils/utils.py:249-257)."""
    x = bicubic2x(x)
    if skip is not None:
        x = torch.cat((x, skip), dim=1)
    return short_res_block(sd, name + ".conv", x, quant, fp8_scale, fp8_gscales)


def depth_activation(sd, name, x, quant=None):
    """Depth_Activation.forward (reference: src/utils/utils.py:285-289)."""
    z = _conv2d(x, sd[name + ".conv_1.weight"], sd[name + ".conv_1.bias"], quant, padding=1)
    a = torch.sigmoid(z)
    return _conv2d(a, sd[name + ".conv_2.weight"], sd[name + ".conv_2.bias"], quant, padding=1)


def seg_block(logits, num_classes):
    """Seg_Block.forward: argmax / num_classes, no gradient (reference: src/utils/utils.py:95-100)."""
    return torch.argmax(logits, dim=1, keepdim=True) / num_classes


def forward(sd, x, cfg, quant=None, masks=None, taps=None, fp8_scales=None, fp8_grad_scales=None):
    """CamRaDepth.forward (reference: src/models/CamRaDepth.py:99-176).

    masks: None (eval) or the dict of synth.make_masks (train mode with injected Dropout2d /
    DropPath masks). Returns the reference's nested output dict.
    fp8_scales: {"depth_upsample.3": s, "depth_upsample.4": s, "seg_upsample.0": s, ...} -- the ConvLayers of those decoder
    stages as fp8 convolutions (_conv2d_fp8) with these per-stage activation scales (what camradepth_amd's calibrate_fp8
    returns; a stage that is not listed stays as `quant` says).
    fp8_grad_scales: {"depth_upsample.4.conv.layers.2": s, ...} -- ConvLayers of fp8 stages whose data gradient is e4m3 as well
    (_Fp8ConvTrain), with the per-tensor scale of their dy (what the HIP plan's device-resident scales held for that step).
    """
    f8 = fp8_scales or {}
    g8 = fp8_grad_scales
    d2 = iter(masks["dropout2d"]) if masks is not None else None
    drop = (lambda t: t) if masks is None else (lambda t: t * next(d2).view(t.shape[0], t.shape[1], 1, 1))
    outs = encoder(sd, x, cfg, quant, masks, taps)
    if taps is not None:
        for i, o in enumerate(outs):
            taps[f"enc{i + 1}"] = o
    e1 = conv_layer(sd, "from_encoder_1", outs[3], 1, quant)
    e2 = conv_layer(sd, "from_encoder_2", outs[2], 1, quant)
    e3 = conv_layer(sd, "from_encoder_3", outs[1], 1, quant)
    e4 = conv_layer(sd, "from_encoder_4", outs[0], 1, quant)
    s1 = drop(decoder_stage(sd, "depth_upsample.0", e1, e2, quant))
    s2 = drop(decoder_stage(sd, "depth_upsample.1", s1, e3, quant))
    s3 = drop(decoder_stage(sd, "depth_upsample.2", s2, e4, quant))
    d3 = depth_activation(sd, "depth_activation_3", s3, quant)
    s3 = torch.cat([s3, d3], 1)
    s4 = drop(decoder_stage(sd, "depth_upsample.3", s3, None, quant, f8.get("depth_upsample.3"), g8))
    sup_map = unsup_map = seg_map = seg_feat = seg_final = None
    if cfg.supervised_seg or cfg.unsupervised_seg:
        seg_feat = drop(decoder_stage(sd, "seg_upsample.0", s3, None, quant, f8.get("seg_upsample.0"), g8))
    if cfg.supervised_seg:
        sup_map = seg_block(_conv2d(seg_feat, sd["seg_conv_stage_4.weight"], sd["seg_conv_stage_4.bias"], quant,
                                    padding=1), cfg.num_classes)
        seg_map = sup_map
    if cfg.unsupervised_seg:
        unsup_map = seg_block(_conv2d(seg_feat, sd["unsup_stage_4.weight"], sd["unsup_stage_4.bias"], quant,
                                      padding=1), 19)
        seg_map = unsup_map if sup_map is None else torch.cat([sup_map, unsup_map], 1)
    if cfg.supervised_seg:
        seg_feat = torch.cat((seg_feat, sup_map), dim=1)
    elif cfg.unsupervised_seg:
        seg_feat = torch.cat((seg_feat, unsup_map), dim=1)
    tmp = torch.cat((s4, seg_map), dim=1) if seg_map is not None else s4
    d4 = depth_activation(sd, "depth_activation_4", tmp, quant)
    s4 = torch.cat([s4, d4], 1)
    s5 = drop(decoder_stage(sd, "depth_upsample.4", s4, x, quant, f8.get("depth_upsample.4"), g8))
    if cfg.supervised_seg or cfg.unsupervised_seg:
        seg_feat = drop(decoder_stage(sd, "seg_upsample.1", seg_feat, x, quant, f8.get("seg_upsample.1"), g8))
    if cfg.supervised_seg:
        seg_final = _conv2d(seg_feat, sd["seg_conv_final.weight"], sd["seg_conv_final.bias"], quant, padding=1)
        sup_map = seg_block(seg_final, cfg.num_classes)
        seg_map = sup_map
    if cfg.unsupervised_seg:
        unsup_map = seg_block(_conv2d(seg_feat, sd["unsup_final.weight"], sd["unsup_final.bias"], quant, padding=1), 19)
        seg_map = unsup_map if sup_map is None else torch.cat([sup_map, unsup_map], 1)
    tmp = torch.cat((s5, seg_map), dim=1) if seg_map is not None else s5
    final = depth_activation(sd, "depth_activation_5", tmp, quant)
    if taps is not None:
        taps.update({"s1": s1, "s2": s2, "s3": s3, "s4": s4, "s5": s5})
    return {"depth": {"intermediate_depths": (None, None, d3, d4), "final_depth": final},
            "seg": {"final_seg": seg_final, "intermediate_seg": None, "unsup_map": unsup_map}}
