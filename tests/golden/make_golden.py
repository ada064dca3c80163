#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by importing the REAL reference on CPU.

Runs only in the build container (needs /root/reference). The reference is imported read-only
with in-process stand-ins for absent third-party packages (easydict, timm.models.layers,
torchinfo, cv2 -- SURVEY.md Appendix C); nothing of the reference is copied: fixtures hold
inputs' seeds and OUTPUT tensors only.

    python tests/golden/make_golden.py            # all variants (one subprocess each: the
                                                  # reference reads a global `args` at import)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
VARIANTS = ["base", "supervised_seg", "unsupervised_seg", "sup_unsup_seg"]


def install_shims():
    import torch
    import torch.nn as nn

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    m = types.ModuleType("easydict")
    m.EasyDict = EasyDict
    sys.modules["easydict"] = m

    class DropPath(nn.Module):  # timm 0.6.12 semantics + mask injection
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep, self.injected = drop_prob, scale_by_keep, None

        def forward(self, x):
            if self.injected is not None:
                return x * self.injected.view((x.shape[0],) + (1,) * (x.ndim - 1))
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                r.div_(keep)
            return x * r

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tl.DropPath, tl.to_2tuple, tl.trunc_normal_ = DropPath, to_2tuple, torch.nn.init.trunc_normal_
    timm.models, tm.layers = tm, tl
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    ti = types.ModuleType("torchinfo")
    ti.summary = lambda *a, **k: None
    sys.modules["torchinfo"] = ti
    sys.modules["cv2"] = types.ModuleType("cv2")
    return DropPath


def t2n(t):
    return t.detach().cpu().numpy()


def sub(t, maxn=8192):
    """Exact fp32 strided subsample of a large tensor: flatten()[::stride] with stride = ceil(numel/maxn).
    Tests recompute the same view (tests/util.py:subsample)."""
    a = t2n(t).reshape(-1)
    stride = max(1, -(-a.size // maxn))
    return a[::stride].copy()


def run_variant(variant):
    import numpy as np
    import torch
    import torch.nn as nn
    torch.manual_seed(0)
    torch.set_num_threads(8)
    DropPath = install_shims()
    tmp = tempfile.mkdtemp()
    sys.argv = ["x", "--split", f"{REF}/src/data/new_split.npy", "--model", variant, "--output_dir", tmp]
    sys.path.insert(0, f"{REF}/src")
    sys.path.insert(0, REPO)
    from models.CamRaDepth import CamRaDepth  # noqa: E402  (reference)
    from models.diffGradNorm import diffGradNorm  # noqa: E402
    from utils.loss_funcs import MaskedSmoothL1Loss, MaskedFocalLoss, MaskedMSELoss  # noqa: E402
    from camradepth_amd.config import ModelConfig
    from camradepth_amd import synth

    cfg = ModelConfig.variant(variant)
    model = CamRaDepth(input_channels=7)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    order = [[n, list(p.shape)] for n, p in model.named_parameters()]
    with open(os.path.join(HERE, f"param_order_{variant}.json"), "w") as f:
        json.dump({"params": order, "state_dict_keys": list(model.state_dict().keys()),
                   "num_params": int(sum(p.numel() for p in model.parameters()))}, f)

    def set_masks(masks):
        blocks = [b for s in range(1, 5) for b in getattr(model.dest_encoder, f"block{s}")]
        for i, b in enumerate(blocks):
            if isinstance(b.drop_path, DropPath):
                b.drop_path.injected = None if masks is None else masks["drop_path"][i]
        if masks is None:
            model.dropout = nn.Dropout2d(0.2)
            model.dropout.train(model.training)
        else:
            it = iter(masks["dropout2d"])

            class Inject(nn.Module):
                def forward(self, x):
                    return x * next(it).view(x.shape[0], x.shape[1], 1, 1)
            model.dropout = Inject()

    def pack_out(out, prefix, store):
        store[prefix + "final_depth"] = t2n(out["depth"]["final_depth"])
        store[prefix + "depth_quarter"] = t2n(out["depth"]["intermediate_depths"][2])
        store[prefix + "depth_half"] = t2n(out["depth"]["intermediate_depths"][3])
        if out["seg"]["final_seg"] is not None:
            store[prefix + "final_seg"] = sub(out["seg"]["final_seg"], 32768)
        if out["seg"]["unsup_map"] is not None:
            store[prefix + "unsup_map"] = t2n(out["seg"]["unsup_map"])

    # ---- G2: eval forward at 1x7x64x96 with encoder maps -------------------------------------
    store = {}
    model.eval()
    batch = synth.make_batch(1, 64, 96, seed=1234)
    with torch.no_grad():
        enc, _ = model.dest_encoder(batch["image"])
        out = model(batch["image"])
    for i, e in enumerate(enc):
        store[f"eval_enc{i + 1}"] = t2n(e)
    pack_out(out, "eval_", store)

    # ---- G2t/G4: train-mode (injected masks) forward + loss + grads at 2x7x64x96 --------------
    crit_d, crit_s, crit_m = MaskedSmoothL1Loss(), MaskedFocalLoss(), MaskedMSELoss()

    def loss_of(out, batch):
        seg = out["seg"]["final_seg"]
        inter = out["depth"]["intermediate_depths"]
        l_seg = (crit_s(seg, batch["seg"]) if seg is not None else 0) * (1 if cfg.supervised_seg else 0)
        l4 = crit_d(inter[-1].squeeze(1), batch["gt_half"].squeeze(1))
        l3 = crit_d(inter[-2].squeeze(1), batch["gt_quarter"].squeeze(1))
        lf = crit_d(out["depth"]["final_depth"], batch["gt_full"])
        w = [1, 1, 1, 0.2, 0.2]
        loss = (w[0] * lf + w[1] * l4 + w[2] * l3 + w[3] * l_seg + w[4] * 0) / sum(w)
        rmse = torch.sqrt(crit_m(out["depth"]["final_depth"], batch["gt_full"]))
        return loss, lf, l4, l3, l_seg, rmse

    grad_names = ["dest_encoder.patch_embed1.proj.weight", "dest_encoder.block1.0.attn.q.weight",
                  "dest_encoder.block1.0.attn.sr.weight", "dest_encoder.block1.0.attn.norm.weight",
                  "dest_encoder.block1.2.mlp1.dwconv.dwconv.weight", "dest_encoder.block2.3.attn.k.bias",
                  "dest_encoder.block2.9.mlp1.norm2.weight", "dest_encoder.block3.7.attn.proj.weight",
                  "dest_encoder.block3.15.mlp1.fc2.weight", "dest_encoder.block4.4.attn.k.weight",
                  "dest_encoder.block4.0.norm1.bias", "dest_encoder.patch_embed3.norm.weight",
                  "from_encoder_2.model.0.weight", "depth_upsample.0.conv.layers.1.model.0.weight",
                  "depth_upsample.2.conv.layers.2.model.1.weight", "depth_upsample.4.conv.layers.0.model.0.weight",
                  "depth_activation_3.conv_1.weight", "depth_activation_5.conv_2.weight",
                  "depth_activation_5.conv_2.bias", "depth_activation_4.conv_1.bias"]
    if cfg.supervised_seg:
        grad_names += ["seg_conv_final.weight", "seg_upsample.0.conv.layers.0.model.0.weight", "seg_conv_stage_4.bias"]
    if cfg.unsupervised_seg:
        grad_names += ["seg_upsample.1.conv.layers.2.model.0.weight"]
    named = dict(model.named_parameters())
    batch2 = synth.make_batch(2, 64, 96, seed=77)
    for mode in ("evalgrad", "train"):
        model.train()
        masks = synth.make_masks(cfg, 2, seed=4321) if mode == "train" else None
        if mode == "evalgrad":
            model.eval()
        set_masks(masks)
        model.zero_grad(set_to_none=True)
        x = batch2["image"].clone().requires_grad_(True)
        out = model(x)
        loss, lf, l4, l3, l_seg, rmse = loss_of(out, batch2)
        loss.backward()
        pack_out(out, mode + "_", store)
        store[mode + "_loss"] = np.array([float(loss), float(lf), float(l4), float(l3), float(l_seg), float(rmse)],
                                         dtype=np.float64)
        store[mode + "_grad_input"] = sub(x.grad, 32768)
        for n in grad_names:
            g = named[n].grad
            store[f"{mode}_grad:{n}"] = sub(g) if g is not None else np.zeros(0, np.float32)
        # per-parameter gradient norms for every tensor (cheap whole-model coverage)
        store[mode + "_gradnorms"] = np.array([float(p.grad.norm()) if p.grad is not None else -1.0
                                               for _, p in model.named_parameters()], dtype=np.float64)
    set_masks(None)
    np.savez_compressed(os.path.join(HERE, f"forward64x96_{variant}.npz"), **store)

    # ---- G3: full-size eval forward 1x7x256x416 ----------------------------------------------
    if variant in ("base", "supervised_seg"):
        model.eval()
        big = synth.make_batch(1, 256, 416, seed=1234)
        with torch.no_grad():
            enc, _ = model.dest_encoder(big["image"])
            out = model(big["image"])
        st = {}
        for i, e in enumerate(enc):
            st[f"enc{i + 1}_stats"] = np.array([float(e.mean()), float(e.norm())], dtype=np.float64)
        st["final_depth"] = t2n(out["depth"]["final_depth"])
        st["depth_half"] = t2n(out["depth"]["intermediate_depths"][3]).astype(np.float16)
        st["depth_quarter"] = t2n(out["depth"]["intermediate_depths"][2])
        if out["seg"]["final_seg"] is not None:
            st["seg_argmax"] = t2n(out["seg"]["final_seg"].argmax(1)).astype(np.uint8)
            st["seg_stats"] = np.array([float(out["seg"]["final_seg"].mean()), float(out["seg"]["final_seg"].norm())])
        loss, lf, l4, l3, l_seg, rmse = loss_of(out, big)
        st["loss"] = np.array([float(loss), float(lf), float(l4), float(l3), float(l_seg), float(rmse)])
        np.savez_compressed(os.path.join(HERE, f"forward256x416_{variant}.npz"), **st)

    if variant != "base":
        return
    # ---- G1: leaf modules at tiny sizes (base process only) ----------------------------------
    from models.simplified_attention import Block, Attention_MaxPool, Mlp, OverlapPatchEmbed
    from utils.utils import ConvLayer, ShortResBlock, Decoder, Depth_Activation, Seg_Block
    rs = np.random.RandomState(5)
    leaf = {}

    def filled(mod, seed):
        mod.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()}, seed))
        return mod.eval()

    def rnd(*shape):
        return torch.from_numpy(rs.standard_normal(size=shape).astype(np.float32))

    for tag, (dim, heads, ratio, sr, H, W) in {"blk_sr2": (32, 2, 4, 2, 8, 12), "blk_sr1": (48, 3, 2, 1, 4, 6),
                                                "blk_sr4": (32, 1, 8, 4, 8, 8)}.items():
        blk = filled(Block(dim=dim, num_heads=heads, mlp_ratio=ratio, qkv_bias=True, sr_ratio=sr), 11)
        x = rnd(2, dim, H * W)
        with torch.no_grad():
            leaf[f"{tag}_x"] = t2n(x)
            leaf[f"{tag}_y"] = t2n(blk(x, H, W))
            xn = blk.norm1(x)
            leaf[f"{tag}_attn_y"] = t2n(blk.attn(xn, H, W))
            leaf[f"{tag}_mlp_y"] = t2n(blk.mlp1(xn, H, W))
    pe = filled(OverlapPatchEmbed(img_size=(32, 48), patch_size=7, stride=4, in_chans=7, embed_dim=32), 12)
    x = rnd(2, 7, 32, 48)
    with torch.no_grad():
        leaf["pe7_x"], leaf["pe7_y"] = t2n(x), t2n(pe(x)[0])
    pe = filled(OverlapPatchEmbed(img_size=(8, 12), patch_size=3, stride=2, in_chans=16, embed_dim=32), 13)
    x = rnd(2, 16, 8, 12)
    with torch.no_grad():
        leaf["pe3_x"], leaf["pe3_y"] = t2n(x), t2n(pe(x)[0])
    cl = filled(ConvLayer(24, 32, 3, padding=1), 14)
    x = rnd(2, 24, 9, 11)
    with torch.no_grad():
        leaf["convlayer_x"], leaf["convlayer_y"] = t2n(x), t2n(cl(x))
    srb = filled(ShortResBlock(24, 128), 15)
    with torch.no_grad():
        leaf["srb_y"] = t2n(srb(x))
    dec = filled(Decoder(16, 128, skip_size=8, dense=True, block=ShortResBlock), 16)
    x, skip = rnd(2, 16, 5, 7), rnd(2, 8, 10, 14)
    with torch.no_grad():
        leaf["dec_x"], leaf["dec_skip"], leaf["dec_y"] = t2n(x), t2n(skip), t2n(dec(x, skip))
        leaf["bicubic_y"] = t2n(dec.upsample(x))
    da = filled(Depth_Activation(24, 1), 17)
    x = rnd(2, 24, 9, 11)
    with torch.no_grad():
        leaf["da_x"] = t2n(x)
        leaf["da_y"] = t2n(da(x))
        leaf["segblock_y"] = t2n(Seg_Block(21)(x[:, :21]))
    np.savez_compressed(os.path.join(HERE, "leaf_modules.npz"), **leaf)

    # ---- G5: diffGradNorm, 40 steps, OneCycleLR-driven lr and beta1 --------------------------
    rs = np.random.RandomState(9)
    ps = [nn.Parameter(torch.from_numpy(rs.standard_normal(size=s).astype(np.float32))) for s in [(7,), (4, 5), (3, 2, 3, 3)]]
    opt = diffGradNorm(ps, lr=6e-5)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=6e-5, total_steps=41, div_factor=2, pct_start=0.15)
    og = {"p0_init": [t2n(p).copy() for p in ps]}
    scales = [1.0] * 6 + [0.03] * 4 + [1.0] * 6 + [0.02] * 3 + [5.0] * 5 + [0.01] * 6 + [1.0] * 10
    hp, grads, traj, egn = [], [], [], []
    for it in range(40):
        gs = [torch.from_numpy((scales[it] * rs.standard_normal(size=tuple(p.shape))).astype(np.float32)) for p in ps]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        hp.append([opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0], opt.param_groups[0]["betas"][1]])
        opt.step()
        sched.step()
        grads.append(gs)
        traj.append([t2n(p).copy() for p in ps])
        egn.append([float(opt.state[p]["exp_grad_norm"]) for p in ps])
    dg = {"hp": np.array(hp, dtype=np.float64), "exp_grad_norm": np.array(egn, dtype=np.float64)}
    for j in range(3):
        dg[f"p{j}_init"] = og["p0_init"][j]
        dg[f"p{j}_grads"] = np.stack([t2n(g[j]) for g in grads])
        dg[f"p{j}_traj"] = np.stack([t[j] for t in traj])
        dg[f"p{j}_exp_avg"] = t2n(opt.state[ps[j]]["exp_avg"])
        dg[f"p{j}_exp_avg_sq"] = t2n(opt.state[ps[j]]["exp_avg_sq"])
        dg[f"p{j}_previous_grad"] = t2n(opt.state[ps[j]]["previous_grad"])
    np.savez_compressed(os.path.join(HERE, "diffgradnorm_40steps.npz"), **dg)

    # ---- G6: losses / test() metrics on a fixed pair -----------------------------------------
    rs = np.random.RandomState(21)
    pred = torch.from_numpy(rs.uniform(-0.2, 1.2, size=(2, 1, 24, 40)).astype(np.float32))
    b = synth.make_batch(2, 24, 40, seed=3)
    logits = torch.from_numpy(rs.standard_normal(size=(2, 21, 24, 40)).astype(np.float32))
    lm = {"pred": t2n(pred), "logits": t2n(logits),
          "smooth_l1": np.float64(crit_d(pred, b["gt_full"])), "mse": np.float64(crit_m(pred, b["gt_full"])),
          "focal": np.float64(crit_s(logits, b["seg"]))}
    # test() metric formulae (reference: src/main/runner.py:443-465), evaluated with the reference's own ops
    p, g = torch.clip(pred[0].squeeze(), 0, 1) * 100, b["gt_full"][0].squeeze() * 100
    g = g.clone()
    g[g > 100] = 0
    idx = torch.where(g > 0)
    err = p[idx] - g[idx]
    lm["metrics"] = np.array([float(nn.L1Loss()(p[idx], g[idx])), float(torch.sqrt(nn.MSELoss()(p[idx], g[idx]))),
                              float(torch.sum(torch.abs(err) / g[idx]) / len(err))], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "losses_metrics.npz"), **lm)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default=None)
    a = ap.parse_args()
    if a.variant is None:
        for v in VARIANTS:
            print("== generating fixtures for", v, flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--variant", v])
    else:
        run_variant(a.variant)
