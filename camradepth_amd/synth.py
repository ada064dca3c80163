"""Seeded synthetic inputs and weights (no dataset or checkpoint exists in any container).

Follows SURVEY.md section 8(d): the tensor contract of the reference dataloader
(src/data/dataloader.py:202-333) and the radar statistics of lib/fuse_radar.py:185-303.
All draws come from numpy's frozen MT19937 stream (np.random.RandomState) so the same
arrays are produced in the build container (golden fixtures) and on the GPU box.
"""
import numpy as np
import torch


def fill_state_dict(shapes, seed=0):
    """Deterministic weights for a {name: shape} mapping, independent of any model code.

    Keys are visited in sorted order. Conv weights ~ N(0,1)/sqrt(fan_in), GroupNorm weights
    ~ 1 + 0.1 N(0,1), biases ~ 0.1 N(0,1). Returns {name: float32 torch tensor}.
    """
    rs = np.random.RandomState(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(int(s) for s in shapes[name])
        z = rs.standard_normal(size=shape).astype(np.float32)
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = z / np.float32(np.sqrt(fan_in))
        elif name.endswith(".weight"):
            v = 1.0 + 0.1 * z
        else:
            v = 0.1 * z
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def min_pool_ignore_zero(t):
    """3x3 stride-2 min-pool that ignores zeros (reference: src/data/dataloader.py:213-222)."""
    x = t.clone()
    x[t == 0] = 255
    x = -torch.nn.functional.max_pool2d(-x, kernel_size=3, stride=2, padding=1)
    x[x == 255] = 0
    return x


def make_batch(B, H, W, seed=1234, with_seg=True, channels=7):
    """One synthetic batch in the reference dataloader's format.

    Returns dict: image [B,7,H,W] f32; gt_full [B,1,H,W]; gt_half [B,1,H/2,W/2];
    gt_quarter [B,1,H/4,W/4]; seg [B,H,W] int64 (labels 0..20, 255 = ignore).
    """
    rs = np.random.RandomState(seed)
    x = np.zeros((B, 7, H, W), dtype=np.float32)
    x[:, 0:3] = rs.standard_normal(size=(B, 3, H, W)).astype(np.float32)
    f = 0.8 * W
    for b in range(B):
        k = int(rs.randint(80, 401))
        k = min(k, H * W // 4)
        rows = rs.randint(0, H, size=k)
        cols = rs.randint(0, W, size=k)
        x[b, 3, rows, cols] = rs.uniform(0.02, 1.0, size=k).astype(np.float32)
        x[b, 4, rows, cols] = ((cols - W / 2) / f + rs.normal(0, 0.01, size=k)).astype(np.float32)
        x[b, 5, rows, cols] = ((rows - H / 2) / f + rs.normal(0, 0.01, size=k)).astype(np.float32)
        x[b, 6, rows, cols] = (rs.uniform(size=k) < 0.15).astype(np.float32)
    gt = np.zeros((B, 1, H, W), dtype=np.float32)
    valid = rs.uniform(size=(B, 1, H, W)) < 0.2
    vals = rs.uniform(0.01, 0.99, size=(B, 1, H, W)).astype(np.float32)
    gt[valid] = vals[valid]
    gt_full = torch.from_numpy(gt)
    gt_half = min_pool_ignore_zero(gt_full)
    gt_quarter = min_pool_ignore_zero(gt_half)
    out = {"image": torch.from_numpy(x[:, :channels].copy()), "gt_full": gt_full,
           "gt_half": gt_half, "gt_quarter": gt_quarter}
    if with_seg:
        bh, bw = (H + 15) // 16, (W + 15) // 16
        blocks = rs.randint(0, 21, size=(B, bh, bw)).astype(np.int64)
        seg = np.repeat(np.repeat(blocks, 16, axis=1), 16, axis=2)[:, :H, :W].copy()
        ign = rs.uniform(size=(B, H, W)) < 0.05
        seg[ign] = 255
        out["seg"] = torch.from_numpy(seg)
    return out


def make_masks(cfg, B, seed=4321, dropout_p=0.2):
    """Injectable train-mode masks, already scaled by 1/keep.

    drop_path[i]: [B] for encoder block i in forward order (timm DropPath per-sample Bernoulli,
    rates from ModelConfig.drop_path_rates; block 0 has rate 0 = Identity).
    dropout2d[j]: [B,128] for the j-th Dropout2d application in CamRaDepth.dest_decoder
    (src/models/CamRaDepth.py:115-152; all applications are on 128-channel maps).
    """
    rs = np.random.RandomState(seed)
    dp = []
    for r in cfg.drop_path_rates:
        keep = 1.0 - r
        m = (rs.uniform(size=B) < keep).astype(np.float32) / np.float32(keep)
        dp.append(torch.from_numpy(m))
    n_drop = 5 + (2 if (cfg.supervised_seg or cfg.unsupervised_seg) else 0)
    keep = 1.0 - dropout_p
    d2 = [torch.from_numpy((rs.uniform(size=(B, 128)) < keep).astype(np.float32) / np.float32(keep))
          for _ in range(n_drop)]
    return {"drop_path": dp, "dropout2d": d2}


def make_learnable_batch(B, H, W, seed=1234, sigma=5.0, gain=1.2):
    """A synthetic batch whose ground truth is a deterministic, smooth function of the INPUT, so that a network can actually learn it
    (make_batch's ground truth is independent noise: training on it only drives the output to the mean).  Used for the RMSE gate at a
    trained operating point (tests/test_gpu_trained.py, tools/train_synth_checkpoint.py):
        z = gaussian_blur(0.8 ch0 - 0.5 ch1 + 0.3 ch2, sigma) scaled to unit variance,   depth = sigmoid(gain z)   in (0, 1)
    -- the reference's inverse-normalised depth (src/data/dataloader.py:236-257).  Ground truth: 20 % of the pixels, min-pooled
    half / quarter maps as in make_batch; radar channels at K ~ U{80..400} pixels carry the metric depth (1 - depth, i.e. d / 100 m)
    consistent with the ground truth (lib/fuse_radar.py:185-197), channels 4-6 as in make_batch."""
    from scipy.ndimage import gaussian_filter
    rs = np.random.RandomState(seed)
    x = np.zeros((B, 7, H, W), dtype=np.float32)
    x[:, 0:3] = rs.standard_normal(size=(B, 3, H, W)).astype(np.float32)
    mix = 0.8 * x[:, 0] - 0.5 * x[:, 1] + 0.3 * x[:, 2]
    z = np.stack([gaussian_filter(mix[b], sigma=sigma, mode="reflect") for b in range(B)])
    z = z * (2.0 * np.sqrt(np.pi) * sigma / np.sqrt(0.98))          # white noise of variance 0.98 blurred: variance 0.98 / (4 pi sigma^2)
    depth = (1.0 / (1.0 + np.exp(-gain * z))).astype(np.float32)
    depth = np.clip(depth, 0.01, 0.99)
    f = 0.8 * W
    for b in range(B):
        k = int(rs.randint(80, 401))
        k = min(k, H * W // 4)
        rows = rs.randint(0, H, size=k)
        cols = rs.randint(0, W, size=k)
        x[b, 3, rows, cols] = 1.0 - depth[b, rows, cols]
        x[b, 4, rows, cols] = ((cols - W / 2) / f + rs.normal(0, 0.01, size=k)).astype(np.float32)
        x[b, 5, rows, cols] = ((rows - H / 2) / f + rs.normal(0, 0.01, size=k)).astype(np.float32)
        x[b, 6, rows, cols] = (rs.uniform(size=k) < 0.15).astype(np.float32)
    valid = rs.uniform(size=(B, 1, H, W)) < 0.2
    gt = np.zeros((B, 1, H, W), dtype=np.float32)
    gt[valid] = depth[:, None][valid]
    gt_full = torch.from_numpy(gt)
    gt_half = min_pool_ignore_zero(gt_full)
    gt_quarter = min_pool_ignore_zero(gt_half)
    return {"image": torch.from_numpy(x), "gt_full": gt_full, "gt_half": gt_half, "gt_quarter": gt_quarter,
            "dense_depth": torch.from_numpy(depth[:, None].copy())}
