"""CPU oracle for the CamRaDepth losses. TEST INFRASTRUCTURE ONLY (see oracle/model.py).

Parity status: PINNED by tests/golden fixtures generated from the imported reference losses.
"""
import torch
import torch.nn.functional as F


def masked_smooth_l1(pred, target):
    """MaskedSmoothL1Loss.forward (reference: src/utils/loss_funcs.py:83-91); beta=1, mean over target>0."""
    assert pred.dim() == target.dim(), "inconsistent dimensions"
    m = (target > 0).detach()
    return F.smooth_l1_loss(pred[m], target[m])


def masked_mse(pred, target):
    """MaskedMSELoss.forward (reference: src/utils/loss_funcs.py:40-46)."""
    assert pred.dim() == target.dim(), "inconsistent dimensions"
    m = (target > 0).detach()
    d = (target - pred)[m]
    return (d ** 2).mean()


def masked_focal(logits, target, gamma=2):
    """MaskedFocalLoss.forward: focal transform of the SCALAR mean CE, ignore_index=255 (loss_funcs.py:25-31)."""
    ce = F.cross_entropy(logits, target, ignore_index=255)
    pt = torch.exp(-ce)
    return ((1 - pt) ** gamma * ce).mean()


def total_loss(out, batch, supervised_seg, update_interval=1):
    """Loss combination of Trainer.train_one_epoch (reference: src/main/runner.py:197-218).

    Returns (loss, parts) with parts = dict of the individual terms and the RMSE metric (:208).
    """
    final = out["depth"]["final_depth"]
    inter = out["depth"]["intermediate_depths"]
    seg = out["seg"]["final_seg"]
    l_seg = (masked_focal(seg, batch["seg"]) if seg is not None else 0) * (1 if supervised_seg else 0)
    l_half = masked_smooth_l1(inter[-1].squeeze(1), batch["gt_half"].squeeze(1))
    l_quarter = masked_smooth_l1(inter[-2].squeeze(1), batch["gt_quarter"].squeeze(1))
    l_full = masked_smooth_l1(final, batch["gt_full"])
    w = [1, 1, 1, 0.2, 0.2]
    loss = (w[0] * l_full + w[1] * l_half + w[2] * l_quarter + w[3] * l_seg + w[4] * 0) / sum(w)
    loss = loss / update_interval
    rmse = torch.sqrt(masked_mse(final, batch["gt_full"]))
    return loss, {"full": l_full, "half": l_half, "quarter": l_quarter, "seg": l_seg, "rmse": rmse}


def test_metrics(pred_full, gt_full, max_depth=100.0, max_distance=100.0):
    """RMSE / MAE / REL of Trainer.test for one frame (reference: src/main/runner.py:443-465)."""
    pred = torch.clip(pred_full.squeeze(), 0, 1) * max_depth
    gt = gt_full.squeeze().clone() * max_depth
    gt[gt > max_distance] = 0
    idx = torch.where(gt > 0)
    if len(idx[0]) == 0:
        return None
    err = pred[idx] - gt[idx]
    rel = torch.abs(err) / gt[idx]
    return {"MAE": err.abs().mean().item(), "RMSE": torch.sqrt((err ** 2).mean()).item(),
            "REL": (rel.sum() / len(rel)).item()}


def seg_iou(logits, target, num_classes=21):
    """The per-frame IoU of Trainer.test (reference: src/main/runner.py:432-438): torchmetrics 0.10.2
    `JaccardIndex(num_classes, ignore_index=255)(pred_seg, gt_seg)`.  PARITY UNPINNED: torchmetrics is not installed in
    the build container; this restates its published algorithm (functional/classification/jaccard.py,
    `_jaccard_from_confmat`): confusion matrix of arg-max predictions, per-class intersection / union with a class absent
    from both scoring absent_score = 0, macro mean over all classes; ignore_index = 255 >= num_classes removes no class,
    and a target label >= num_classes raises ValueError in the input checks, which the reference catches -> NaN."""
    t = target.reshape(-1)
    if int(t.max()) >= num_classes or int(t.min()) < 0:
        return float("nan")
    pred = logits.argmax(1).reshape(-1)
    conf = torch.bincount(t * num_classes + pred, minlength=num_classes * num_classes).reshape(num_classes, num_classes).double()
    inter = torch.diag(conf)
    union = conf.sum(0) + conf.sum(1) - inter
    scores = inter / union
    scores[union == 0] = 0.0
    return float(scores.mean())
