"""Device-side evaluation metrics: the counterpart of the per-frame loop in `Trainer.test`
(reference: src/main/runner.py:443-465).  The reference pulls every frame to the host and calls `.item()` three times
per frame; here one kernel reduces all frames of a batch and the host reads B x 4 floats once."""
import math

import torch

from . import lib as L


class DepthMetrics:
    """Accumulates MAE / RMSE / REL over frames exactly like Trainer.test: per-frame metrics, averaged over the frames
    that have at least one valid ground-truth pixel (frames without are skipped, runner.py:449-451)."""

    def __init__(self, max_depth=100.0, max_distance=100.0):
        self.max_depth, self.max_distance = float(max_depth), float(max_distance)
        self.rows = []          # device tensors [frames, 4], read on result()

    def update(self, pred, gt):
        """pred, gt: fp32 cuda tensors [B,1,H,W] (or [B,H,W]); normalised inverse... as produced by the model / dataloader."""
        if not pred.is_cuda:
            raise L.CrdError("DepthMetrics runs on the GPU (no CPU fallback; see oracle.losses.test_metrics for the CPU check)")
        pred = pred.detach().contiguous().float()
        gt = gt.detach().contiguous().float()
        frames = pred.shape[0]
        n = pred.numel() // frames
        acc = torch.zeros(frames, 4, dtype=L.SUM_DTYPE, device=pred.device)      # crd_sum_t (reproducible sums)
        L.check(L.load().crd_test_metrics(pred.data_ptr(), gt.data_ptr(), frames, n, self.max_depth, self.max_distance,
                                          acc.data_ptr(), L.stream()), "crd_test_metrics")
        self.rows.append(acc)

    def per_frame(self):
        a = L.stat_value(torch.cat(self.rows).cpu())
        out = []
        for sa, sq, sr, cnt in a.tolist():
            out.append(None if cnt == 0 else {"MAE": sa / cnt, "RMSE": math.sqrt(sq / cnt), "REL": sr / cnt})
        return out

    def result(self):
        ms = [m for m in self.per_frame() if m is not None]
        if not ms:
            return None
        return {k: sum(m[k] for m in ms) / len(ms) for k in ("MAE", "RMSE", "REL")}


class SegIoU:
    """The segmentation metric of Trainer.test (runner.py:432-436,508): per frame
    `JaccardIndex(num_classes, ignore_index=255)(pred_seg, gt_seg)` of torchmetrics 0.10.2 -- macro average over all
    num_classes classes of intersection / union from the confusion matrix of arg-max predictions, a class absent from both
    scoring 0 -- averaged over frames with np.nanmean.  Label 255 is NOT ignored by that call: ignore_index >= num_classes
    removes no class and torchmetrics raises ValueError on a target label >= num_classes, which the reference catches
    (runner.py:437-438), leaving that frame's IoU NaN.  One kernel builds the per-frame confusion matrices."""

    def __init__(self, num_classes=21):
        self.C = int(num_classes)
        self.mats, self.oor = [], []

    def update(self, logits, labels):
        """logits fp32 cuda [B,C,H,W], labels int64 cuda [B,H,W]."""
        if not logits.is_cuda:
            raise L.CrdError("SegIoU runs on the GPU (no CPU fallback; see oracle.losses.seg_iou for the CPU check)")
        logits = logits.detach().contiguous().float()
        labels = labels.detach().contiguous().to(torch.int64)
        B, C = logits.shape[0], logits.shape[1]
        assert C == self.C
        hw = logits.numel() // (B * C)
        mat = torch.zeros(B, C, C, dtype=torch.int64, device=logits.device)
        oor = torch.zeros(B, dtype=torch.int64, device=logits.device)
        L.check(L.load().crd_seg_confusion(logits.data_ptr(), labels.data_ptr(), B, C, hw, mat.data_ptr(), oor.data_ptr(), L.stream()),
                "crd_seg_confusion")
        self.mats.append(mat)
        self.oor.append(oor)

    def per_frame(self):
        mats, oor = torch.cat(self.mats).cpu().double(), torch.cat(self.oor).cpu()
        out = []
        for m, bad in zip(mats, oor.tolist()):
            if bad:
                out.append(float("nan"))
                continue
            inter = torch.diag(m)
            union = m.sum(0) + m.sum(1) - inter
            scores = torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(inter))
            out.append(float(scores.mean()))
        return out

    def result(self):
        """np.nanmean over the frames (NaN when every frame is NaN)."""
        vals = [v for v in self.per_frame() if not math.isnan(v)]
        return sum(vals) / len(vals) if vals else float("nan")
