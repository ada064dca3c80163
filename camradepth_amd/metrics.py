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
        acc = torch.zeros(frames, 4, device=pred.device)
        L.check(L.load().crd_test_metrics(pred.data_ptr(), gt.data_ptr(), frames, n, self.max_depth, self.max_distance,
                                          acc.data_ptr(), L.stream()), "crd_test_metrics")
        self.rows.append(acc)

    def per_frame(self):
        a = torch.cat(self.rows).cpu()
        out = []
        for sa, sq, sr, cnt in a.tolist():
            out.append(None if cnt == 0 else {"MAE": sa / cnt, "RMSE": math.sqrt(sq / cnt), "REL": sr / cnt})
        return out

    def result(self):
        ms = [m for m in self.per_frame() if m is not None]
        if not ms:
            return None
        return {k: sum(m[k] for m in ms) / len(ms) for k in ("MAE", "RMSE", "REL")}
