"""HIP implementations of the CamRaDepth loss callables (same class names and call signature as
src/utils/loss_funcs.py:14-46,77-91).  Each loss is one masked-reduction kernel plus an analytic
backward kernel; no boolean-mask gather, no host sync.

Under data parallelism the reference computes every masked mean over the GATHERED global batch
(nn.DataParallel gathers outputs on device 0, src/main/runner.py:136,197-203).  To reproduce that
exactly with one process per GPU the (sum, count) partials are all-reduced before they are used,
so the value is the global loss and the local gradient is already divided by the global count
(gradients are then SUM-reduced across ranks, see parallel.GradSync).
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import lib as L


def _allreduce_acc(acc):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc)


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mode):
        lb = L.load()
        pred_c, target_c = pred.contiguous().float(), target.contiguous().float()
        acc = torch.zeros(4, dtype=L.SUM_DTYPE, device=pred.device)     # crd_sum_t: order-independent, exact under all-reduce
        L.check(lb.crd_masked_l1_fwd(pred_c.data_ptr(), target_c.data_ptr(), pred_c.numel(), acc.data_ptr(), L.stream()),
                "crd_masked_l1_fwd")
        _allreduce_acc(acc)
        ctx.save_for_backward(pred_c, target_c, acc)
        ctx.mode = mode
        a = L.stat_checked(acc)              # NaN when a non-finite partial was dropped (the reference's float sums report it)
        return ((a[0] if mode == "smooth_l1" else a[2]) / a[1]).float()

    @staticmethod
    def backward(ctx, gout):
        pred, target, acc = ctx.saved_tensors
        if ctx.mode != "smooth_l1":
            raise NotImplementedError("MaskedMSELoss is a metric in the reference (runner.py:208); no backward")
        lb = L.load()
        d = torch.empty_like(pred)
        g = gout.contiguous().float()
        L.check(lb.crd_masked_l1_bwd(pred.data_ptr(), target.data_ptr(), pred.numel(), acc.data_ptr(), g.data_ptr(), 1.0,
                                     d.data_ptr(), L.stream()), "crd_masked_l1_bwd")
        return d, None, None


class MaskedSmoothL1Loss(nn.Module):
    """SmoothL1(beta=1) mean over target > 0 (reference: src/utils/loss_funcs.py:77-91)."""

    def forward(self, pred, target):
        assert pred.dim() == target.dim(), "inconsistent dimensions"
        return _MaskedL1.apply(pred, target, "smooth_l1")


class MaskedMSELoss(nn.Module):
    """mean((target-pred)^2) over target > 0 (reference: src/utils/loss_funcs.py:36-46)."""

    def forward(self, pred, target):
        assert pred.dim() == target.dim(), "inconsistent dimensions"
        self.loss = _MaskedL1.apply(pred.detach(), target, "mse")
        return self.loss


class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        lb = L.load()
        lg = logits.contiguous().float()
        tg = target.contiguous().to(torch.int64)
        B, Cc = lg.shape[0], lg.shape[1]
        HW = lg.numel() // (B * Cc)
        acc = torch.zeros(4, dtype=L.SUM_DTYPE, device=lg.device)
        L.check(lb.crd_ce_fwd(lg.data_ptr(), tg.data_ptr(), B, Cc, HW, acc.data_ptr(), L.stream()), "crd_ce_fwd")
        _allreduce_acc(acc)
        ctx.save_for_backward(lg, tg, acc)
        a = L.stat_checked(acc)
        ce = (a[0] / a[1]).float()
        pt = torch.exp(-ce)
        return (1 - pt) ** 2 * ce

    @staticmethod
    def backward(ctx, gout):
        lg, tg, acc = ctx.saved_tensors
        lb = L.load()
        B, Cc = lg.shape[0], lg.shape[1]
        HW = lg.numel() // (B * Cc)
        d = torch.empty_like(lg)
        g = gout.contiguous().float()
        L.check(lb.crd_ce_focal_bwd(lg.data_ptr(), tg.data_ptr(), B, Cc, HW, acc.data_ptr(), g.data_ptr(), 1.0, d.data_ptr(),
                                    L.stream()), "crd_ce_focal_bwd")
        return d, None


class MaskedFocalLoss(nn.Module):
    """Focal transform (gamma=2) of the scalar mean cross entropy, ignore_index=255
    (reference: src/utils/loss_funcs.py:14-34)."""

    def __init__(self, weight=None, gamma=2, reduction="mean"):
        super().__init__()
        assert gamma == 2 and weight is None, "only the reference's configuration (gamma=2, no class weights) is implemented"
        self.gamma, self.reduction = gamma, reduction

    def forward(self, inputs, target):
        return _Focal.apply(inputs, target)


def total_loss(out, batch, supervised_seg, update_interval=1, criterion=None):
    """Loss combination of Trainer.train_one_epoch (reference: src/main/runner.py:197-218)."""
    crit = criterion or {"depth": MaskedSmoothL1Loss(), "seg": MaskedFocalLoss()}
    final, inter, seg = out["depth"]["final_depth"], out["depth"]["intermediate_depths"], out["seg"]["final_seg"]
    l_seg = (crit["seg"](seg, batch["seg"]) if seg is not None else 0) * (1 if supervised_seg else 0)
    l_half = crit["depth"](inter[-1].squeeze(1), batch["gt_half"].squeeze(1))
    l_quarter = crit["depth"](inter[-2].squeeze(1), batch["gt_quarter"].squeeze(1))
    l_full = crit["depth"](final, batch["gt_full"])
    w = [1, 1, 1, 0.2, 0.2]
    loss = (w[0] * l_full + w[1] * l_half + w[2] * l_quarter + w[3] * l_seg + w[4] * 0) / sum(w)
    return loss / update_interval, {"full": l_full, "half": l_half, "quarter": l_quarter, "seg": l_seg}
