"""Epoch-level entry points with the reference's names: `Trainer.train_one_epoch`, `Trainer.eval`, `Trainer.test`,
`Trainer.train` (reference: src/main/runner.py:166-270, 273-350, 352-392, 394-519), tying together what the hot path
provides -- TrainStep (HIP-graph training iteration incl. accumulation windows and the scheduler lag), InferenceGraph
(graph-replayed eval forward), the loss kernels and DepthMetrics / SegIoU.  No argparse, TensorBoard, tqdm or checkpoint
naming (out of scope); batches may be the reference dataloader's nested dict
(`batch["image"]`, `batch["gt"]["depth"]["lidar_depth"]`, `["lidar_depth_partial"]`, `batch["gt"]["seg"]["final_seg"]`,
src/data/dataloader.py:320-333) or the flat dict of camradepth_amd.synth / camradepth_amd.batch."""
import math
import time

import torch

from . import lib as L
from . import losses as HL
from .inference import InferenceGraph
from .metrics import DepthMetrics, SegIoU
from .trainer import TrainStep, one_cycle


def _nanmean(v):
    v = [x for x in v if x is not None and not (isinstance(x, float) and math.isnan(x))]
    return sum(v) / len(v) if v else float("nan")


def unpack_batch(batch, input_channels):
    """-> flat dict(image, gt_full, gt_half, gt_quarter[, seg]) on the host or device the batch lives on."""
    if "gt" not in batch:
        out = dict(batch)
        out["image"] = batch["image"][:, :input_channels]
        return out
    gt = batch["gt"]
    half, quarter = gt["depth"]["lidar_depth_partial"][0], gt["depth"]["lidar_depth_partial"][1]
    out = {"image": batch["image"].to(torch.float32)[:, :input_channels], "gt_full": gt["depth"]["lidar_depth"].to(torch.float32),
           "gt_half": half.to(torch.float32), "gt_quarter": quarter.to(torch.float32)}
    if gt.get("seg") is not None and gt["seg"].get("final_seg") is not None:
        out["seg"] = gt["seg"]["final_seg"].to(torch.long)
    if "name" in batch:
        out["name"] = batch["name"]
    return out


class Trainer:
    def __init__(self, model, train_dataloader=None, val_dataloader=None, test_dataloader=None, learning_rate=6e-5, num_epochs=1,
                 update_interval=1, div_factor=2.0, max_depth=100.0, max_distances=(100.0, 50.0), num_classes=21, group=None,
                 use_graph=True):
        if model.flat is None or not model.flat.is_cuda:
            raise L.CrdError("camradepth_amd.runner.Trainer needs the model on an MI355X (no CPU fallback)")
        self.model, self.cfg = model, model.cfg
        self.train_dataloader, self.val_dataloader, self.test_dataloader = train_dataloader, val_dataloader, test_dataloader
        self.learning_rate, self.num_epochs, self.update_interval, self.div_factor = learning_rate, num_epochs, update_interval, div_factor
        self.max_depth, self.max_distances, self.num_classes = max_depth, tuple(max_distances), num_classes
        self.group, self.use_graph = group, use_graph
        self.criterion = {"depth": HL.MaskedSmoothL1Loss(), "seg": HL.MaskedFocalLoss()}      # runner.py:149
        self.step = None                     # the TrainStep of the batch shape seen last
        self._steps, self._train_state = {}, None     # (B, H, W) -> TrainStep, all sharing one TrainState
        self._infer = {}                     # (B, H, W) -> InferenceGraph
        self.training_steps = self.val_steps = 0

    # ------------------------------------------------------------------ training (runner.py:166-270)
    def _train_step_for(self, b):
        """The TrainStep for this batch's shape.  The reference DataLoader keeps the smaller last batch of an epoch
        (src/data/dataloader.py:40: no drop_last), so a run sees (at least) two shapes: each gets its own plan and graphs, all of
        them continue ONE TrainState -- optimizer moments, bias-correction step count, OneCycle position, open accumulation
        window (an optimizer that restarted twice per epoch would be silent and wrong)."""
        B, _, H, W = b["image"].shape
        key = (B, H, W)
        if key not in self._steps:
            steps = len(self.train_dataloader) * self.num_epochs       # OneCycleLR(steps_per_epoch=len(loader), epochs): runner.py:151
            self.model.train()
            self._steps[key] = TrainStep(self.model, B, H, W, lr=self.learning_rate, update_interval=self.update_interval,
                                         schedule=one_cycle(max(steps, 2), self.learning_rate, div_factor=self.div_factor),
                                         use_graph=self.use_graph, group=self.group, state=self._train_state)
            self._train_state = self._steps[key].state
        self.step = self._steps[key]
        return self.step

    def train_one_epoch(self, epoch, save=False):
        """One pass over train_dataloader: every batch is one iteration (forward, losses / update_interval, backward into the
        accumulating gradients); the optimizer runs every update_interval-th batch and on the last one (runner.py:222); the
        scheduler lags as in runner.py:269-270.  Returns the epoch means the reference shows in its progress bar."""
        self.model.train()
        L.nonfinite()          # clear the sticky non-finite flag: what a validation pass raised must not turn this epoch's first loss into NaN
        n = len(self.train_dataloader)
        depth, stage4, rmse, seg = [], [], [], []
        for i, batch in enumerate(self.train_dataloader):
            b = unpack_batch(batch, self.cfg.input_channels)
            ts = self._train_step_for(b)
            if i == 0:
                ts.start_epoch()
            ts.set_batch(b)
            if ts.step(last_of_epoch=(i + 1 == n)):
                self.training_steps += 1
            v = ts.losses()                                   # (one host read per iteration, as the reference's .item() calls)
            depth.append(v["full"]); stage4.append(v["half"]); rmse.append(v["rmse"] * self.max_depth); seg.append(v["seg"])
        return {"depth_mean": _nanmean(depth), "depth_stage_4_mean": _nanmean(stage4), "RMSE": _nanmean(rmse), "seg_mean": _nanmean(seg)}

    # ------------------------------------------------------------------ validation (runner.py:273-350)
    def _forward_eval(self, image):
        B, _, H, W = image.shape
        key = (B, H, W)
        if key not in self._infer:
            self._infer[key] = InferenceGraph(self.model, B, H, W)
        return self._infer[key].run(image.cuda(non_blocking=True))

    def eval(self, epoch, save=False):
        """-> (val_loss, RMSE): the mean final-depth SmoothL1 loss and the mean RMSE (x max_depth) over val_dataloader."""
        self.model.eval()
        L.nonfinite()          # ... and the other way round.  The loss modules report a dropped non-finite partial as NaN
        rows = []              # (lib.stat_checked), which the nanmean below leaves out exactly like the reference's np.nanmean (runner.py:320-347)
        with torch.no_grad():
            for batch in self.val_dataloader:
                b = unpack_batch(batch, self.cfg.input_channels)
                out = self._forward_eval(b["image"])
                fd, inter = out["depth"]["final_depth"], out["depth"]["intermediate_depths"]
                gt = b["gt_full"].cuda()
                l_final = float(self.criterion["depth"](fd, gt))
                l_stage4 = float(self.criterion["depth"](inter[-1].squeeze(1), b["gt_half"].cuda().squeeze(1)))
                l_seg = float(self.criterion["seg"](out["seg"]["final_seg"], b["seg"].cuda())) if (out["seg"]["final_seg"] is not None and "seg" in b) else 0.0
                rmse = float(torch.sqrt(HL.MaskedMSELoss()(fd, gt))) * self.max_depth
                rows.append((l_final, l_stage4, rmse, l_seg))
                self.val_steps += 1
        self.model.train()
        return _nanmean([r[0] for r in rows]), _nanmean([r[2] for r in rows])

    def train(self, save=False):
        """runner.py:352-392 without checkpoint files: epochs of train_one_epoch + eval; returns the best validation loss."""
        best = float("inf")
        for epoch in range(self.num_epochs):
            self.train_one_epoch(epoch, save)
            val_loss, _ = self.eval(epoch, save) if self.val_dataloader is not None else (float("nan"), None)
            if val_loss < best:
                best = val_loss
        return best

    # ------------------------------------------------------------------ test (runner.py:394-519)
    def test(self, save=False):
        """Per-frame metrics of Trainer.test averaged with nanmean: RMSE / MAE / REL within max_distances[0], the second set
        with ground truth below max_distances[1] dropped as well (runner.py:489-491), IoU (supervised seg), mean forward time."""
        self.model.eval()
        m100, m50 = DepthMetrics(self.max_depth, self.max_distances[0]), DepthMetrics(self.max_depth, self.max_distances[0])
        iou = SegIoU(self.num_classes) if self.cfg.supervised_seg else None
        times = []
        with torch.no_grad():
            for batch in self.test_dataloader:
                b = unpack_batch(batch, self.cfg.input_channels)
                x = b["image"].cuda()
                torch.cuda.synchronize()
                t0 = time.time()
                out = self._forward_eval(x)
                torch.cuda.synchronize()                      # (the reference times an un-synchronised call, runner.py:417-420)
                times.append(time.time() - t0)
                fd, gt = out["depth"]["final_depth"], b["gt_full"].cuda().to(torch.float32)
                m100.update(fd, gt)
                gt50 = torch.where(gt * self.max_depth < self.max_distances[1], torch.zeros_like(gt), gt)
                m50.update(fd, gt50)
                if iou is not None and out["seg"]["final_seg"] is not None and "seg" in b:
                    iou.update(out["seg"]["final_seg"], b["seg"].cuda())
        self.model.train()
        r100, r50 = m100.result(), m50.result()
        res = {"time": _nanmean(times), "max_depth_%g" % self.max_distances[0]: r100, "max_depth_%g" % self.max_distances[1]: r50}
        if iou is not None:
            res["IoU"] = iou.result()
        return res
