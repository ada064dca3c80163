"""Training-step driver: the counterpart of `Trainer.train_one_epoch`'s inner iteration
(reference: src/main/runner.py:179-270) for one process per GPU.

One step = zero grads -> forward -> masked losses -> backward -> (gradient all-reduce) -> diffGradNorm,
entirely as HIP kernels enqueued on one stream with static buffers, so the step is captured once
into HIP graphs and replayed; the host only updates five hyper-parameter floats per step
(OneCycleLR drives lr and beta1 every iteration, runner.py:151-152,270).

Data parallelism replaces nn.DataParallel (runner.py:135-136): batch-sharded replicas, gradients
SUM-all-reduced with RCCL in four buckets that become ready in backward order (decoder, stages 4+3,
stage 2, stage 1 + patch embeds), each launched as soon as its backward segment is enqueued so the
transfer overlaps the remaining backward.  The masked-mean denominators are made global first
(one 16-float all-reduce), which reproduces the reference's loss over the gathered batch exactly.
"""
import math
import os

import torch
import torch.distributed as dist

from . import lib as L
from . import trace
from .optim import _CHUNK

LOSS_W = (1.0, 1.0, 1.0, 0.2, 0.2)   # runner.py:213


def one_cycle(total_steps, max_lr, div_factor=2.0, pct_start=0.15, final_div_factor=1e4, base_m=0.85, max_m=0.95):
    """(lr, beta1) schedule of torch OneCycleLR(anneal='cos', cycle_momentum=True) as configured in runner.py:151-152."""
    initial, up_end = max_lr / div_factor, float(pct_start * total_steps) - 1
    min_lr = initial / final_div_factor

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)
    out = []
    for s in range(total_steps):
        if s <= up_end:
            pct = s / up_end if up_end > 0 else 1.0
            out.append((cos(initial, max_lr, pct), cos(max_m, base_m, pct)))
        else:
            pct = (s - up_end) / (total_steps - 1 - up_end)
            out.append((cos(max_lr, min_lr, pct), cos(base_m, max_m, pct)))
    return out


class GradSync:
    """Bucketed gradient all-reduce (SUM) over the flat gradient buffer, RCCL over xGMI."""

    ORDER = (("dec",), ("enc3", "enc2"), ("enc1",), ("enc0",))

    def __init__(self, model, group=None):
        self.model, self.group = model, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # CRD_FORCE_DIST=1: run the collectives even in a group of one (exercises the multi-GPU control flow on one GPU)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("CRD_FORCE_DIST") is not None)
        names, offs = model._names, model._offsets
        total = model.flat.numel()

        def first(prefix):
            return offs[next(i for i, n in enumerate(names) if n.startswith(prefix))]
        b2, b3, dec = first("dest_encoder.block2."), first("dest_encoder.block3."), first("from_encoder_1.")
        # flat layout: [patch embeds | block1 | block2 | block3 | block4 | decoder, heads, seg]
        self.ranges = {("dec",): (dec, total), ("enc3", "enc2"): (b3, dec), ("enc1",): (b2, b3), ("enc0",): (0, b2)}
        self.pending = []

    def bucket(self, key):
        lo, hi = self.ranges[key]
        return self.model.flat_grad[lo:hi]

    def launch(self, key):
        if self.active:
            self.pending.append(dist.all_reduce(self.bucket(key), op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def after_backward(self):
        """Eager-autograd path (model._grad_sync): reduce everything once the whole backward is enqueued."""
        for key in self.ORDER:
            self.launch(key)
        self.wait()


class TrainState:
    """What a training run carries from one iteration to the next, independent of the batch shape: diffGradNorm's state over the
    flat parameter buffer (exp_avg, exp_avg_sq, previous_grad, exp_grad_norm: diffGradNorm.py:62-71), the per-tensor block tables
    of the multi-tensor optimizer launch, the hyper-parameter upload ring, the (lr, beta1) schedule with its position, the
    optimizer step count of the bias corrections and the open accumulation window.  A TrainStep is the shape-specific half
    (plan, static input buffers, captured graphs); the reference's DataLoader has no drop_last (src/data/dataloader.py:40), so
    the last batch of an epoch is smaller and a second TrainStep for that shape must continue THIS state, not restart it."""

    def __init__(self, model, lr, betas, eps, weight_decay, update_interval, schedule):
        dev = model.flat.device
        n = model.flat.numel()
        self.m, self.v, self.pg = (torch.zeros(n, device=dev) for _ in range(3))
        nt = len(model._names)
        self.egn, self.fac = (torch.zeros(nt, device=dev) for _ in range(2))
        seg, b2s, b2c = [], [], []
        for t, (name, o) in enumerate(zip(model._names, model._offsets)):
            numel = model._param(name).numel()
            seg.append([o, o + numel])
            for c in range((numel + _CHUNK - 1) // _CHUNK):
                b2s.append(t)
                b2c.append(c)
        self.seg_host, self.b2s_host = seg, b2s
        self.seg = torch.tensor(seg, dtype=torch.int64, device=dev)
        self.b2s = torch.tensor(b2s, dtype=torch.int32, device=dev)
        self.b2c = torch.tensor(b2c, dtype=torch.int32, device=dev)
        self.nt, self.nblk = nt, len(b2s)
        self.nsq = torch.zeros(self.nblk, device=dev)        # per-workgroup parts of ||g||^2 (summed in a fixed order)
        self.hp = torch.zeros(8, device=dev)
        self.hp_ring = [torch.zeros(8).pin_memory() for _ in range(64)]
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.update_interval = update_interval
        self.schedule = schedule
        self.iter_count = 0          # iterations (micro-batches) seen
        self.epoch_iter = 0          # ... in the current epoch (scheduler lag, runner.py:269)
        self.sched_steps = 0         # scheduler.step() calls so far = index into `schedule`
        self.step_count = 0          # optimizer steps taken
        self._window_open, self._window_pos = False, 0


_STATE_FIELDS = ("m", "v", "pg", "egn", "fac", "seg", "b2s", "b2c", "nt", "nblk", "nsq", "hp", "hp_ring", "lr", "betas", "eps", "wd",
                 "update_interval", "schedule", "iter_count", "epoch_iter", "sched_steps", "step_count", "_window_open", "_window_pos")


class TrainStep:
    """One training ITERATION per step() call (runner.py:179-270).  With update_interval = k the gradients of k
    iterations accumulate in the flat gradient buffer (zeroed on the first, runner.py:175,266), the gradient all-reduce
    and diffGradNorm run on the k-th (runner.py:222,264), each loss is divided by k (runner.py:218).  The schedule
    entry used by an optimizer step follows the reference's scheduler lag: `scheduler.step()` is only called from the
    (k+1)-th iteration of an epoch on (runner.py:269-270); start_epoch() marks the epoch boundary.  Parameters with
    requires_grad=False are frozen: no weight-gradient launch is recorded for them and the optimizer skips them
    (diffGradNorm.py:54-55)."""

    def __init__(self, model, B, H, W, lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, update_interval=1,
                 use_graph=True, schedule=None, group=None, state=None):
        """state: the TrainState of another TrainStep of the same model (another batch shape of the same run) to continue;
        lr / betas / eps / weight_decay / update_interval / schedule are then taken from it."""
        assert model.training, "TrainStep drives the training path: call model.train() first"
        assert update_interval >= 1
        if state is not None and state.m.numel() != model.flat.numel():
            raise L.CrdError("TrainStep(state=...): the state belongs to a model with a different parameter count")
        self.state = state if state is not None else TrainState(model, lr, betas, eps, weight_decay, update_interval, schedule)
        self.model, self.B, self.H, self.W = model, B, H, W
        self.dev = model.flat.device
        self.lib = L.load()
        x = torch.zeros((B, model.cfg.input_channels, H, W), device=self.dev)
        self.sync = GradSync(model, group)
        self.world = self.sync.world
        self.dist_active = self.sync.active
        model.rng_rank = dist.get_rank(group) if dist.is_initialized() else 0
        # graph step: the weight gradients of every backward segment run as graphs of their own on a second stream, next to the
        # following segment's latency-bound chain; the decoder's streaming kernels with fewer workgroups (the plan sizes their
        # partial-copy buffers accordingly)
        from .engine import LATE_WGRAD, W3_LATE_WGS
        self.late_wgrad = bool(LATE_WGRAD and use_graph)
        model.w3_total_wgs = W3_LATE_WGS if self.late_wgrad else None
        model.__dict__["_need_grad"] = True
        self.plan = model._plan_for(x)
        model._ensure_grad_views()
        self.sup = model.cfg.supervised_seg
        self.gt = {"full": torch.zeros((B, 1, H, W), device=self.dev), "half": torch.zeros((B, 1, H // 2, W // 2), device=self.dev),
                   "quarter": torch.zeros((B, 1, H // 4, W // 4), device=self.dev),
                   "seg": torch.zeros((B, H, W), dtype=torch.int64, device=self.dev)}
        self.acc = torch.zeros(16, dtype=L.SUM_DTYPE, device=self.dev)     # crd_sum_t: 4 x (sum, count, sum sq, -) for full/half/quarter/ce
        seg, b2s, nt = self.state.seg_host, self.state.b2s_host, self.state.nt
        trainable = torch.tensor([1 if model._param(n_).requires_grad else 0 for n_ in model._names], dtype=torch.uint8)
        self.frozen_names = [n_ for n_ in model._names if not model._param(n_).requires_grad]
        self._params = [model._param(n_) for n_ in model._names]
        self._frozen_sig = tuple(p.requires_grad for p in self._params)      # fixed at construction: the plan records no weight
                                                                             # gradients for frozen tensors (checked in step())
        self.trainable_mask = trainable.to(self.dev) if self.frozen_names else None
        # per gradient bucket: its tensors' slice of the block tables and an `active` mask (the optimizer of a bucket can run
        # as soon as that bucket's gradients are final -- on the late stream, behind their un-packing)
        self.opt_parts = {}
        for key, (lo, hi) in self.sync.ranges.items():
            ts_ = [t for t, (a, _) in enumerate(seg) if lo <= a < hi]
            blks = [i for i, t in enumerate(b2s) if lo <= seg[t][0] < hi]
            assert ts_ == list(range(ts_[0], ts_[-1] + 1)) and blks == list(range(blks[0], blks[-1] + 1))
            mask = torch.zeros(nt, dtype=torch.uint8)
            mask[ts_[0]:ts_[-1] + 1] = 1
            self.opt_parts[key] = (blks[0], len(blks), (mask & trainable).to(self.dev))
        self.use_graph = use_graph
        self.graphs = None
        self._zero, self._opt = True, True

    def start_epoch(self):
        """Epoch boundary of the reference loop: the batch index restarts (scheduler lag) and pending accumulated
        gradients were flushed by step(last_of_epoch=True)."""
        self.epoch_iter = 0

    # ------------------------------------------------------------------ pieces of one step
    def _forward_and_loss_partials(self):
        p, st = self.plan, L.stream
        if self._zero:
            self.model.flat_grad.zero_()
        self.acc.zero_()
        p.forward(pack=False)                  # step() keeps the packed weights current (ensure_packed / _optimizer)
        for i, (j, key) in enumerate(((5, "full"), (4, "half"), (3, "quarter"))):
            pred, tgt = p.out_depth[j].t, self.gt[key]
            L.check(self.lib.crd_masked_l1_fwd(pred.data_ptr(), tgt.data_ptr(), pred.numel(), self.acc.data_ptr() + 32 * i, st()),
                    "crd_masked_l1_fwd")
        if self.sup:
            L.check(self.lib.crd_ce_fwd(p.seg_out.data_ptr(), self.gt["seg"].data_ptr(), self.B, self.model.cfg.num_classes,
                                        self.H * self.W, self.acc.data_ptr() + 96, st()), "crd_ce_fwd")

    def _loss_backward(self):
        p, st = self.plan, L.stream
        scale = 1.0 / sum(LOSS_W) / self.update_interval
        for i, (j, key) in enumerate(((5, "full"), (4, "half"), (3, "quarter"))):
            pred, tgt, d = p.out_depth[j].t, self.gt[key], p.out_depth[("grad", j)].t
            L.check(self.lib.crd_masked_l1_bwd(pred.data_ptr(), tgt.data_ptr(), pred.numel(), self.acc.data_ptr() + 32 * i, None,
                                               LOSS_W[i] * scale, d.data_ptr(), st()), "crd_masked_l1_bwd")
        if self.sup:
            L.check(self.lib.crd_ce_focal_bwd(p.seg_out.data_ptr(), self.gt["seg"].data_ptr(), self.B, self.model.cfg.num_classes,
                                              self.H * self.W, self.acc.data_ptr() + 96, None, LOSS_W[3] * scale,
                                              p.seg_grad_in.data_ptr(), st()), "crd_ce_focal_bwd")

    def _optimizer(self, key=None):
        """key: only the tensors of that gradient bucket (block-table slice + `active` mask)."""
        m = self.model
        b0, nb, mask = (0, self.nblk, self.trainable_mask) if key is None else self.opt_parts[key]
        L.check(self.lib.crd_diffgradnorm_step(m.flat.data_ptr(), m.flat_grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                               self.pg.data_ptr(), self.egn.data_ptr(), self.nsq.data_ptr() + 4 * b0, self.fac.data_ptr(),
                                               self.seg.data_ptr(), self.b2s.data_ptr() + 4 * b0, self.b2c.data_ptr() + 4 * b0,
                                               self.nt, nb, None if mask is None else mask.data_ptr(),
                                               0.0, 0.0, 0.0, 0.0, 0.0, 1, self.hp.data_ptr(), L.stream()),
                "crd_diffgradnorm_step")
        # the bucket's new weights in the kernels' bf16 layouts, right behind its update (late stream: under the encoder's
        # backward) instead of one 169-us launch at the head of the next forward
        lo, hi = (None, None) if key is None else self.sync.ranges[key]
        self.plan.pack(lo, hi)

    def _segments(self):
        """The iteration as a list of (callable, bucket-to-launch-after | None | 'loss'), for the current
        (self._zero, self._opt): zero the gradients first / all-reduce and run the optimizer last."""
        segs = [(self._forward_and_loss_partials, "loss")]
        first = True
        for key in GradSync.ORDER:
            def run(key=key, first=first):
                if first:
                    self._loss_backward()
                self.plan.backward(tags=key)
            segs.append((run, key if self._opt else None))
            first = False
        if self._opt:
            segs.append((self._optimizer, None))
        return segs

    def _variants(self):
        k = self.update_interval
        if k == 1:
            return [(True, True)]
        return [(True, False)] + ([(False, False)] if k > 2 else []) + [(False, True)]

    def _capture(self):
        """Captures one set of graphs per (zero gradients, optimizer) variant the accumulation schedule needs.  Capturing
        executes nothing; one eager warm-up iteration runs first (allocator, lazy module loading) and every buffer it
        changes is restored afterwards."""
        self.plan.ensure_packed()
        saved = [t.clone() for t in (self.model.flat, self.m, self.v, self.pg, self.egn, self.nsq, self.fac, self.model.flat_grad)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        self._zero, self._opt = True, True
        with torch.cuda.stream(s):
            for fn, _ in self._segments():
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.plan.fp8_grad_layers:
            # fp8 data gradients: the warm-up iteration above ran with just-in-time scaling and left every layer's scale = its dy's
            # amax / 448 on THIS batch (the calibration); the captured graphs use delayed scaling from here on -- each backward pass
            # starts by turning the previous pass's amax into the scales it quantises with (engine: crd_fp8_scale_update)
            self.plan.fp8_jit = False
        self.graphs = {}
        if self.late_wgrad:
            self.plan.split_late = True
            self.late_stream = (getattr(self, "late_stream_factory", None) or torch.cuda.Stream)()
        for zero, opt in self._variants():
            self._zero, self._opt = zero, opt
            self.graphs[(zero, opt)] = self._capture_variant()
        torch.cuda.synchronize()
        for t, sv in zip((self.model.flat, self.m, self.v, self.pg, self.egn, self.nsq, self.fac, self.model.flat_grad), saved):
            t.copy_(sv)
        self.plan.packed_version = None        # the warm-up iteration packed ITS updated weights

    def _capture_variant(self):
        graphs = []
        opt = self._opt
        if self.late_wgrad:
            # main stream:  [forward, loss] [decoder backward] [enc3+enc2 backward] [enc1] [enc0]            [optimizer]
            # late stream:                                    [decoder weight grads][enc3+enc2 ...]  ...  [enc0 ...]
            # (branches of ONE captured graph do not run concurrently on this stack; separate graphs on two streams do).
            # Multi-GPU: the loss all-reduce follows the first graph, and each bucket's gradient all-reduce is enqueued
            # behind its late graph (last iteration of an accumulation window only).
            segs = self._segments()
            main = torch.cuda.current_stream()
            bsegs = segs[1:-1] if opt else segs[1:]
            keys = list(GradSync.ORDER)
            g0 = None
            mains = []
            if self.dist_active:
                g0 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g0):
                    segs[0][0]()
            for i, (fn, _) in enumerate(bsegs):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if i == 0 and g0 is None:
                        segs[0][0]()
                    fn()
                mains.append(g)
            chain = []                                       # [(main graph, late graph, bucket key, optimizer-slice graph)]
            for g, key in zip(mains, keys):
                gl = torch.cuda.CUDAGraph()
                self.late_stream.wait_stream(main)
                with torch.cuda.graph(gl, stream=self.late_stream):
                    self.plan.run_late(key)
                    if opt and not self.dist_active:    # this bucket's gradients are final: its optimizer slice follows at once
                        self._optimizer(key)
                gopt = None
                if opt and self.dist_active:
                    # multi-GPU: the bucket's optimizer slice is a graph of its own, replayed on the late stream behind THAT
                    # bucket's all-reduce (step()), so that the update of the decoder's parameters overlaps the encoder's
                    # backward and only the last bucket's slice is exposed
                    gopt = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gopt, stream=self.late_stream):
                        self._optimizer(key)
                main.wait_stream(self.late_stream)
                chain.append((g, gl, key, gopt))
            return [(("late", g0, chain, None), None)]
        if not self.dist_active:      # no collective between the segments: the whole step is one graph (five fewer launches)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for fn, _ in self._segments():
                    fn()
            return [(g, None)]
        for fn, after in self._segments():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            graphs.append((g, after))
        return graphs

    # stream plumbing of the late-stream replay, as methods so that the CPU control-flow test (tests/test_ddp_cpu.py, gloo world 2)
    # can run the very same _replay_late with stand-ins for the HIP graphs and streams
    @staticmethod
    def _current_stream():
        return torch.cuda.current_stream()

    @staticmethod
    def _stream_wait(waiter, on):
        waiter.wait_stream(on)

    @staticmethod
    def _on_stream(stream):
        return torch.cuda.stream(stream)

    def _replay_late(self, g, opt):
        """One iteration of the late-stream variant (see _capture_variant): per backward segment its main graph, then -- on the late
        stream, behind it -- the segment's weight-gradient graph, and on the closing iteration of a window of a multi-GPU run that
        bucket's all-reduce and optimizer slice, while the main stream already runs the next segment."""
        _, g0, chain, go = g
        main = self._current_stream()
        if g0 is not None:
            g0.replay()
            dist.all_reduce(self.acc, group=self.sync.group)      # global loss denominators before the backward
        for gm, gl, key, gopt in chain:
            with trace.range("main:" + "+".join(key)):
                gm.replay()
            self._stream_wait(self.late_stream, main)
            with self._on_stream(self.late_stream), trace.range("late:" + "+".join(key)):
                gl.replay()
                if self.dist_active and opt:
                    self.sync.launch(key)      # this bucket's all-reduce, behind the graph that finishes its gradients
                    self.sync.wait()           # (the LATE stream waits for it; the main stream runs on)
                    gopt.replay()              # ... and the bucket's optimizer slice follows at once
        probe = getattr(self, "tail_probe", None)
        if probe is not None:                  # bench.py: how long the main stream idles behind the late stream at the end of an iteration
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
        self._stream_wait(main, self.late_stream)
        if probe is not None:
            e1.record(main)
            probe.append((e0, e1))
        if go is not None:
            go.replay()

    def set_batch(self, batch):
        self.plan.x_in.copy_(batch["image"], non_blocking=True)
        self.gt["full"].copy_(batch["gt_full"], non_blocking=True)
        self.gt["half"].copy_(batch["gt_half"], non_blocking=True)
        self.gt["quarter"].copy_(batch["gt_quarter"], non_blocking=True)
        if "seg" in batch:
            self.gt["seg"].copy_(batch["seg"], non_blocking=True)

    def step(self, last_of_epoch=False):
        """One iteration on the batch currently in the static input buffers (set_batch): forward, losses, backward into
        the accumulating gradient buffer and -- on every update_interval-th iteration or the last of an epoch
        (runner.py:222) -- gradient all-reduce + optimizer step.  Returns True when the optimizer ran."""
        if tuple(p.requires_grad for p in self._params) != self._frozen_sig:
            raise L.CrdError("camradepth_amd.TrainStep: requires_grad of a parameter changed after the step was built (its plan "
                             "and optimizer mask are fixed at construction): build a new TrainStep")
        k = self.update_interval
        zero = not self._window_open             # first iteration of an accumulation window: zero the gradients
        pos = self._window_pos if self._window_open else 0
        opt = (pos + 1 == k) or last_of_epoch
        self._zero, self._opt = zero, opt
        if opt:
            self.step_count += 1
            lr, b1 = (self.schedule[min(self.sched_steps, len(self.schedule) - 1)] if self.schedule else (self.lr, self.betas[0]))
            b2 = self.betas[1]
            bc1, bc2 = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
            hp_host = self.hp_ring[self.step_count % len(self.hp_ring)]   # ring: the async copy may still be pending
            hp_host[0], hp_host[1], hp_host[2], hp_host[3] = b1, b2, self.eps, self.wd
            hp_host[4] = lr * math.sqrt(bc2) / (bc1 + 1e-8)
            self.hp.copy_(hp_host, non_blocking=True)
        if self.use_graph and self.graphs is None:
            self._capture()
            self._zero, self._opt = zero, opt
        self.plan.ensure_packed()              # first step / parameters written from outside since the last one
        if self.use_graph and (zero, opt) not in self.graphs:      # e.g. a flush right after an update (last_of_epoch)
            self.graphs[(zero, opt)] = self._capture_variant()
        runs = self.graphs[(zero, opt)] if self.use_graph else [(None, a) for _, a in self._segments()]
        fns = None if self.use_graph else [f for f, _ in self._segments()]
        for i, (g, after) in enumerate(runs):
            if isinstance(g, tuple):           # ("late", first graph, [(main graph, late graph, bucket, optimizer slice)], -)
                self._replay_late(g, opt)
            elif g is not None:
                with trace.range("graph:%d" % i):
                    g.replay()
            else:
                with trace.range("segment:%d" % i):
                    fns[i]()
            if after == "loss":
                if self.dist_active:
                    dist.all_reduce(self.acc, group=self.sync.group)
            elif after is not None and self.dist_active:
                self.sync.launch(after)
                if after == GradSync.ORDER[-1]:
                    self.sync.wait()
        if opt:                                # every bucket was re-packed behind its optimizer slice
            self.model.mark_params_changed()
            self.plan.packed_version = self.model._param_version
        # bookkeeping of the reference loop
        self.iter_count += 1
        self.epoch_iter += 1
        self._window_open, self._window_pos = (not opt), (pos + 1)
        if self.epoch_iter > k:                # "to prevent a scheduler step before the optimizer step" (runner.py:269-270)
            self.sched_steps += 1
        return opt

    def losses(self):
        """Host view of the last iteration's loss terms (synchronises)."""
        a = L.stat_value(self.acc.cpu())
        if L.nonfinite():
            # a NaN / infinite / out-of-range partial was dropped from a fixed-point sum since the last check (include/camradepth_hip.h:
            # crd_nonfinite_status): the sums are not what the reference would have computed -- it reports NaN here, so do we
            nan = float("nan")
            return {"loss": nan, "full": nan, "half": nan, "quarter": nan, "seg": nan, "rmse": nan}
        full, half, quarter = (float(a[4 * i] / a[4 * i + 1]) for i in range(3))
        rmse = math.sqrt(float(a[2] / a[1]))
        seg = 0.0
        if self.sup:
            ce = float(a[12] / a[13])
            seg = (1 - math.exp(-ce)) ** 2 * ce
        total = (LOSS_W[0] * full + LOSS_W[1] * half + LOSS_W[2] * quarter + LOSS_W[3] * seg) / sum(LOSS_W) / self.update_interval
        return {"loss": total, "full": full, "half": half, "quarter": quarter, "seg": seg, "rmse": rmse}


def _delegate(name):
    return property(lambda self: getattr(self.state, name), lambda self, v: setattr(self.state, name, v))


for _n in _STATE_FIELDS:
    setattr(TrainStep, _n, _delegate(_n))
