"""Data-parallel plumbing on CPU with the gloo backend (world_size 2): bucket layout of the flat gradient buffer,
SUM all-reduce semantics, and the global masked-mean bookkeeping of the losses.
No HIP kernel runs here; the GPU path is exercised by bench.py under torchrun."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camradepth_amd.model import CamRaDepth
        from camradepth_amd.trainer import GradSync
        from camradepth_amd.losses import _allreduce_acc
        torch.manual_seed(0)
        m = CamRaDepth(input_channels=7, depths=(1, 1, 1, 1))      # parameters live in one flat CPU buffer
        m._ensure_grad_views()
        sync = GradSync(m)
        assert sync.world == world
        # buckets tile the flat buffer exactly, in backward order
        spans = sorted(sync.ranges.values())
        assert spans[0][0] == 0 and spans[-1][1] == m.flat.numel()
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        lo, hi = sync.ranges[("dec",)]
        assert m._offsets[m._index["from_encoder_1.model.0.weight"]] == lo
        assert m._offsets[m._index["dest_encoder.block4.0.mlp1.norm2.bias"]] < lo
        # rank-dependent gradients -> SUM over ranks, identical on every rank
        m.flat_grad.copy_(torch.arange(m.flat.numel(), dtype=torch.float32) * (rank + 1) * 1e-6)
        sync.after_backward()
        expect = torch.arange(m.flat.numel(), dtype=torch.float32) * 1e-6 * sum(r + 1 for r in range(world))
        ok = torch.allclose(m.flat_grad, expect, rtol=1e-6)
        g = m._param("depth_activation_5.conv_2.bias").grad          # param.grad is a view of the reduced buffer
        ok = ok and torch.allclose(g, expect[m._offsets[m._index["depth_activation_5.conv_2.bias"]]:][:1])
        # global masked mean: (sum, count) partials are summed across ranks before the division
        acc = torch.tensor([2 * (rank + 1), 3 + rank, 0, 0], dtype=torch.int64)      # crd_sum_t partials: exact under SUM
        _allreduce_acc(acc)
        ok = ok and float(acc[0] / acc[1]) == pytest.approx((2.0 + 4.0) / (3.0 + 4.0))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gradsync_and_global_loss_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _worker_step(rank, world, port, q):
    """TrainStep.step()'s control flow (accumulation windows, loss all-reduce every iteration, bucket all-reduces and
    optimizer only on the closing iteration) with the HIP segments replaced by CPU stand-ins."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camradepth_amd.model import CamRaDepth
        from camradepth_amd.trainer import GradSync, TrainStep
        m = CamRaDepth(input_channels=7, depths=(1, 1, 1, 1))
        m._ensure_grad_views()
        import types
        ts = object.__new__(TrainStep)
        ts.state = types.SimpleNamespace()          # the shape-independent half (trainer.TrainState): counters, schedule, window
        ts.model, ts.sync = m, GradSync(m)
        ts.dist_active, ts.world, ts.update_interval, ts.use_graph, ts.graphs = True, world, 2, False, None
        ts.schedule, ts.lr, ts.betas, ts.eps, ts.wd = None, 1e-3, (0.9, 0.999), 1e-8, 0.0
        ts.iter_count = ts.epoch_iter = ts.sched_steps = ts.step_count = 0
        ts._window_open, ts._window_pos, ts._zero, ts._opt = False, 0, True, True
        ts.hp, ts.hp_ring, ts.acc = torch.zeros(8), [torch.zeros(8) for _ in range(4)], torch.zeros(16, dtype=torch.int64)
        import types
        ts.plan = types.SimpleNamespace(ensure_packed=lambda: None, packed_version=None)     # (the weight packing is a HIP segment too)
        ts._params, ts._frozen_sig = [], ()
        seen, accs = [], []

        def fwd():
            if ts._zero:
                m.flat_grad.zero_()
            ts.acc.zero_()
            ts.acc[0] += rank + 1
            ts.acc[1] += 1

        def bwd(key):
            lo, hi = ts.sync.ranges[key]
            m.flat_grad[lo:hi] += (rank + 1.0) * (ts.iter_count + 1)

        def optim():
            seen.append(m.flat_grad.clone())

        def segments():
            segs = [(fwd, "loss")] + [((lambda k=k: bwd(k)), k if ts._opt else None) for k in GradSync.ORDER]
            return segs + ([(optim, None)] if ts._opt else [])
        ts._segments = segments
        ran = []
        for it in range(5):
            ran.append(ts.step(last_of_epoch=(it == 4)))
            accs.append(float(ts.acc[0] / ts.acc[1]))
        ok = ran == [False, True, False, True, True] and ts.step_count == 3
        ok = ok and accs == [sum(r + 1.0 for r in range(world)) / world] * 5            # global mean on every iteration
        ranks = sum(r + 1.0 for r in range(world))
        for got, its in zip(seen, ((1, 2), (3, 4), (5,))):                               # SUM over ranks of the window's iterations
            ok = ok and torch.allclose(got, torch.full_like(got, ranks * sum(its)))
        ok = ok and not ts.sync.pending
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_train_step_control_flow_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_step, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _worker_late(rank, world, port, q):
    """The LATE-STREAM distributed variant of TrainStep (trainer._capture_variant / _replay_late: per backward segment a main graph,
    behind it on a second stream the segment's weight-gradient graph, then -- closing iteration of a window only -- that bucket's
    all-reduce and its optimizer slice) replayed end to end through TrainStep.step(), with CPU stand-ins for the HIP graphs and
    streams, against the single-process result computed in closed form."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        import types
        from camradepth_amd.model import CamRaDepth
        from camradepth_amd.trainer import GradSync, TrainStep
        torch.manual_seed(0)
        m = CamRaDepth(input_channels=7, depths=(1, 1, 1, 1))
        m._ensure_grad_views()
        p0 = m.flat.detach().clone()
        ts = object.__new__(TrainStep)
        ts.state = types.SimpleNamespace()
        ts.model, ts.sync = m, GradSync(m)
        ts.dist_active, ts.world, ts.update_interval, ts.use_graph = True, world, 2, True
        ts.schedule, ts.lr, ts.betas, ts.eps, ts.wd = None, 1e-3, (0.9, 0.999), 1e-8, 0.0
        ts.iter_count = ts.epoch_iter = ts.sched_steps = ts.step_count = 0
        ts._window_open, ts._window_pos, ts._zero, ts._opt = False, 0, True, True
        ts.hp, ts.hp_ring, ts.acc = torch.zeros(8), [torch.zeros(8) for _ in range(4)], torch.zeros(16, dtype=torch.int64)
        ts.plan = types.SimpleNamespace(ensure_packed=lambda: None, packed_version=None)
        ts._params, ts._frozen_sig = [], ()
        ts.late_stream = "late"
        log = []
        ts._current_stream = lambda: "main"
        ts._stream_wait = lambda waiter, on: log.append(("wait", waiter, on))
        ts._on_stream = lambda stream: contextlib.nullcontext()
        launch = ts.sync.launch
        ts.sync.launch = lambda key: (log.append(("allreduce", key)), launch(key))[1]

        class G:                                   # stand-in for a captured graph
            def __init__(self, tag, fn):
                self.tag, self.fn = tag, fn

            def replay(self):
                log.append(self.tag)
                self.fn()

        LR = 0.5

        def variant(zero, opt):
            def fwd():
                if zero:
                    m.flat_grad.zero_()
                ts.acc.zero_()
                ts.acc[0] += rank + 1
                ts.acc[1] += 1

            def late(key):                          # the bucket's weight gradients, rank- and iteration-dependent
                lo, hi = ts.sync.ranges[key]
                m.flat_grad[lo:hi] += (rank + 1.0) * (ts.iter_count + 1)

            def optim(key):                         # the bucket's optimizer slice on the REDUCED gradients
                lo, hi = ts.sync.ranges[key]
                with torch.no_grad():
                    m.flat[lo:hi] -= LR * m.flat_grad[lo:hi]
            chain = [(G(("main", key), lambda: None), G(("late", key), lambda key=key: late(key)), key,
                      G(("opt", key), lambda key=key: optim(key)) if opt else None) for key in GradSync.ORDER]
            return [(("late", G(("fwd",), fwd), chain, None), None)]
        ts.graphs = {(z, o): variant(z, o) for z in (True, False) for o in (True, False)}
        ran, accs = [], []
        for it in range(5):
            ran.append(ts.step(last_of_epoch=(it == 4)))
            accs.append(float(ts.acc[0] / ts.acc[1]))
        ok = ran == [False, True, False, True, True] and ts.step_count == 3
        ok = ok and accs == [sum(r + 1.0 for r in range(world)) / world] * 5
        # single-process result: every optimizer step applies the SUM over ranks and over the window's iterations
        ranks = sum(r + 1.0 for r in range(world))
        expect = p0 - LR * ranks * ((1 + 2) + (3 + 4) + 5)
        ok = ok and torch.allclose(m.flat.detach(), expect, rtol=1e-6, atol=1e-6) and not ts.sync.pending
        # order inside a closing iteration: per bucket main graph -> (late stream waits for main) -> late graph -> all-reduce ->
        # optimizer slice; buckets in backward order; the main stream joins the late stream at the end
        last = log[len(log) - 1 - log[::-1].index(("fwd",)):]
        want = [("fwd",)]
        for key in GradSync.ORDER:
            want += [("main", key), ("wait", "late", "main"), ("late", key), ("allreduce", key), ("opt", key)]
        want += [("wait", "main", "late")]
        ok = ok and last == want
        # ... and an iteration that only accumulates launches no collective on the gradients
        first = log[:log.index(("fwd",), 1)]
        ok = ok and not any(e[0] in ("allreduce", "opt") for e in first)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_late_stream_distributed_step_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_late, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_one_cycle_schedule_matches_torch():
    from camradepth_amd.trainer import one_cycle
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=6e-5, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=6e-5, total_steps=50, div_factor=2, pct_start=0.15)
    mine = one_cycle(50, 6e-5)
    for i in range(49):
        assert opt.param_groups[0]["lr"] == pytest.approx(mine[i][0], rel=1e-9)
        assert opt.param_groups[0]["betas"][0] == pytest.approx(mine[i][1], rel=1e-9)
        opt.step()
        sched.step()
