#!/usr/bin/env python3
"""Headline benchmark: CamRaDepth training images/s at 256x416, bf16, on N MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full optimizer step (zero grads, forward, masked losses, backward, RCCL gradient
all-reduce, diffGradNorm) of BASELINE.json's config C2 per GPU: base model, batch 8, 7x256x416
synthetic image + sparse-radar batch, random-init weights.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FWD_GFLOP = {"base": 201.54, "supervised_seg": 364.74}   # SURVEY.md section 8(d), per image at 256x416
MFMA_BF16_PEAK_TFLOPS = 2500.0                             # MI355X_MICROARCH.md: dense bf16 MFMA peak


HBM_SUSTAINED_TBS = 6.3          # what a streaming kernel sustains on HBM3E (guide: ~8 TB/s peak; fills / copies measured 5.5-6.3)
MFMA_SUSTAINED_TFLOPS = 1500.0   # = 60 % of the nominal dense bf16 peak: the rate at the 1.5 GHz the chip holds under MFMA load
DISPATCH_FLOOR_US = 3.2          # a dependent kernel boundary inside a replayed graph (k_sum_partials_bf16 alone: tools/chain_table.py)


def floor_budget(plan):
    """The step's FLOOR with the current launch decomposition: every launch on the dependency chain at max(its algorithmic FLOPs at the
    sustained MFMA rate, its algorithmic bytes at the sustained HBM rate, one dispatch boundary); the late stream's weight
    gradients counted as fully hidden.  Per phase: launches, MFMA floor, byte floor, dependency floor (launches x 3.2 us) and the
    phase floor = sum over its launches of the max of the three.  Fusing launches lowers this floor; it is the yardstick for the
    kernels as they are, not for the model."""
    from camradepth_amd.engine import LATE
    marks = plan.fwd_marks + [("end", len(plan.fwd))]
    phases = {}

    def add(name, op):
        ph = phases.setdefault(name, {"launches": 0, "mfma_ms": 0.0, "byte_ms": 0.0, "dep_ms": 0.0, "floor_ms": 0.0, "gflop": 0.0, "gbyte": 0.0})
        fl = float((op.meta or {}).get("flops", 0.0))
        by = float(plan.op_bytes(op))
        n = 1 + (op.meta or {}).get("kernel", "").count("+")
        tm, tb, td = fl / (MFMA_SUSTAINED_TFLOPS * 1e9), by / (HBM_SUSTAINED_TBS * 1e9), n * DISPATCH_FLOOR_US * 1e-3
        ph["launches"] += n; ph["mfma_ms"] += tm; ph["byte_ms"] += tb; ph["dep_ms"] += td; ph["floor_ms"] += max(tm, tb, td)
        ph["gflop"] += fl / 1e9; ph["gbyte"] += by / 1e9
    for (name, a), (_, b) in zip(marks[:-1], marks[1:]):
        for op in plan.fwd[a:b]:
            if plan.live(op):
                add("fwd:" + name, op)
    for op in plan.fwd[:marks[0][1]]:
        if plan.live(op):
            add("fwd:" + marks[0][0], op)
    for tag, a, b in plan.bwd_segments:
        for op in plan.bwd[a:b]:
            if plan.live(op):
                add(("late:" if op.stream == LATE else "bwd:") + tag, op)
    for ph in phases.values():
        for k in ph:
            ph[k] = round(ph[k], 3) if isinstance(ph[k], float) else ph[k]
    chain = sum(v["floor_ms"] for k, v in phases.items() if not k.startswith("late:"))
    return {"floor_ms": round(chain, 3), "late_stream_floor_ms": round(sum(v["floor_ms"] for k, v in phases.items() if k.startswith("late:")), 3),
            "assumes": f"{MFMA_SUSTAINED_TFLOPS:.0f} TFLOP/s MFMA, {HBM_SUSTAINED_TBS} TB/s HBM, {DISPATCH_FLOOR_US} us per dependent launch; "
                       "late-stream weight gradients, optimizer slices and weight packing hidden",
            "phases": phases}


def excess_attribution(ts, step_ms, tail_ms, reps=20, warm_reps=10):
    """Where the time between the WARM chain and the replayed step goes, measured in this process on the captured
    graphs of the step:
      warm_chain_ms   sum over the launches of the dependency chain of each launch replayed ALONE from a HIP graph (its inputs warm in the
                      caches: tools/floor_table.py's figure),
      chain_only_ms   the chain's own graphs replayed back to back with the late stream's graphs left out (nothing beside them),
      late_only_ms    the late stream's graphs (weight gradients, un-packing, optimizer slices, re-pack) replayed alone, one after the other,
      cold       = chain_only - warm_chain : what a launch pays for running right behind its producer instead of behind itself (the
                   non-coherent L2s written back / invalidated at every kernel boundary, first-touch misses, dispatch gaps),
      interference = step - tail - chain_only : what the late stream's kernels cost the chain's while they run beside them,
      tail       = the main stream's idle time behind the late stream at the end of an iteration (exposed_tail_us)."""
    plan = ts.plan
    key = (True, True)
    g = ts.graphs[key][0][0]
    if not (isinstance(g, tuple) and g[0] == "late"):
        return None
    _, g0, chain, _ = g
    if g0 is not None:
        return None                                   # (the distributed step has collectives between the graphs: not replayable piecewise)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    chain_only = timed(lambda: [gm.replay() for gm, _, _, _ in chain], reps)
    late_only = timed(lambda: [gl.replay() for _, gl, _, _ in chain], reps)

    def warm(op):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            plan.run_ops([op])
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg, stream=st):
                for _ in range(warm_reps):
                    plan.run_ops([op])
            gg.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gg.replay()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / warm_reps
    from camradepth_amd.engine import LATE
    saved = plan.split_late
    plan.split_late = False
    warm_chain = warm_late = 0.0
    n_chain = 0
    for op in plan.fwd + plan.bwd:
        if not plan.live(op):
            continue
        t = warm(op)
        if op.stream == LATE:
            warm_late += t
        else:
            warm_chain += t
            n_chain += 1 + (op.meta or {}).get("kernel", "").count("+")
    plan.split_late = saved
    return {"step_ms": round(step_ms, 3), "warm_chain_ms": round(warm_chain, 3), "chain_only_ms": round(chain_only, 3),
            "late_only_ms": round(late_only, 3), "warm_late_ms": round(warm_late, 3), "chain_launches": n_chain,
            "cold": round(chain_only - warm_chain, 3), "interference": round(step_ms - tail_ms - chain_only, 3), "tail": round(tail_ms, 3),
            "what": "step = warm_chain + cold + interference + tail; cold: chain graphs replayed without the late stream minus the sum of "
                    "its launches replayed alone; interference: step minus tail minus chain-only (the losses / optimizer launches of the "
                    "forward graph are part of the chain)"}


def per_kernel_timing(ts, reps=3):
    """Times every kernel launch of one step with HIP events on the launch stream and aggregates the MFMA kernels
    by template instance: {kernel: (launches, total ms, algorithmic flops)} per step."""
    plan = ts.plan
    import camradepth_amd.lib as L
    agg = {}
    st = L.stream()
    # An event pair around an eager launch also times the launch path (~5 us), which would make the ~380 small encoder
    # GEMMs look like the dominant kernel.  Calibrate it: the same pair around a 64-float scale kernel, whose true cost
    # inside a captured graph is 1.7 us (tools/bench_launch.py), and take the difference off every measured launch.
    pad = torch.zeros(64, device=plan.dev)
    lib = L.load()
    cal = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.crd_scale_f32(pad.data_ptr(), pad.data_ptr(), 64, 1.0, st)
        e1.record()
        cal.append((e0, e1))
    torch.cuda.synchronize()
    null_ms = sorted(e0.elapsed_time(e1) for e0, e1 in cal)[len(cal) // 2]
    overhead_ms = max(null_ms - 0.0017, 0.0)
    for _ in range(reps):
        ts._forward_and_loss_partials()           # leaves valid activations / loss partials for the backward ops
        ts._loss_backward()
        plan.zb_arena.zero_()
        for ops in (plan.fwd, plan.bwd):
            if ops is plan.fwd:
                plan.zf_arena.zero_()
            evs = []
            for op in ops:
                if not plan.live(op):              # join marker of a side branch (this pass runs everything on one stream) / inactive variant
                    continue
                if op.meta is None:
                    op.fn(*op.args, st)
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                op.fn(*op.args, st)
                e1.record()
                evs.append((op.meta, e0, e1))
            torch.cuda.synchronize()
            for meta, e0, e1 in evs:
                a = agg.setdefault(meta["kernel"], [0, 0.0, 0.0])
                a[0] += 1
                a[1] += max(e0.elapsed_time(e1) - overhead_ms, 0.001)
                a[2] += meta["flops"]
    return {k: (v[0] / reps, v[1] / reps, v[2] / reps) for k, v in agg.items()}


def family_of(kernel):
    """Kernel template family of a per-op label: the name before '<'; the 3x3 halo convolution is ONE family -- the persistent
    kernel k_conv3x3p<...> and the two-workgroup kernel k_conv3x3<...> that takes the small grids and the ragged column
    tails -- i.e. every rocprof row `k_conv3x3*` except the e4m3 kernel."""
    key = kernel.split("<")[0]
    return "k_conv3x3" if key == "k_conv3x3p" else key


def family_replay_timing(ts, family, reps=10):
    """All launches of one kernel family of the step -- forward and backward, in plan order, on the activations the last
    step left -- captured into ONE HIP graph and replayed back to back: HIP events around `reps` replays on the stream the
    kernels are launched on.  Back-to-back replay keeps the chip under sustained load, as inside the step (eager launches
    with host gaps between them run at a higher boost clock: the isolated per-launch table reads 5-7 % faster than the
    rocprofv3 trace of a replayed step).  Returns (device launches per replay, algorithmic flops, ms per replay)."""
    import camradepth_amd.lib as L
    plan = ts.plan
    ops = [op for op in plan.fwd + plan.bwd if plan.live(op) and op.meta is not None and family_of(op.meta["kernel"]) == family]
    n = sum(1 + op.meta["kernel"].count("+") for op in ops)
    flops = sum(op.meta["flops"] for op in ops)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for op in ops:                                   # warm-up outside the capture
            op.fn(*op.args, L.stream())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for op in ops:
                op.fn(*op.args, L.stream())
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
    torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(s)
    return n, flops, e0.elapsed_time(e1) / reps


def csrc_sha():
    """sha256 over the HIP sources: the PMC traffic file is only valid for the kernels it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "camradepth_amd", "csrc", "*.hip")) + glob.glob(os.path.join(REPO, "camradepth_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, made by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE passes of this same command: FETCH_SIZE doubled per the gfx950
    correction, + WRITE_SIZE).  Counters cannot be read from inside the timed process, so the file carries the sha of the
    kernel sources it was measured on; a stale file (different sha), another workload or a missing file gives None."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    data = json.load(open(path))
    if data.get("csrc_sha") != csrc_sha():
        return None
    ks = data["kernels"]
    prefix = kernel.rstrip("*").replace(" ", "")       # "k_conv3x3*": every instantiation of both 3x3 halo kernels
    n = tot = 0.0
    for name, v in ks.items():
        if name.replace(" ", "").startswith(prefix):
            n += v["launches"]
            tot += v["hbm_bytes_per_launch"] * v["launches"]
    return round(tot / n) if n else None


def trace_frac(kernel, flops):
    """The same family's fraction of the MFMA peak in the COMMITTED rocprofv3 trace of one replayed step (profiles/
    the newest rNN_one_step_kernels.txt, made by tools/profile_round.sh from this command; valid while profiles/pmc_traffic.json carries the
    sha of the current kernel sources, i.e. both were refreshed together): sum of the family's rows there.  The live `frac` is
    measured with the family replayed back to back; the trace's per-kernel durations include the dispatch boundary (~3 us per
    launch) and the traced step runs ~10 % slower, so it reads 0.01-0.02 lower.  None when the files are absent or stale."""
    import glob
    import re
    tables = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_one_step_kernels.txt")))
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not (tables and os.path.exists(pmc)) or json.load(open(pmc)).get("csrc_sha") != csrc_sha():
        return None
    path = tables[-1]
    prefix, ms = kernel.rstrip("*"), 0.0
    for line in open(path):
        m = re.match(r"\s*([\d.]+) ms\s+\d+x\s+[\d.]+ us\s+(\S+)", line)
        if m and m.group(2).startswith(prefix):
            ms += float(m.group(1))
    return round(flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if ms > 0 else None


def forward_only(model, batch, B, H, W, variant, reps=20):
    """Eval-mode forward (forward-only roofline fraction), replayed from one HIP graph."""
    from camradepth_amd.inference import InferenceGraph
    ig = InferenceGraph(model, B, H, W)
    ig.run(batch["image"].cuda())
    with torch.cuda.stream(ig.stream):
        for _ in range(3):
            ig.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ig.replay()
        e1.record()
    torch.cuda.synchronize()
    model.train()
    ms = e0.elapsed_time(e1) / reps
    ips = B / (ms * 1e-3)
    scale = (H * W) / (256 * 416)
    return {"images_per_s": round(ips, 1), "ms": round(ms, 3),
            "mfma_frac": round(ips * FWD_GFLOP[variant] * scale / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4)}


def cpu_baseline(variant, B_gpu=8, H=256, W=416, warm=2, timed_steps=5, budget_s=150.0):
    """The CPU oracle (oracle/, a port of the reference verified against it) timed on this box's host cores, as SURVEY 8(d)
    prescribes: fp32, B = 2 and the GPU's per-device batch, 2 warm-up + 5 timed train steps (forward, losses, backward, diffGradNorm)
    each, and forward-only passes.  Threads = min(host CPUs, 64): torch's default on a 256-CPU box is 128 oversubscribed oneDNN threads,
    which ran HALF as fast (0.24 vs 0.46 images/s, BENCH_r03 vs r02).  The sample is bounded: when the first warm-up step predicts more
    than `budget_s` seconds for a leg, its timed count is cut (never below 2) and `sample` says so.  `value` = the per-device-batch
    train-step rate."""
    from camradepth_amd import synth
    from camradepth_amd.config import ModelConfig
    from camradepth_amd.params import param_specs
    from oracle import losses as ol
    from oracle import model as om
    from oracle import optim as oo
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    cfg = ModelConfig.variant(variant)
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0).items()}
    states = {k: oo.new_state(v.detach()) for k, v in sd.items()}

    def train(batch, masks):
        for v in sd.values():
            v.grad = None
        out = om.forward(sd, batch["image"], cfg, masks=masks)
        loss, _ = ol.total_loss(out, batch, cfg.supervised_seg)
        loss.backward()
        with torch.no_grad():
            for k, v in sd.items():
                if v.grad is not None:
                    oo.step_tensor(v, v.grad, states[k], 6e-5, 0.9, 0.999)

    def fwd(batch):
        with torch.no_grad():
            om.forward(sd, batch["image"], cfg)

    def leg(fn, n_warm, n_timed, budget):
        t0 = time.time()
        fn()
        first = time.time() - t0
        for _ in range(n_warm - 1):
            fn()
        n = max(2, min(n_timed, int(budget / max(first, 1e-3)) - n_warm))
        t0 = time.time()
        for _ in range(n):
            fn()
        return (time.time() - t0) / n, n
    legs, counts = {}, {}
    for Bc in sorted({2, B_gpu}):
        bB = synth.make_batch(Bc, H, W, seed=1234)
        mB = synth.make_masks(cfg, Bc, seed=4321)
        share = budget_s * (0.2 if Bc < B_gpu else 0.6)
        dt, n = leg(lambda: train(bB, mB), warm, timed_steps, share)
        legs[f"train_b{Bc}"], counts[f"train_b{Bc}"] = round(Bc / dt, 3), n
        dt, n = leg(lambda: fwd(bB), 1, 3, budget_s * 0.1)
        legs[f"forward_b{Bc}"], counts[f"forward_b{Bc}"] = round(Bc / dt, 3), n
    return {"value": legs[f"train_b{B_gpu}"], "unit": "images/s", "cores": threads, "kind": "port", "legs_images_per_s": legs,
            "timed_passes": counts,
            "sample": f"fp32 CPU oracle, {variant}, 7x{H}x{W}: {warm} warm-up + up to {timed_steps} timed train steps (fwd+loss+bwd+diffGradNorm) "
                      f"at batch 2 and batch {B_gpu} [= value], 1 warm-up + up to 3 timed forward-only passes each (timed counts in "
                      f"timed_passes; cut only to stay inside {budget_s:.0f} s); {threads} torch threads on {cores} host CPUs"}


# ---- N ranks without a launcher ------------------------------------------------------------------------------------
def rank_env(rank, world, port, base=None):
    """Environment of rank `rank` of `world` single-node ranks (one per GPU; rendezvous on 127.0.0.1: the container's
    host name may not resolve).  HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller set it: RCCL needs dmabuf IPC on this driver."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def free_port():
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def spawn_ranks(world, argv, timeout_s=None):
    """`python bench.py --gpus N` started without torchrun: the reference's parallelism is one call (nn.DataParallel,
    src/main/runner.py:135-136), so is this.  N fresh child processes are started BEFORE this process makes any HIP call
    (a process that has touched the GPU must not exec another); rank 0's stdout -- the one JSON line -- is passed through.
    All children are polled: the first non-zero exit (or the overall timeout, CRD_BENCH_TIMEOUT seconds, default 1800)
    terminates the others (kill after a grace period) and is the launcher's exit code -- a rank that dies early must not leave
    the rest in a rendezvous or a collective forever."""
    timeout_s = float(os.environ.get("CRD_BENCH_TIMEOUT", "1800")) if timeout_s is None else timeout_s
    port = free_port()
    import tempfile
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=rank_env(r, world, port),
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    t0, rc, live = time.time(), 0, set(range(world))
    while live and rc == 0:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0:
                    rc = c
        if time.time() - t0 > timeout_s:
            rc = 124
        if live and rc == 0:
            time.sleep(0.05)
    if live:                                   # a rank failed or the run timed out: take the others down
        for r in live:
            procs[r].terminate()
        t1 = time.time()
        while any(procs[r].poll() is None for r in live) and time.time() - t1 < 10:
            time.sleep(0.05)
        for r in live:
            if procs[r].poll() is None:
                procs[r].kill()
            procs[r].wait()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return rc


def stub_rank(a):
    """--stub (tests): the launcher contract without a GPU -- gloo process group, K sleeps as 'steps' between the same
    barriers, MAX over ranks of the elapsed time, one JSON line from rank 0."""
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("CRD_STUB_FAIL_RANK") == str(rank):      # tests: a rank that dies before the rendezvous
        sys.exit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.001 * (rank + 1))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": metric_name(a), "value": round(a.batch * world * a.steps / float(t), 2), "unit": "images/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "data": "stub"}), flush=True)


def metric_name(a):
    return f"training images/sec at {a.height}x{a.width} bf16"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 warm-up + 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (config C2: 8)")
    ap.add_argument("--variant", default="base", choices=["base", "supervised_seg"])
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=416)
    ap.add_argument("--freeze-seg", action="store_true",
                    help="config C4: transfer learning with the segmentation branch frozen (seg_* parameters requires_grad=False)")
    ap.add_argument("--update-interval", type=int, default=1, help="gradient accumulation (runner.py:218-222)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-excess", action="store_true", help="skip the warm-chain / chain-only / late-only attribution of the step time")
    ap.add_argument("--inference", action="store_true",
                    help="SURVEY 8f N4: eval-mode forward only, replayed from one HIP graph (e.g. --batch 1 --height 416 --width 800)")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE config 5 -- the ConvLayers of the two largest decoder stages on the fp8 (e4m3) MFMA, activation "
                         "scales calibrated on the bench batch (model.calibrate_fp8); in a training step: fp8 forward, bf16 backward")
    ap.add_argument("--fp8-grad", action="store_true",
                    help="with --fp8 in a training step: the data gradients of the same ConvLayers in e4m3 too (delayed per-tensor scaling); "
                         "weight gradients stay bf16")
    ap.add_argument("--tune-narrow", type=int, default=None, choices=[0, 1, 2],
                    help="A/B aid: crd_tune_pw_narrow(n) before the plans are built (0: the narrow streaming pointwise kernel off)")
    ap.add_argument("--tune-rege", type=int, default=None, choices=[0, 1],
                    help="A/B aid: crd_tune_igemm_reg_epilogue(n) (0: the LDS-staged epilogue for the 64 x 64 igemm tiles)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)      # tests: launcher contract on CPU (gloo)
    a = ap.parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:       # no launcher: be the launcher (before any HIP call in this process)
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    if a.stub:
        return stub_rank(a)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    # CRD_FORCE_DIST=1 (developer aid): take the multi-GPU code path -- RCCL process group, loss all-reduce, bucketed
    # gradient all-reduce between the per-segment graphs -- with a single rank, to exercise it on a 1-GPU box
    force_dist = os.environ.get("CRD_FORCE_DIST") is not None
    if world > 1 or force_dist:
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "NONE"      # every rank would print RCCL's banner to stdout, around rank 0's JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from camradepth_amd import synth
    from camradepth_amd.model import CamRaDepth
    from camradepth_amd.trainer import TrainStep, one_cycle

    if a.tune_narrow is not None:
        import camradepth_amd.lib as _L
        _L.load().crd_tune_pw_narrow(a.tune_narrow)
    if a.tune_rege is not None:
        import camradepth_amd.lib as _L
        _L.load().crd_tune_igemm_reg_epilogue(a.tune_rege)
    sup = a.variant == "supervised_seg"
    model = CamRaDepth(input_channels=7, supervised_seg=sup, seed=0).cuda()      # same init on every rank
    if a.inference:       # the reference's own "runtime" figure (Trainer.test(), runner.py:417-420), without its missing device sync
        batch = synth.make_batch(a.batch, a.height, a.width, seed=1234)
        scales = model.calibrate_fp8(batch["image"].cuda()) if a.fp8 else None
        r = forward_only(model, batch, a.batch, a.height, a.width, a.variant, reps=max(a.steps, 10))
        line = {"metric": "inference frames/sec (eval forward, HIP graph)", "value": r["images_per_s"], "unit": "images/s",
                "n_gpus": 1, "ms_per_forward": r["ms"], "higher_is_better": True, "dtype": "fp8" if a.fp8 else "bf16", "data": "synthetic",
                "config": {"workload": f"CamRaDepth {a.variant} eval forward, {a.batch}x7x{a.height}x{a.width}"
                                       + (", decoder stages 3-4 ConvLayers in e4m3" if a.fp8 else "")},
                "mfma_frac": r["mfma_frac"]}        # (fraction of the bf16 peak in both cases: the flops are the same)
        if scales:
            line["fp8_activation_scales"] = {str(k): v for k, v in scales.items()}
        print(json.dumps(line), flush=True)
        return
    model.train()
    if a.fp8:             # config 5 as a training step: fp8 forward convolutions in decoder stages 3-4, bf16 backward
        model.calibrate_fp8(synth.make_batch(a.batch, a.height, a.width, seed=1234)["image"].cuda(), train=True, grads=a.fp8_grad)
    if a.freeze_seg:
        for n, p in model.named_parameters():
            if n.startswith("seg_"):
                p.requires_grad_(False)
    total_sched = max(a.steps + a.warmup + 8, 64)
    ts = TrainStep(model, a.batch, a.height, a.width, lr=6e-5, schedule=one_cycle(total_sched, 6e-5),
                   use_graph=not a.no_graph, update_interval=a.update_interval)
    batch = synth.make_batch(a.batch, a.height, a.width, seed=1234 + rank)
    ts.set_batch({k: v.cuda() for k, v in batch.items()})

    for _ in range(a.warmup):
        ts.step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    if getattr(ts, "late_wgrad", False):
        ts.tail_probe = []                    # (two event records per step on the main stream; no synchronisation)
    for i in range(a.steps):
        ts.step()
        marks[i + 1].record()                 # (an event record per step: the median step time next to the mean the value is made of)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    losses = ts.losses()
    ms = 1e3 * dt / a.steps
    value = a.batch * world * a.steps / dt

    out = {"metric": metric_name(a), "value": round(value, 2), "unit": "images/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": ("fp8 forward + data gradients (decoder stages 3-4) / bf16" if a.fp8_grad else "fp8 forward (decoder stages 3-4) / bf16") if a.fp8 else "bf16", "data": "synthetic",
           "config": {"workload": f"CamRaDepth {a.variant}{' (seg branch frozen)' if a.freeze_seg else ''} (image+radar) train "
                                  f"{'iteration' if a.update_interval > 1 else 'step'}, {a.batch}x7x{a.height}x{a.width} per GPU, "
                                  f"bf16 MFMA / fp32 accumulate, diffGradNorm + OneCycleLR, Dropout2d/DropPath on"
                                  + (f", gradients accumulated over {a.update_interval} iterations" if a.update_interval > 1 else ""),
                      "global_batch": a.batch * world, "parallelism": f"dp{world}", "hip_graph": not a.no_graph},
           "ms_per_step_median": round(per_step[len(per_step) // 2], 3),
           "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
           "loss": round(losses["loss"], 6), "rmse_norm": round(losses["rmse"], 6)}
    if getattr(ts, "tail_probe", None):
        # EXPOSED tail: what the main stream waits for after its last backward graph -- the last bucket's weight gradients, (multi-GPU)
        # that bucket's all-reduce, and its optimizer slice + weight re-pack.  With N > 1 ranks only this and the loss all-reduce are
        # not overlapped with the backward: the prediction the first multi-GPU run is to be compared with.
        tails = sorted(e0.elapsed_time(e1) for e0, e1 in ts.tail_probe)
        out["exposed_tail_us"] = {"median": round(1e3 * tails[len(tails) // 2], 1), "max": round(1e3 * tails[-1], 1),
                                  "what": "main stream idle behind the late stream at the end of a step: last bucket's weight gradients"
                                          + (" + all-reduce" if ts.dist_active else "") + " + optimizer slice + re-pack"}
        ts.tail_probe = None
    if ts.dist_active:          # the exchange step alone: each gradient bucket's SUM all-reduce, HIP events on the stream it runs on
        from camradepth_amd.trainer import GradSync
        ar = {}
        for key in GradSync.ORDER:
            buf = ts.sync.bucket(key)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            dist.all_reduce(buf.clone(), op=dist.ReduceOp.SUM, group=ts.sync.group)      # warm-up
            tmp = buf.clone()
            torch.cuda.synchronize()
            e[0].record()
            for _ in range(5):
                dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=ts.sync.group)
            e[1].record()
            torch.cuda.synchronize()
            ar["+".join(key)] = {"bytes": buf.numel() * 4, "us": round(1e3 * e[0].elapsed_time(e[1]) / 5, 1)}
        out["allreduce_per_bucket"] = ar
    scale = (a.height * a.width) / (256 * 416)
    train_tflop_per_img = 3 * FWD_GFLOP[a.variant] * scale / 1e3
    out["mfma_frac_train_step"] = round(value / world * train_tflop_per_img / MFMA_BF16_PEAK_TFLOPS, 4)

    if rank == 0 and not a.no_roofline:
        agg = per_kernel_timing(ts)
        # the dominant kernel is a kernel TEMPLATE (all its tile instantiations together): k_conv3x3<*>, k_igemm<*>, ...
        # -- the 3x3 halo convolution is ONE family: the persistent kernel k_conv3x3p<...> and the two-workgroup kernel
        # k_conv3x3<...> that takes the small grids and ragged column tails, i.e. every rocprof row `k_conv3x3*`
        fam = {}
        for k, (n_, ms_, fl_) in agg.items():
            f = fam.setdefault(family_of(k), [0.0, 0.0, 0.0])
            f[0] += n_ * (1 + k.count("+")); f[1] += ms_; f[2] += fl_     # "a+b": one call, two device launches
        domf = max(fam, key=lambda k: fam[k][1])                          # the family the step spends most MFMA-kernel time in
        n_e, ms_e, fl_e = fam[domf]
        dom = domf + "*"
        # its roofline entry: every launch of the family replayed back to back from one HIP graph (sustained load, as in the step)
        n, fl, ms_tot = family_replay_timing(ts, domf)
        ach = fl / (ms_tot * 1e-3) / 1e12
        c2 = (a.batch, a.height, a.width, a.variant) == (8, 256, 416, "base")
        committed = trace_frac(dom, fl) if c2 else None
        # `frac` is THIS RUN's in-step-order figure (VERDICT / ADVICE r4): HIP events around every launch of the family, eager, in the
        # order of the step, on the activations of a real step.  The back-to-back replay of the family alone -- no dependent small
        # kernels in between -- is frac_isolated; the family's rows of the committed rocprofv3 trace of one replayed step
        # (profiles/, valid while the kernel sources match) is frac_committed_trace: a second opinion, not this run's measurement.
        frac_live = fl_e / (ms_e * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS
        traffic = pmc_traffic(dom) if c2 else None
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(frac_live * MFMA_BF16_PEAK_TFLOPS, 1),
                           "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(frac_live, 4),
                           "frac_source": "HIP events around every launch of the family, eager, in step order (this run)",
                           "frac_isolated": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                           "frac_committed_trace": committed,
                           "traffic": traffic,
                           "launches_per_step": n, "avg_launch_us": round(1e3 * ms_e / max(n_e, 1), 2),
                           "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3),
                           "method": "frac: per-launch HIP events in step order; frac_isolated: 10 replays of one HIP graph holding all launches of the family",
                           "isolated_replay_tflops": round(ach, 1), "eager_in_order_tflops": round(fl_e / (ms_e * 1e-3) / 1e12, 1)}
        fam_ops = [op for op in ts.plan.fwd + ts.plan.bwd if ts.plan.live(op) and op.meta is not None and family_of(op.meta["kernel"]) == domf]
        alg = sum(ts.plan.op_bytes(op) for op in fam_ops) / max(n, 1)
        out["roofline"]["algorithmic_bytes_per_launch"] = round(alg)
        if traffic:
            out["roofline"]["traffic_over_algorithmic"] = round(traffic / alg, 3)
        out["kernels"] = {k: {"launches": v[0], "ms_per_step": round(v[1], 3), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    if rank == 0:
        fb = floor_budget(ts.plan)
        out["floor_ms"], out["step_over_floor"], out["floor_budget"] = fb["floor_ms"], round(ms / fb["floor_ms"], 3), fb
    if rank == 0 and world == 1 and not a.no_excess and not a.no_graph and "exposed_tail_us" in out and a.update_interval == 1:
        ex = excess_attribution(ts, ms, out["exposed_tail_us"]["median"] * 1e-3)
        if ex is not None:
            out["excess_ms"] = ex
    if rank == 0 and world == 1 and not a.no_roofline:
        out["forward_only"] = forward_only(model, batch, a.batch, a.height, a.width, a.variant)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        small = a.height * a.width <= 256 * 416        # bounded: at larger frames only the batch-2-sized legs fit the time budget
        out["cpu_baseline"] = cpu_baseline(a.variant, a.batch if small else 1, a.height, a.width, timed_steps=5 if small else 2)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL's version banner sits in the C library's stdio buffer until then
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
