"""Round 6 experiment: the LATE stream (weight gradients, un-packing, optimizer slices, re-pack) on a HIP stream with a compute-unit mask
(hipExtStreamCreateWithCUMask) -- does keeping its kernels off some of the chip's XCDs take the interference off the dependency chain?
(bench.excess_attribution: 2.2 of the 17.4 ms.)  Usage: python tools/exp_cumask.py  -> one line per mask."""
import ctypes
import sys

import torch

from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep, one_cycle

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def run(name, bits, w3=None):
    model = CamRaDepth(input_channels=7, seed=0).cuda().train()
    if w3 is not None:
        model.w3_total_wgs = w3
    ts = TrainStep(model, 8, 256, 416, lr=6e-5, schedule=one_cycle(400, 6e-5))
    if w3 is not None:
        model.w3_total_wgs = w3
    if bits is not None:
        ts.late_stream_factory = lambda: masked_stream(bits)
    ts.set_batch({k: v.cuda() for k, v in synth.make_batch(8, 256, 416, seed=1234).items()})
    for _ in range(10):
        ts.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        ts.step()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / 40:.3f} ms per step", flush=True)
    del ts, model
    torch.cuda.empty_cache()


ALL = (1 << 256) - 1
interleaved = lambda xccs: sum(1 << i for i in range(256) if (i % 8) in xccs)      # bit i -> XCC i % 8 (if the mask is dealt round-robin)
contiguous = lambda lo, hi: sum(1 << i for i in range(lo, hi))                      # bit i -> XCC i // 32 (if it is laid out XCC by XCC)
run("unmasked (torch stream)", None)
run("all 256 CUs (masked stream, full mask)", ALL)
for n in (6, 5, 4, 3):
    run(f"late stream on {n} XCCs, mask dealt round-robin", interleaved(set(range(n))))
for n in (192, 160, 128, 96):
    run(f"late stream on the first {n} mask bits", contiguous(0, n))
run("unmasked (torch stream)", None)
