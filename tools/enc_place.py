"""Developer tool (a -DCRD_ENC_PROF library through CRD_LIB): time of the persistent stage-3 launch in THIS process and where its
workgroups ran (XCC id per workgroup) -- run it several times: the launch has a fast and a slow mode between processes."""
import ctypes
import os
os.environ.setdefault("CRD_DEV_SWITCHES", "1")      # the persistent stage is a developer path since round 5
import torch
from camradepth_amd import synth, engine, lib as L
from camradepth_amd.config import ModelConfig
from camradepth_amd.model import CamRaDepth
from camradepth_amd.params import param_specs

os.environ["CRD_ENC_PERSIST"] = "1"
cfg = ModelConfig.variant("base")
sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
x = synth.make_batch(8, 256, 416, seed=5)["image"].cuda()
m = CamRaDepth(input_channels=cfg.input_channels, depths=cfg.depths)
m.load_state_dict(sd)
m = m.cuda().eval()
with torch.no_grad():
    m(x)
plan = m._plans[m._plan_key(x)]
ops = [op for op in plan.fwd if op.name == "crd_enc_stage_fwd"]
lib = L.load()
for si, op in enumerate(ops):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan.run_ops([op])
    e0.record()
    for _ in range(5):
        plan.run_ops([op])
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_uint * 512)()
    lib.crd_dbg_enc_place(buf)
    n = 128 if si == 0 else 64
    xcc = [buf[i] >> 16 for i in range(n)]
    cu = [buf[i] & 0xffff for i in range(n)]
    per_set = {}
    for i in range(n):
        per_set.setdefault(i % 8, set()).add(xcc[i])
    print(f"stage {3 + si}: {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us; XCCs per sample set: {[sorted(v) for v in per_set.values()]}; "
          f"distinct (xcc, hw_id): {len(set(zip(xcc, cu)))} of {n}")

# the whole forward as the inference graph (what bench.py --inference replays): the slow mode shows up here
from camradepth_amd.inference import InferenceGraph
with torch.no_grad():
    g = InferenceGraph(m, 8, 256, 416)
    g.run(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.run(x, clone=False)
    e1.record()
    torch.cuda.synchronize()
buf = (ctypes.c_uint * 512)()
lib.crd_dbg_enc_place(buf)
xcc = [buf[i] >> 16 for i in range(64)]
per_set = {}
for i in range(64):
    per_set.setdefault(i % 8, set()).add(xcc[i])
print(f"graph forward: {e0.elapsed_time(e1) / 20:7.3f} ms; last persistent launch (stage 4) XCCs per set: {[sorted(v) for v in per_set.values()]}")
