"""Forward time of encoder stages 1-4 (graph-replayed slices of the plan's forward op list), persistent stage kernel
(CRD_ENC_PERSIST=1, stages 3-4) against the per-launch path (=0), eval and train plans.
Usage: python tools/prof_enc_stage.py [B]"""
import os
os.environ.setdefault("CRD_DEV_SWITCHES", "1")      # the persistent stage is a developer path since round 5
import sys

import torch

from camradepth_amd import synth
from camradepth_amd.config import ModelConfig
from camradepth_amd.model import CamRaDepth
from camradepth_amd.params import param_specs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = ModelConfig.variant("base")
sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
x = synth.make_batch(B, 256, 416, seed=5)["image"].cuda()


def time_slice(plan, a, b, reps=20):
    ops = plan.fwd[a:b]
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        plan.zf_arena.zero_()
        plan.run_ops(ops)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            plan.run_ops(ops)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, len(ops)


from camradepth_amd import engine

for train in (False, True):
    for persist, rpw in (("0", 0), ("1", 2), ("1", 1)):
        os.environ["CRD_ENC_PERSIST"] = persist
        engine.ENC_ROWS_PER_WG = rpw
        m = CamRaDepth(input_channels=cfg.input_channels, depths=cfg.depths)
        m.load_state_dict(sd)
        m = m.cuda().train(train)
        with torch.enable_grad() if train else torch.no_grad():
            m(x)
        plan = m._plans[m._plan_key(x)]
        plan.training_masks_fixed = True
        marks = dict(plan.fwd_marks)
        names = ["enc0", "enc1", "enc2", "enc3", "dec"]
        row = []
        for i in range(4):
            us, n = time_slice(plan, marks[names[i]], marks[names[i + 1]])
            row.append(f"stage{i + 1} {us:8.1f} us ({n:3d} launches)")
        print(f"train={int(train)} persist={persist} rows/wg={rpw}  " + "  ".join(row), flush=True)
        for stt in plan.enc_status:
            assert int(stt.item()) == 0, "persistent stage timed out"

# --- per-phase stamps (a library built with -DCRD_ENC_PROF and loaded through CRD_LIB: tools/prof_enc_stage.sh)
import ctypes
from camradepth_amd import lib as L
lib = L.load()
if hasattr(lib, "crd_dbg_enc_prof"):
    names = ["(idle)", "E0 gather+tables", "xn", "q/sr(k) gemm + u", "E1 gather (+k gemm)", "scores + S", "x1", "E2 + tables", "xn2", "fc1 gemm",
             "h1 store + publish", "E3 + tables", "dwconv (rest)", "E4 + tables", "h3 (gelu)", "fc2 gemm", "x2 store + E0 publish",
             "  L2 warm-up issue", "  E0 gather", "  dw r0: stage halo + norm", "  dw r0: stencil", "  dw r0: reduce + barrier",
             "  dw r1: stage halo + norm", "  dw r1: stencil", "  dw r1: reduce + barrier", "  q gemm (wave 0)", "  sr / k gemm (wave 0)",
             "  E1 gather + attn.norm"]
    os.environ["CRD_ENC_PERSIST"] = "1"
    engine.ENC_ROWS_PER_WG = int(os.environ.get("PROF_ROWS", "0"))
    for train in (False, True):
        m = CamRaDepth(input_channels=cfg.input_channels, depths=cfg.depths)
        m.load_state_dict(sd)
        m = m.cuda().train(train)
        with torch.enable_grad() if train else torch.no_grad():
            m(x)
        plan = m._plans[m._plan_key(x)]
        plan.training_masks_fixed = True
        ops = [op for op in plan.fwd if op.name == "crd_enc_stage_fwd"]
        for si, op in enumerate(ops):
            buf = (ctypes.c_ulonglong * 32)()
            torch.cuda.synchronize()
            lib.crd_dbg_enc_prof(buf, 1)
            reps = 5
            for _ in range(reps):
                plan.run_ops([op])
            torch.cuda.synchronize()
            lib.crd_dbg_enc_prof(buf, 1)
            nb = cfg.depths[2 + si]
            tot = sum(buf[1:28]) / 100.0 / reps / nb
            print(f"train={int(train)} stage {3 + si}: {tot:7.1f} us per block (workgroup 0), phases:")
            for i in range(1, 28):
                print(f"    {names[i]:28s} {buf[i] / 100.0 / reps / nb:7.2f} us")
