"""The persistent encoder stage (crd_enc_stage_fwd, csrc/enc_stage.hip) against the per-launch kernels it replaces, tensor by
tensor and Block by Block (reference: Block.forward simplified_attention.py:141-145 and what it calls), and its reproducibility.

Both paths implement the same arithmetic with the same rounding points (bf16 GEMM operands / results, fp32 accumulation,
GroupNorm from fixed-point sums); they differ in the ORDER of fp32 partial sums, so tensors agree to a few bf16 ulps on the
first Block and drift apart at the rate the chaotic max-pool attention amplifies that; the bounds below are on
the FIRST block of each stage (tight) and on the stage output (loose), the arg-max table agrees except at near-ties.  The
persistent path is opt-in (engine.enc_persist_default; CRD_ENC_PERSIST=1 / 0 as the tests here set it); test_persistent_stage_vs_oracle below holds it to the
oracle directly."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from camradepth_amd import synth
from camradepth_amd.config import ModelConfig
from camradepth_amd.params import param_specs

# Round 5: the persistent stage is a parked DEVELOPER path (forward-only, at parity in time, an unexplained per-process slow mode under
# graph replay -- DESIGN section 4): these tests run only with CRD_DEV_SWITCHES=1 and no longer take ~2 minutes of every GPU test run.
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("CRD_DEV_SWITCHES") != "1", reason="persistent encoder stage: developer path (CRD_DEV_SWITCHES=1)")]


def build(cfg, sd, train):
    from camradepth_amd.model import CamRaDepth
    m = CamRaDepth(input_channels=cfg.input_channels, depths=cfg.depths)
    m.load_state_dict(sd)
    return m.cuda().train(train)


def _run(model, x, masks, persist):
    os.environ["CRD_ENC_PERSIST"] = "1" if persist else "0"
    try:
        out = model(x, masks=masks)
        plan = model._plans[model._plan_key(x)]
        torch.cuda.synchronize()
        for st in plan.enc_status:
            assert int(st.item()) == 0, "a workgroup of the persistent stage gave up waiting"
        return out, plan
    finally:
        os.environ.pop("CRD_ENC_PERSIST", None)


def _val(v):
    while not isinstance(v, torch.Tensor):      # engine.PM / engine._Lazy -> the tensor behind it
        v = v.t
    return v.detach().clone()


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(params=[1, 2], ids=["rows1", "rows2"])
def rows_per_wg(request, monkeypatch):
    """Both decompositions of the persistent kernel (one / two image rows per workgroup; the library would choose by B x H)."""
    from camradepth_amd import engine
    monkeypatch.setattr(engine, "ENC_ROWS_PER_WG", request.param)
    return request.param


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("B,H,W,depths", [(2, 256, 416, (1, 1, 2, 2)), (8, 256, 416, (1, 1, 3, 2)), (3, 64, 96, (1, 1, 2, 1)),
                                           (1, 128, 192, (1, 1, 1, 2))])
def test_persistent_stage_matches_per_launch_path(B, H, W, depths, train, rows_per_wg):
    cfg = dataclasses.replace(ModelConfig.variant("base"), depths=depths)
    sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
    model = build(cfg, sd, train)
    x = synth.make_batch(B, H, W, seed=11)["image"].cuda()
    masks = synth.make_masks(cfg, B, seed=4321) if train else None
    with torch.enable_grad() if train else torch.no_grad():
        out0, plan0 = _run(model, x, masks, persist=False)
        taps0 = {n: {k: _val(v) for k, v in d.items()} for n, d in plan0.enc_taps.items()}
        out1, plan1 = _run(model, x, masks, persist=True)
    assert plan1 is not plan0
    ops = [op.name for op in plan1.fwd]
    n_persist = ops.count("crd_enc_stage_fwd")
    assert n_persist >= 1, "no stage of this shape took the persistent path"
    stats_keys = ("st1", "ch1", "stk", "st2", "sth1")
    for name, d1 in plan1.enc_taps.items():
        stage = int(name.split("block")[1][0])
        if stage < 3:
            continue
        first = name.endswith(".0") and stage == 3         # (stage 4's first block already sees the drift of stage 3's blocks)
        for k, v in d1.items():
            if not train and k != "x2":
                continue                              # inference plans store the stage results only
            if k == "x2" and not (train or name.endswith(f".{depths[stage - 1] - 1}")):
                continue
            a, b = _val(v), taps0[name][k]
            if k == "idx":
                mism = float((a != b).float().mean())
                assert mism < (0.02 if first else 0.2), (name, k, mism)
            elif k == "sth2":
                # group totals: the per-launch path keeps them per 16-channel slab, the persistent kernel in the group's first slab
                g = a.shape[1] // (cfg.dims[stage - 1] // 16)
                sa = a.view(a.shape[0], -1, g, 2).sum(2).double()
                sb = b.view(b.shape[0], -1, g, 2).sum(2).double()
                assert rel(sa, sb) < (2e-3 if first else 0.1), (name, k, rel(sa, sb))
            elif k in stats_keys:
                assert rel(a, b) < (2e-3 if first else 0.1), (name, k, rel(a, b))
            else:
                tol = 8e-3 if first else 0.25
                assert rel(a, b) < tol, (name, k, rel(a, b))
    assert rel(out1["depth"]["final_depth"], out0["depth"]["final_depth"]) < 2e-2


def test_persistent_stage_is_bit_reproducible_and_graph_safe(monkeypatch, rows_per_wg):
    """Two eager runs and a graph replay of the full-depth stages give identical bits (fixed-order sums, epoch tags that survive
    replays without any re-initialisation)."""
    cfg = ModelConfig.variant("base")
    sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
    model = build(cfg, sd, False)
    x = synth.make_batch(8, 256, 416, seed=5)["image"].cuda()          # (B x H = 128 at stage 3: the library's own choice is one row)
    monkeypatch.setenv("CRD_ENC_PERSIST", "1")
    with torch.no_grad():
        a = model(x)["depth"]["final_depth"].clone()
        b = model(x)["depth"]["final_depth"].clone()
        plan = model._plans[model._plan_key(x)]
        assert [op.name for op in plan.fwd].count("crd_enc_stage_fwd") == 2
        from camradepth_amd.inference import InferenceGraph
        g = InferenceGraph(model, 8, 256, 416)
        c = g.run(x)["depth"]["final_depth"].clone()
        d = g.run(x)["depth"]["final_depth"].clone()
    torch.cuda.synchronize()
    for st in plan.enc_status:
        assert int(st.item()) == 0
    assert torch.equal(a, b) and torch.equal(c, d) and torch.equal(a, c)


def test_persistent_stage_vs_oracle(monkeypatch, rows_per_wg):
    """The whole model with stages 3-4 on the persistent kernel against the CPU oracle (bf16 mode), forward and every parameter
    gradient, shallow depths, train mode with injected masks (the bounds of tests/test_gpu_model.py::_shallow_vs_oracle)."""
    monkeypatch.setenv("CRD_ENC_PERSIST", "1")
    from tests.test_gpu_model import _shallow_vs_oracle
    _shallow_vs_oracle("base", "train", 2, 256, 416)
