"""CPU: the execution plan (camradepth_amd.engine.Plan) is pure recording -- buffers, descriptors, op lists -- so it can be BUILT
without a GPU (nothing is launched).  This catches host-logic errors before a GPU box is spent on them, and pins the host-side
bookkeeping the measurement rests on: every recorded launch carries its algorithmic HBM bytes (bench.floor_budget, tools/floor_table.py),
the phase marks partition the forward list, the weight-gradient launches are all on the late stream."""
import pytest
import torch

import bench
from camradepth_amd.engine import LATE, Plan
from camradepth_amd.model import CamRaDepth


def _plan(train=True, **variant):
    m = CamRaDepth(input_channels=7, depths=(1, 1, 1, 1), **variant)
    m.train(train)
    if train:
        m._ensure_grad_views()
    return m, Plan(m, 2, 64, 96, train)


@pytest.mark.parametrize("variant", [{}, {"supervised_seg": True}, {"supervised_seg": True, "unsupervised_seg": True}])
def test_plan_builds_without_a_gpu_and_every_launch_has_bytes(variant):
    m, p = _plan(True, **variant)
    ops = [op for op in p.fwd + p.bwd if p.live(op)]
    assert len(p.fwd) > 85 and len(p.bwd) > 120
    zero = sorted({op.name for op in ops if p.op_bytes(op) <= 0})
    assert not zero, f"launches without algorithmic bytes: {zero}"
    # weight gradients never sit on the dependency chain
    for op in p.bwd:
        if op.fn is not None and op.name in ("crd_conv_wgrad", "crd_conv_wgrad_grouped", "crd_dwconv3x3_wgrad", "crd_head_conv2_wgrad"):
            assert op.stream == LATE, op.name
    # the first op of every phase mark is what the builder recorded there (marks survive the slice-copy inserts)
    names = [n for n, _ in p.fwd_marks]
    assert names == ["enc0", "enc1", "enc2", "enc3", "dec"]
    for n, i in p.fwd_marks[1:4]:
        assert p.fwd[i].name == "crd_conv_igemm" and "k3 s2" in p.fwd[i].meta["shape"], (n, p.fwd[i].name)     # the stage's patch embed


def test_floor_budget_adds_up():
    m, p = _plan(True)
    fb = bench.floor_budget(p)
    ph = fb["phases"]
    assert set(ph) == {"fwd:enc0", "fwd:enc1", "fwd:enc2", "fwd:enc3", "fwd:dec", "bwd:dec", "bwd:enc3", "bwd:enc2", "bwd:enc1", "bwd:enc0",
                       "late:dec", "late:enc3", "late:enc2", "late:enc1", "late:enc0"}
    chain = sum(v["floor_ms"] for k, v in ph.items() if not k.startswith("late:"))
    assert abs(chain - fb["floor_ms"]) < 1e-2
    n_ops = sum(1 + (op.meta or {}).get("kernel", "").count("+") for op in p.fwd + p.bwd if p.live(op))
    assert sum(v["launches"] for v in ph.values()) == n_ops
    for k, v in ph.items():
        assert v["floor_ms"] >= max(v["mfma_ms"], v["byte_ms"], v["dep_ms"]) - 1e-3 and v["floor_ms"] <= v["mfma_ms"] + v["byte_ms"] + v["dep_ms"] + 1e-3, k
    # the decoder carries the FLOPs, the encoder the launches
    assert ph["fwd:dec"]["gflop"] > 5 * sum(ph[f"fwd:enc{s}"]["gflop"] for s in range(4))


def test_eval_plan_records_no_backward_only_tensors():
    _, pt = _plan(True)
    m, pe = _plan(False)
    assert len(pe.fwd) <= len(pt.fwd)
    assert pe.training is False


def test_fp8_gradient_plan_records_both_scaling_variants():
    """Config 5 as a training plan (e4m3 forward + data gradients): the just-in-time variant (scale update + re-quantisation behind the
    GroupNorm backward of every e4m3 layer) and the delayed variant (one update at the head of the backward pass) are both recorded;
    plan.fp8_jit selects -- what TrainStep flips after its calibration iteration."""
    m = CamRaDepth(input_channels=7, depths=(1, 1, 1, 1))
    m.train(True)
    m._ensure_grad_views()
    m.__dict__["fp8_scales"] = {"depth_upsample.3": 0.01, "depth_upsample.4": 0.01}
    m.__dict__["fp8_train"] = True
    m.__dict__["fp8_grad"] = True
    p = Plan(m, 8, 128, 192, True)                  # 8 x 6 x 8 = 384 tiles at full resolution: that stage takes the fp8 route, 96 at half do not
    assert p.fp8_grad_layers == ["depth_upsample.4"]          # round 6: one scale per STAGE, all three write-once data gradients in e4m3
    assert sum(op.name == "crd_conv3x3_fp8" for op in p.fwd) == 3
    jit = [op.name for op in p.bwd if p.live(op)]
    p.fp8_jit = False
    delayed = [op.name for op in p.bwd if p.live(op)]
    assert jit.count("crd_conv3x3_fp8_dgrad") == delayed.count("crd_conv3x3_fp8_dgrad") == 3
    assert jit.count("crd_quant_fp8_dev") == 3 and delayed.count("crd_quant_fp8_dev") == 0
    assert jit.count("crd_fp8_scale_update") == 3 and delayed.count("crd_fp8_scale_update") == 1
    assert delayed[0] == "crd_fp8_scale_update" and jit[0] != "crd_fp8_scale_update"      # delayed: at the head of the backward pass
    i = jit.index("crd_gn_bwd_apply_fp8")
    assert jit[i + 1:i + 3] == ["crd_fp8_scale_update", "crd_quant_fp8_dev"]              # just-in-time: right behind the layer's GroupNorm backward
    assert bench.floor_budget(p)["floor_ms"] > 0


def test_round6_plan_structure():
    """The launch structure round 6 built, checked without a GPU: the decoder's data gradients are write-once (no 3x3 launch of the decoder's
    backward accumulates into the concat gradient: three K-concatenated launches per stage), every encoder Block folds Mlp.norm1's and
    attn.norm's backward apply into the consuming GEMM (crd_gn_bwd_conv), stage 3 runs q and sr as one launch (crd_gn_conv2), and the
    K-concatenated weight matrices are fed by extra pack-table entries."""
    m, p = _plan(True)
    dec = [op for (tag, a, b) in p.bwd_segments if tag == "dec" for op in p.bwd[a:b]]
    kcat = [op for op in dec if op.name == "crd_conv_igemm" and "dgrad-kcat" in op.meta["shape"]]
    assert len(kcat) >= 15                                              # 5 stages x 3 launches (+ ragged column splits)
    for op in dec:
        if op.name == "crd_conv_igemm" and " k3 " in op.meta["shape"] and "dgrad" in op.meta["shape"] and "Cin32" not in op.meta["shape"]:
            assert "dgrad-kcat" in op.meta["shape"], op.meta["shape"]   # (the heads' 32 -> 128 data gradients still accumulate into d(stage))
    assert sum(op.name == "crd_gn_bwd_conv" for op in p.bwd) == 3 + 3   # fc1 behind Mlp.norm1 in the Blocks of stages 1-3 (stage 4 keeps the two launches), the sr scatter at stages 1-3
    assert sum(op.name == "crd_gn_conv2" for op in p.fwd) == 1          # stage 3
    assert len(p.kcat_entries) == 5 * 6                                 # per stage: layer 2 -> WA, WB, WC; layer 1 -> WB, WC; layer 0 -> WC
    # what is left of crd_gn_bwd_apply in a Block: Mlp.norm2 (+ GELU) and Block.norm1; Block.norm2's rides in crd_attn_out_bwd_gn
    enc = [op.name for (tag, a, b) in p.bwd_segments if tag != "dec" for op in p.bwd[a:b]]
    assert enc.count("crd_gn_bwd_apply") == 4 * 2 + 1 + 4               # 2 per Block (+ Mlp.norm1's at stage 4) + the four patch embeds' norms
