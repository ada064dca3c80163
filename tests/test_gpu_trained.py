"""The north-star accuracy gate at a TRAINED operating point (BASELINE.json: depth RMSE within 1e-3 of the reference; the reference's
RMSE: src/main/runner.py:208, src/utils/loss_funcs.py:40-46).  VERDICT r3: at the reference initialisation the output is almost
input-independent (the gate is met trivially), with the golden weights the RMSE is 107 m (non-physical).  Here the network is first
trained on the HIP path on a learnable synthetic task (camradepth_amd.synth.make_learnable_batch: ground truth = a smooth function of the
input; recipe of tools/train_synth_checkpoint.py), saved and re-loaded through camradepth_amd.checkpoint, and then held-out batches at
256 x 416 are evaluated by the HIP eval forward and by the CPU oracle (fp32 = the reference's arithmetic, pinned in
tests/test_oracle_golden.py) with the SAME weights."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
STEPS = int(os.environ.get("CRD_TRAINED_STEPS", "1500"))
TRAINED_BLOCK_OUT, TRAINED_BLOCK_UPD = 3e-3, 6.5e-3        # 2 x the round-5 measurement: worst output 1.5e-3 (block2.0), worst update 3.2e-3 (block3.2), median 2.5e-3


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    from camradepth_amd import checkpoint
    from camradepth_amd.model import CamRaDepth
    from tools.train_synth_checkpoint import train
    model = CamRaDepth(input_channels=7, seed=0).cuda().train()
    log = []
    ts = train(model, STEPS, 3e-4, log=log.append)
    path = str(tmp_path_factory.mktemp("ckpt") / "trained.pth")
    checkpoint.save_checkpoint(path, model, steps=(STEPS, 0))
    fresh = CamRaDepth(input_channels=7, seed=1).cuda()
    missing, mismatched, steps = checkpoint.load_checkpoint(path, fresh)
    assert not missing and not mismatched and steps[0] == STEPS
    assert torch.equal(fresh.flat, model.flat)
    return fresh, log


def test_training_on_the_learnable_task_learns(trained):
    _, log = trained
    first = float(log[0].split("rmse(norm)")[1].split()[0])
    last = float(log[-1].split("rmse(norm)")[1].split()[0])
    assert last < 0.5 * first, log          # the RMSE on the training stream at least halves (init: ~0.29 = the spread of the target)


def test_rmse_within_1e3_of_the_fp32_oracle_at_a_trained_operating_point(trained):
    from tools.train_synth_checkpoint import evaluate
    model, _ = trained
    rows = evaluate(model, seeds=(777, 778), B=2)
    for r in rows:
        # a trained-like operating point: clearly better than predicting the mean of the target (RMSE 0.29 normalised = 29 m)
        assert r["rmse_oracle_fp32"] < 0.2, r
        assert abs(r["rmse_hip"] - r["rmse_oracle_fp32"]) < 1e-3, r                  # THE north-star gate
        assert abs(r["rmse_oracle_bf16"] - r["rmse_oracle_fp32"]) < 1e-3, r          # (the yardstick: the oracle's own bf16 mode)
        assert r["rel_l2_hip_vs_fp32"] < 2.5 * r["rel_l2_bf16_vs_fp32"] + 2e-3, r


def test_every_block_at_the_trained_weights_on_the_oracles_input(trained):
    """the trained operating point pins the BLOCKS as the golden weights do (tests/test_gpu_blocks.py): each of the
    34 encoder Blocks alone on the oracle's (bf16 mode) input of that block at 2 x 256 x 416, held-out batch, trained weights."""
    import numpy as np
    from camradepth_amd import synth
    from camradepth_amd.config import ModelConfig
    from oracle import model as om
    from tests.test_gpu_blocks import block_errors
    model, _ = trained
    cfg = ModelConfig.variant("base")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = synth.make_learnable_batch(2, 256, 416, seed=779)["image"]
    taps = {}
    with torch.no_grad():
        om.forward(sd, x, cfg, quant="bf16", taps=taps)
    model.eval()
    with torch.no_grad():
        model(x.cuda())
    plan = model._plans[model._plan_key(x.cuda())]
    out_err, upd_err = block_errors(plan, taps)
    model.train()
    worst_out, worst_upd = max(out_err.items(), key=lambda kv: kv[1]), max(upd_err.items(), key=lambda kv: kv[1])
    med = float(np.median(list(upd_err.values())))
    print(f"trained weights, teacher-forced Blocks: worst output rel-L2 {worst_out}, worst UPDATE {worst_upd}, median update {med:.2e}")
    assert len(out_err) == 34
    assert worst_out[1] < TRAINED_BLOCK_OUT and worst_upd[1] < TRAINED_BLOCK_UPD, (worst_out, worst_upd)


def test_fp8_mode_at_the_trained_operating_point(trained):
    """VERDICT r4 item 1 'done' criterion: |RMSE - RMSE_fp32| < 1e-3 at the trained operating point IN FP8 MODE -- the e4m3 inference path
    (ConvLayers of the full-resolution decoder stage at this batch size) on held-out batches against the fp32 oracle -- and config 5 as a
    TRAINING mode from there: 60 more steps with e4m3 forward + data gradients (delayed scaling inside the graphs) keep the loss finite
    and leave the held-out RMSE where the bf16 run had it."""
    from camradepth_amd import synth
    from camradepth_amd.trainer import TrainStep
    from oracle import model as om
    from tools.train_synth_checkpoint import rmse_of
    model, _ = trained
    saved = model.flat.clone()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    held = [synth.make_learnable_batch(2, 256, 416, seed=s) for s in (777, 778)]
    model.eval()
    scales = model.calibrate_fp8(synth.make_learnable_batch(2, 256, 416, seed=20001)["image"].cuda())
    assert "depth_upsample.4" in scales
    rows = []
    for b in held:
        with torch.no_grad():
            out8 = model(b["image"].cuda())["depth"]["final_depth"].cpu()
            o32 = om.forward(sd, b["image"], model.cfg)["depth"]["final_depth"]
        plan = model._plans[model._plan_key(b["image"].cuda())]
        assert sum(op.name == "crd_conv3x3_fp8" for op in plan.fwd) >= 3
        rows.append((rmse_of(out8, b["gt_full"]), rmse_of(o32, b["gt_full"])))
    print("fp8 inference at the trained operating point: RMSE (e4m3 path, fp32 oracle) per held-out batch:", rows)
    for r8, r32 in rows:
        assert abs(r8 - r32) < 1e-3, (r8, r32)                     # the north-star gate, in fp8 mode
    # -- config 5 as a training mode from this point, beside the SAME 60 steps in bf16 from the same weights (the control: 60 steps at
    #    lr 1e-4 with dropout move the held-out RMSE of a 2-sample batch by up to +-1.5e-3 whatever the arithmetic -- three builds of round
    #    6 measured e4m3-after minus before = -6.6e-4 / +7e-5, -6.8e-4 / +7e-5 and, after one encoder launch changed, +1.3e-3 on seed 778)
    batches = [{k: v.cuda() for k, v in synth.make_learnable_batch(8, 256, 416, seed=10000 + i).items() if k != "dense_depth"} for i in range(12)]

    def sixty_steps(fp8):
        with torch.no_grad():
            model.flat.copy_(saved)
        model.mark_params_changed()
        model.train()
        model.calibrate_fp8(synth.make_learnable_batch(8, 256, 416, seed=20002)["image"].cuda(), train=True, grads=True) if fp8 else model.calibrate_fp8(None)
        ts = TrainStep(model, 8, 256, 416, lr=1e-4)
        assert len(ts.plan.fp8_grad_layers) == (2 if fp8 else 0)
        first = last = None
        for i in range(60):
            ts.set_batch(batches[i % 12])
            ts.step()
            if i in (0, 59):
                v = ts.losses()
                first, last = (v if i == 0 else first), v
        # (training RMSE of ONE batch each -- different batches at steps 0 and 59, dropout on: +-3e-3 between batches; a regression from the
        # e4m3 gradients would show in the held-out gates below)
        assert (not fp8 or not ts.plan.fp8_jit) and math.isfinite(last["loss"]) and last["rmse"] < first["rmse"] + 6e-3, (first, last)
        model.calibrate_fp8(None)
        model.eval()
        res = []
        for b in held:
            with torch.no_grad():
                res.append(rmse_of(model(b["image"].cuda())["depth"]["final_depth"].cpu(), b["gt_full"]))
        return first, last, res

    first, last, after = sixty_steps(True)
    _, _, ctrl = sixty_steps(False)
    print(f"60 steps with e4m3 forward + data gradients: training rmse {first['rmse']:.5f} -> {last['rmse']:.5f}; held-out RMSE (bf16 eval) "
          f"before {[round(r[1], 5) for r in rows]} (fp32 oracle) after {[round(a, 5) for a in after]}; the same 60 steps in bf16: {[round(c, 5) for c in ctrl]}")
    for a, c, (r8, r32) in zip(after, ctrl, rows):
        # ADVICE r5 (the old gate allowed 1.8x the fp32 RMSE): e4m3 training ends where bf16 training ends, and neither walks away from the
        # trained point -- bounds = the measured run-to-run spread of 60 dropout steps (above), not a multiple of the RMSE
        assert abs(a - c) < 2.5e-3, (a, c)
        assert a < r32 + 3e-3 and c < r32 + 3e-3, (a, c, r32)
    with torch.no_grad():
        model.flat.copy_(saved)                                    # leave the fixture as it was found
    model.mark_params_changed()
    model.train()
