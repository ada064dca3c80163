"""Trains CamRaDepth (base) on the HIP path on synth.make_learnable_batch -- ground truth a smooth function of the input -- and
saves a checkpoint through camradepth_amd.checkpoint (the RMSE gate needs an operating point that resembles a
trained network; the reference initialisation gives an input-independent output and the golden weights an RMSE of 107 m).
Then evaluates on held-out seeds: RMSE of the HIP eval forward, of the CPU oracle in fp32 and in bf16 mode.

    python tools/train_synth_checkpoint.py [steps=3000] [out=gpurun_out/trained_synth.pth] [lr=3e-4]
The checkpoint is fp32 (88 MB): not committed; tests/test_gpu_trained.py trains its own (same recipe, fewer steps)."""
import math
import os
import sys
import time

sys.path.insert(0, ".")
import torch

from camradepth_amd import checkpoint, synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep, one_cycle


def train(model, steps, lr, B=8, H=256, W=416, pool=48, log=print):
    """`steps` graph-replayed training iterations over a pool of `pool` pre-generated batches (seeds 10000...)."""
    ts = TrainStep(model, B, H, W, lr=lr, schedule=one_cycle(steps + 8, lr))
    batches = [{k: v.cuda() for k, v in synth.make_learnable_batch(B, H, W, seed=10000 + i).items() if k != "dense_depth"} for i in range(pool)]
    t0 = time.time()
    for i in range(steps):
        ts.set_batch(batches[(i * 7) % pool])
        ts.step()
        if i % 250 == 0 or i == steps - 1:
            v = ts.losses()
            log(f"step {i:5d}: loss {v['loss']:.5f}  rmse(norm) {v['rmse']:.5f}  ({time.time() - t0:.0f} s)")
    return ts


def rmse_of(pred, gt):
    m = gt > 0
    return math.sqrt(float(((pred[m] - gt[m]) ** 2).mean()))


def evaluate(model, seeds=(777, 778), B=2, H=256, W=416, log=print):
    """Held-out batches: RMSE (normalised depth) of the HIP eval forward, the CPU oracle in fp32 and in bf16 mode, same weights."""
    from oracle import model as om
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.eval()
    rows = []
    for seed in seeds:
        b = synth.make_learnable_batch(B, H, W, seed=seed)
        with torch.no_grad():
            hip = model(b["image"].cuda())["depth"]["final_depth"].cpu()
            o32 = om.forward(sd, b["image"], model.cfg)["depth"]["final_depth"]
            o16 = om.forward(sd, b["image"], model.cfg, quant="bf16")["depth"]["final_depth"]
        r = {"seed": seed, "rmse_hip": rmse_of(hip, b["gt_full"]), "rmse_oracle_fp32": rmse_of(o32, b["gt_full"]),
             "rmse_oracle_bf16": rmse_of(o16, b["gt_full"]),
             "rel_l2_hip_vs_fp32": float((hip - o32).norm() / o32.norm()), "rel_l2_bf16_vs_fp32": float((o16 - o32).norm() / o32.norm())}
        rows.append(r)
        log(" ".join(f"{k}={v:.6f}" if isinstance(v, float) else f"{k}={v}" for k, v in r.items()))
    model.train()
    return rows


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/trained_synth.pth"
    lr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-4
    model = CamRaDepth(input_channels=7, seed=0).cuda().train()
    train(model, steps, lr)
    rows = evaluate(model)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    if os.environ.get("CRD_SAVE_BF16"):      # 44 MB: small enough to travel back from the GPU box (weights rounded to bf16 first)
        sd = {k: v.detach().cpu().to(torch.bfloat16) for k, v in model.state_dict().items()}
        torch.save({"state_dict": sd, "steps": [steps, 0]}, out)
    else:
        checkpoint.save_checkpoint(out, model, steps=(steps, 0))
    print("saved", out, os.path.getsize(out) // 2 ** 20, "MiB")
