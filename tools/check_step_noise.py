"""Run-to-run and eager-vs-graph differences of one optimizer step on the shallow model (developer aid for test tolerances)."""
import dataclasses, sys, torch
sys.path.insert(0, ".")
from camradepth_amd import losses as hl, synth
from camradepth_amd.config import ModelConfig
from camradepth_amd.optim import diffGradNorm
from camradepth_amd.params import param_specs
from camradepth_amd.trainer import TrainStep
from tests.test_gpu_model import build, rel

cfg = dataclasses.replace(ModelConfig.variant("supervised_seg"), depths=(1, 1, 1, 1))
sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
batch = {k: v.cuda() for k, v in synth.make_batch(2, 64, 96, seed=9).items()}
masks = synth.make_masks(cfg, 2, seed=1)


def eager():
    m1 = build(cfg, sd, train=True)
    opt = diffGradNorm(m1.parameters(), lr=1e-3)
    out = m1(batch["image"], masks=masks)
    loss, _ = hl.total_loss(out, batch, True)
    opt.zero_grad(); loss.backward()
    g = m1.flat_grad.clone(); opt.step()
    return g, m1.flat.clone()


def graph():
    m2 = build(cfg, sd, train=True)
    ts = TrainStep(m2, 2, 64, 96, lr=1e-3, use_graph=True)
    ts.set_batch(batch)
    ts.plan.training_masks_fixed = True
    ts.plan.dp_masks.copy_(torch.stack([t.cuda() for t in masks["drop_path"]]))
    ts.plan.d2_masks.copy_(torch.stack([t.cuda() for t in masks["dropout2d"]]))
    ts.step(); torch.cuda.synchronize()
    return m2.flat_grad.clone(), m2.flat.clone()


a, b, c, d = eager(), eager(), graph(), graph()
for n, (u, v) in {"eager-eager": (a, b), "eager-graph": (a, c), "graph-graph": (c, d)}.items():
    flips = float(((u[0] * v[0]) < 0).float().mean())
    print(f"{n}: grad rel {rel(u[0], v[0]):.3e}  param rel {rel(u[1], v[1]):.3e}  sign flips {flips:.3e}")
