"""Developer check: the graph step (two-stream late mode) overfits one fixed synthetic batch -- the loss has to fall."""
import sys
sys.path.insert(0, ".")
import torch
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep, one_cycle
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
model = CamRaDepth(input_channels=7, seed=0).cuda().train()
ts = TrainStep(model, 4, 128, 224, lr=3e-4, schedule=one_cycle(steps + 8, 3e-4))
batch = synth.make_batch(4, 128, 224, seed=7)
ts.set_batch({k: v.cuda() for k, v in batch.items()})
for i in range(steps):
    ts.step()
    if i % 25 == 0 or i == steps - 1:
        l = ts.losses()
        print(f"step {i:4d}: loss {l['loss']:.5f}  rmse {l['rmse']:.5f}", flush=True)
