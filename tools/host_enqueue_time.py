import sys, time
sys.path.insert(0, ".")
import torch
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep, one_cycle
m = CamRaDepth(input_channels=7, seed=0).cuda().train()
ts = TrainStep(m, 8, 256, 416, lr=6e-5, schedule=one_cycle(200, 6e-5))
b = synth.make_batch(8, 256, 416, seed=1234)
ts.set_batch({k: v.cuda() for k, v in b.items()})
for _ in range(5): ts.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): ts.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 30:.3f} ms/step, total {1e3 * (t2 - t0) / 30:.3f} ms/step")
