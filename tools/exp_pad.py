import sys, torch
sys.path.insert(0, ".")
from camradepth_amd.engine import Plan
from camradepth_amd.model import CamRaDepth
m = CamRaDepth(input_channels=7).cuda().train()
m._ensure_grad_views()
p = Plan(m, 8, 256, 416, True)
def fb(): p.forward(); p.backward()
fb(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): fb()
for _ in range(3): g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"ops {len(p.fwd) + len(p.bwd)}  fwd+bwd {e0.elapsed_time(e1) / 10:.3f} ms")
