"""Developer tool: N eval forwards of the base model at 8x7x256x416 (the program rocprofv3 --pmc wraps for the encoder-kernel
counters, tools/pmc_encoder.sh)."""
import sys
sys.path.insert(0, ".")
import torch
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
train = len(sys.argv) > 2 and sys.argv[2] == "train"
m = CamRaDepth(input_channels=7, seed=0).cuda().train(train)
b = synth.make_batch(8, 256, 416, seed=1234)
x = b["image"].cuda()
if train:
    from camradepth_amd import losses as hl
    for _ in range(n):
        out = m(x)
        loss, _ = hl.total_loss(out, {k: v.cuda() for k, v in b.items()}, False)
        loss.backward()
        m.zero_grad()
else:
    with torch.no_grad():
        for _ in range(n):
            m(x)
torch.cuda.synchronize()
