import os, sys, time, torch
sys.path.insert(0, '.')
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
mp = int(os.environ.get("CRD_MAIN_PRIO", "99"))
main = torch.cuda.Stream(priority=mp) if mp != 99 else torch.cuda.current_stream()
with torch.cuda.stream(main):
    m = CamRaDepth(input_channels=7).cuda().train()
    ts = TrainStep(m, 8, 256, 416, lr=6e-5)
    ts.start_epoch()
    ts.set_batch({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth.make_batch(8, 256, 416, seed=1).items()})
    for _ in range(5): ts.step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(40): ts.step()
    torch.cuda.synchronize()
    print("main prio", mp, "late prio", os.environ.get("CRD_LATE_PRIO", "0"), "main stream prio", main.priority, "late", ts.late_stream.priority, "%.3f ms" % ((time.time() - t0) / 40 * 1e3))
