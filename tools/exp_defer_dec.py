"""Round 6 experiment: what would deferring the DECODER bucket's late work (weight gradients, un-pack, optimizer slice, re-pack) to the head of
the NEXT step -- beside its forward pass instead of beside the encoder's backward -- buy?  Timing only (the replays below violate the
data dependencies of a real step): (a) the step as captured; (b) the decoder's late graph replayed beside the first main graph (forward +
loss + decoder backward) of the next step, the encoder buckets' late graphs where they are."""
import torch
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep, one_cycle

B = 8
model = CamRaDepth(input_channels=7, seed=0).cuda().train()
ts = TrainStep(model, B, 256, 416, lr=6e-5, schedule=one_cycle(400, 6e-5))
ts.set_batch({k: v.cuda() for k, v in synth.make_batch(B, 256, 416, seed=1234).items()})
for _ in range(10):
    ts.step()
torch.cuda.synchronize()
_, g0, chain, _ = ts.graphs[(True, True)][0][0]
main, late = torch.cuda.current_stream(), ts.late_stream


def as_captured():
    for gm, gl, key, _ in chain:
        gm.replay()
        late.wait_stream(main)
        with torch.cuda.stream(late):
            gl.replay()
    main.wait_stream(late)


def deferred():
    (gm0, gl0, _, _), rest = chain[0], chain[1:]
    with torch.cuda.stream(late):
        gl0.replay()                       # the PREVIOUS step's decoder bucket, beside this step's forward / decoder backward
    gm0.replay()
    main.wait_stream(late)                 # (a real implementation waits in front of the decoder's forward)
    for gm, gl, key, _ in rest:
        gm.replay()
        late.wait_stream(main)
        with torch.cuda.stream(late):
            gl.replay()
    main.wait_stream(late)


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for rep in range(3):
    print(f"as captured {timed(as_captured):.3f} ms   decoder bucket deferred to the next step's first graph {timed(deferred):.3f} ms", flush=True)
