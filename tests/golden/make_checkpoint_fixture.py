#!/usr/bin/env python3
"""Write tests/golden/ref_checkpoint_tiny.pth.gz + ref_checkpoint_tiny.npz with the REAL reference.

Runs only in the build container (needs /root/reference).  A small reference model (depths 1,1,1,1, dims 16) is wrapped
in nn.DataParallel exactly as the trainer does (src/main/runner.py:135-136), trained for two iterations with the
reference's own losses and its diffGradNorm, and saved with the dictionary of runner.py:369-371
({'state_dict', 'optimizer', 'lr', 'steps'}, torch.save) -- the file is data written by the reference, gzip-ed.
The npz holds what the next iteration does in the reference: the gradients it saw and the parameters after its
optimizer.step(), so that the build can resume from the file and must land on the same parameters.

The big decoder 3x3 weights are frozen (requires_grad=False) in this run, which also makes the optimizer state sparse in
the way the reference's `grad is None` rule produces (src/models/diffGradNorm.py:54-55); weights are drawn from the
shared seeded generator and truncated to 4 (frozen decoder weights: 0) mantissa bits so the archive stays small in the repository.
"""
import gzip
import importlib.util
import io
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
CFG = dict(depths=(1, 1, 1, 1), dims=(16, 16, 16, 16), heads=(1, 1, 1, 1))


def frozen(name):
    return name.startswith("depth_upsample.") and name.endswith(".model.0.weight")


def main():
    import numpy as np
    import torch
    import torch.nn as nn
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.install_shims()
    torch.manual_seed(0)
    tmp = tempfile.mkdtemp()
    sys.argv = ["x", "--split", f"{REF}/src/data/new_split.npy", "--model", "base", "--output_dir", tmp]
    sys.path.insert(0, f"{REF}/src")
    sys.path.insert(0, REPO)
    from models.CamRaDepth import CamRaDepth            # reference
    from models.diffGradNorm import diffGradNorm        # reference
    from utils.loss_funcs import MaskedSmoothL1Loss     # reference
    from camradepth_amd import synth

    model = CamRaDepth(input_channels=7, **CFG)
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3)
    for k, v in sd.items():          # low-entropy fp32: keep sign, exponent and 4 mantissa bits (frozen decoder weights: none)
        sd[k] = (v.view(torch.int32) & (~0x7FFFFF if frozen(k) else ~0x7FFFF)).view(torch.float32).clone()
    model.load_state_dict(sd, strict=True)
    for n, p in model.named_parameters():
        if frozen(n):
            p.requires_grad_(False)
    wrapped = nn.DataParallel(model)                    # runner.py:135-136 (keys get the 'module.' prefix)
    opt = diffGradNorm(wrapped.parameters(), lr=1e-3)
    crit = MaskedSmoothL1Loss()
    model.eval()                                        # deterministic: Dropout2d / DropPath off for the fixture run

    def iteration(seed):
        b = synth.make_batch(1, 32, 32, seed=seed)
        out = model(b["image"])
        inter = out["depth"]["intermediate_depths"]
        loss = (crit(out["depth"]["final_depth"], b["gt_full"]) + crit(inter[-1].squeeze(1), b["gt_half"].squeeze(1))
                + crit(inter[-2].squeeze(1), b["gt_quarter"].squeeze(1))) / 3.4
        opt.zero_grad(set_to_none=True)
        loss.backward()
        return float(loss)

    for it in range(2):
        iteration(40 + it)
        opt.step()
    state = {"state_dict": wrapped.to("cpu").state_dict(), "optimizer": opt.state_dict(), "lr": opt.param_groups[0]["lr"],
             "steps": [2, 0]}
    buf = io.BytesIO()
    torch.save(state, buf)
    with gzip.GzipFile(os.path.join(HERE, "ref_checkpoint_tiny.pth.gz"), "wb", compresslevel=9, mtime=0) as f:
        f.write(buf.getvalue())
    # the next iteration in the reference
    iteration(42)
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    out = {"names": np.array(names)}
    for n, p in model.named_parameters():
        if n in grads:
            out["grad:" + n] = grads[n].numpy()
            out["after:" + n] = p.detach().numpy().copy()
    out["exp_grad_norm"] = np.array([float(opt.state[p]["exp_grad_norm"]) for n, p in model.named_parameters() if n in grads])
    out["frozen"] = np.array([n for n, _ in model.named_parameters() if frozen(n)])
    np.savez_compressed(os.path.join(HERE, "ref_checkpoint_tiny.npz"), **out)
    print("checkpoint bytes", len(buf.getvalue()), "gz", os.path.getsize(os.path.join(HERE, "ref_checkpoint_tiny.pth.gz")),
          "npz", os.path.getsize(os.path.join(HERE, "ref_checkpoint_tiny.npz")), "trainable tensors", len(names))


if __name__ == "__main__":
    main()
