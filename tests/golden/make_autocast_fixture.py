#!/usr/bin/env python3
"""A reference-derived bf16 yardstick: the REAL reference (/root/reference, CPU) under
torch.autocast("cpu", dtype=torch.bfloat16) -- the only autocast this container can run -- at 64x96 and 256x416 with the golden
weights: final depth, the loss terms and every parameter's gradient norm.  Stores outputs only -> tests/golden/ref_autocast_bf16.npz.

What this fixture is and is not.  The reference trains under CUDA autocast (src/main/runner.py:191, fp16 there; bf16 is this
build's choice).  CPU autocast shares the rounding points of convolutions / matmuls (inputs, weights, bias and result in bf16, fp32
accumulation) but NOT the fp32 policy for group_norm (CUDA autocast runs GroupNorm in fp32; on CPU it stays in its input's bf16), so
it is NOISIER than CUDA autocast and than the oracle's bf16 mode, which follows the CUDA policy.  It bounds the bf16 spread of the
reference's own arithmetic from above; tests/test_oracle_golden.py places oracle(quant="bf16") inside it.

    python tests/golden/make_autocast_fixture.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)


def main():
    import numpy as np
    import torch
    from make_golden import install_shims, t2n
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_shims()
    tmp = tempfile.mkdtemp()
    sys.argv = ["x", "--split", f"{REF}/src/data/new_split.npy", "--model", "base", "--output_dir", tmp]
    sys.path.insert(0, f"{REF}/src")
    from models.CamRaDepth import CamRaDepth  # noqa: E402  (reference)
    from utils.loss_funcs import MaskedSmoothL1Loss, MaskedMSELoss  # noqa: E402
    from camradepth_amd import synth

    model = CamRaDepth(input_channels=7)
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    model.load_state_dict(sd, strict=True)
    model.eval()
    crit_d, crit_m = MaskedSmoothL1Loss(), MaskedMSELoss()

    def loss_of(out, batch):
        inter = out["depth"]["intermediate_depths"]
        l4 = crit_d(inter[-1].float().squeeze(1), batch["gt_half"].squeeze(1))
        l3 = crit_d(inter[-2].float().squeeze(1), batch["gt_quarter"].squeeze(1))
        lf = crit_d(out["depth"]["final_depth"].float(), batch["gt_full"])
        w = [1, 1, 1, 0.2, 0.2]
        loss = (w[0] * lf + w[1] * l4 + w[2] * l3) / sum(w)
        rmse = torch.sqrt(crit_m(out["depth"]["final_depth"].float(), batch["gt_full"]))
        return loss, lf, l4, l3, rmse

    store = {}
    for tag, (B, H, W, seed) in {"64": (2, 64, 96, 77), "256": (1, 256, 416, 1234)}.items():
        batch = synth.make_batch(B, H, W, seed=seed)
        for mode in ("fp32", "autocast"):
            model.zero_grad(set_to_none=True)
            if mode == "autocast":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    out = model(batch["image"])
                    loss, lf, l4, l3, rmse = loss_of(out, batch)
            else:
                out = model(batch["image"])
                loss, lf, l4, l3, rmse = loss_of(out, batch)
            loss.backward()
            store[f"{mode}_final_depth_{tag}"] = t2n(out["depth"]["final_depth"].float())
            store[f"{mode}_loss_{tag}"] = np.array([float(loss), float(lf), float(l4), float(l3), float(rmse)], dtype=np.float64)
            store[f"{mode}_gradnorms_{tag}"] = np.array([float(p.grad.float().norm()) if p.grad is not None else -1.0
                                                        for _, p in model.named_parameters()], dtype=np.float64)
        a, f = store[f"autocast_final_depth_{tag}"], store[f"fp32_final_depth_{tag}"]
        print(tag, "reference CPU autocast(bf16) vs its own fp32: final depth rel-L2",
              float(np.linalg.norm(a - f) / np.linalg.norm(f)), "loss", store[f"autocast_loss_{tag}"][0], store[f"fp32_loss_{tag}"][0])
    np.savez_compressed(os.path.join(HERE, "ref_autocast_bf16.npz"), **store)


if __name__ == "__main__":
    main()
