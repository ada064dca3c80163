#!/usr/bin/env python3
"""More golden outputs of the REAL reference (build container only: needs /root/reference), one subprocess per model
variant because the reference reads a global `args` at import:

  * forward416x800_base.npz            -- the reference's native evaluation frame (Trainer.test's `runtime` forward,
                                          src/main/runner.py:402-420): SURVEY 8f N4's parity fixture;
  * forward256x416_{unsupervised_seg,sup_unsup_seg}.npz -- the two seg variants at BASELINE's frame size;
  * forward_rgb_{base,sup_unsup_seg}.npz -- the RGB-only variants (input_channels = 3, src/utils/args.py:164-166) at 64x96
                                          (outputs, train-mode loss and gradient norms) and 256x416 (outputs).

    python tests/golden/make_extra_golden.py
"""
import importlib.util
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
JOBS = ["base@416x800", "unsupervised_seg@256x416", "sup_unsup_seg@256x416", "base (rgb)@rgb", "sup_unsup_seg (rgb)@rgb"]


def run(job):
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    variant, what = job.split("@")
    tmp = tempfile.mkdtemp()
    sys.argv = ["x", "--split", f"{REF}/src/data/new_split.npy", "--model", variant, "--output_dir", tmp]
    sys.path.insert(0, f"{REF}/src")
    sys.path.insert(0, REPO)
    from models.CamRaDepth import CamRaDepth            # reference
    from utils.loss_funcs import MaskedMSELoss, MaskedSmoothL1Loss, MaskedFocalLoss   # reference
    from utils.args import args                          # reference (input_channels as the reference derives it)
    from camradepth_amd import synth
    cin = args.input_channels
    model = CamRaDepth(input_channels=cin)
    model.load_state_dict(synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0), strict=True)
    model.eval()
    t2n = mg.t2n

    def eval_forward(H, W):
        b = synth.make_batch(1, H, W, seed=1234)
        with torch.no_grad():
            out = model(b["image"][:, :cin])             # runner.py:193,418: inputs[:, :args.input_channels]
        fd, inter = out["depth"]["final_depth"], out["depth"]["intermediate_depths"]
        st = {"final_depth": t2n(fd).astype(np.float32), "depth_half": t2n(inter[3]).astype(np.float16),
              "depth_quarter": t2n(inter[2]).astype(np.float32),
              "rmse": np.array([float(torch.sqrt(MaskedMSELoss()(fd, b["gt_full"])))])}
        if out["seg"]["final_seg"] is not None:
            st["seg_argmax"] = t2n(out["seg"]["final_seg"].argmax(1)).astype(np.uint8)
        if out["seg"]["unsup_map"] is not None:
            st["unsup_map"] = t2n(out["seg"]["unsup_map"]).astype(np.float16)
        return st

    if what == "416x800":
        st = eval_forward(416, 800)
        st["final_depth"] = st["final_depth"].astype(np.float16)     # 333k values: half precision keeps the fixture small
        np.savez_compressed(os.path.join(HERE, "forward416x800_base.npz"), **st)
    elif what == "256x416":
        st = eval_forward(256, 416)
        np.savez_compressed(os.path.join(HERE, f"forward256x416_{variant}.npz"), **st)
    else:
        tag = variant.split(" ")[0]
        st = {"input_channels": np.array([cin]), "num_params": np.array([sum(p.numel() for p in model.parameters())])}
        for k, v in eval_forward(256, 416).items():
            st["e256_" + k] = v
        # train-mode step at 2 x 3 x 64 x 96 with injected masks: loss terms and per-parameter gradient norms
        from camradepth_amd.config import ModelConfig
        import dataclasses
        cfg = dataclasses.replace(ModelConfig.variant(tag), input_channels=cin)
        masks = synth.make_masks(cfg, 2, seed=4321)
        DropPath = sys.modules["timm.models.layers"].DropPath
        blocks = [blk for s_ in range(1, 5) for blk in getattr(model.dest_encoder, f"block{s_}")]
        for i, blk in enumerate(blocks):
            if isinstance(blk.drop_path, DropPath):
                blk.drop_path.injected = masks["drop_path"][i]
        it = iter(masks["dropout2d"])

        class Inject(torch.nn.Module):
            def forward(self, x):
                return x * next(it).view(x.shape[0], x.shape[1], 1, 1)
        model.dropout = Inject()
        model.train()
        b2 = synth.make_batch(2, 64, 96, seed=77)
        out = model(b2["image"][:, :cin])
        crit_d, crit_s = MaskedSmoothL1Loss(), MaskedFocalLoss()
        inter = out["depth"]["intermediate_depths"]
        seg = out["seg"]["final_seg"]
        l_seg = (crit_s(seg, b2["seg"]) if seg is not None else 0) * (1 if args.supervised_seg else 0)
        l4 = crit_d(inter[-1].squeeze(1), b2["gt_half"].squeeze(1))
        l3 = crit_d(inter[-2].squeeze(1), b2["gt_quarter"].squeeze(1))
        lf = crit_d(out["depth"]["final_depth"], b2["gt_full"])
        loss = (lf + l4 + l3 + 0.2 * l_seg) / 3.4
        loss.backward()
        st["train_final_depth"] = t2n(out["depth"]["final_depth"])
        st["train_loss"] = np.array([float(loss), float(lf), float(l4), float(l3), float(l_seg)])
        st["train_gradnorms"] = np.array([float(p.grad.norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
        np.savez_compressed(os.path.join(HERE, f"forward_rgb_{tag}.npz"), **st)
    print(job, {k: (v.shape if getattr(v, "ndim", 0) else v) for k, v in st.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--job":
        run(sys.argv[2])
    else:
        for j in JOBS:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--job", j], check=True)
