"""The CPU oracle in bf16 mode against the reference's fp32 output on the 928x1600 supervised_seg fixture (golden weights): the
distance bf16 storage alone produces at this size with these ill-conditioned weights, next to which the HIP path's distance is
read (tests/test_gpu_model.py::test_full_resolution_928x1600_matches_reference_golden).  Runs on CPU in the build container;
writes tests/golden/oracle_bf16_gap_fullres.json."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camradepth_amd import synth
from camradepth_amd.config import ModelConfig
from tests.util import golden_state_dict, load_npz
from oracle import model as om, losses as ol
torch.set_num_threads(8)
cfg = ModelConfig.variant("supervised_seg")
g = load_npz("forward928x1600_supervised_seg.npz")
sd = golden_state_dict(cfg)
batch = synth.make_batch(1, 928, 1600, seed=1234)
out = {"fixture": "forward928x1600_supervised_seg.npz (golden weights, batch seed 1234)"}
rel = lambda a, b: float((a - b).norm() / b.norm())
for quant in (None, "bf16"):
    t0 = time.time()
    with torch.no_grad():
        o = om.forward(sd, batch["image"], cfg, quant=quant) if quant else om.forward(sd, batch["image"], cfg)
    fd = o["depth"]["final_depth"]
    am = o["seg"]["final_seg"][0].argmax(0)[::4, ::4].numpy().astype(np.uint8)
    miss = am != g["seg_argmax_s4"]
    clear = g["seg_margin_s4"].astype(np.float32) > 0.5 * float(g["seg_logit_rms"][0])
    r = {"rmse": float(torch.sqrt(ol.masked_mse(fd, batch["gt_full"]))), "final_rel_l2": rel(fd[0, 0, ::4, ::4], torch.from_numpy(g["final_depth_s4"])),
         "half_rel_l2": rel(o["depth"]["intermediate_depths"][3][0, 0, ::4, ::4], torch.from_numpy(g["depth_half_s4"])),
         "quarter_rel_l2": rel(o["depth"]["intermediate_depths"][2][0, 0, ::2, ::2], torch.from_numpy(g["depth_quarter_s2"])),
         "seg_argmax_mismatch": float(miss.mean()), "seg_argmax_mismatch_clear_margin": float(miss[clear].mean())}
    out["oracle_" + ("bf16" if quant else "fp32")] = r
    print(quant, r, f"{time.time() - t0:.0f}s", flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oracle_bf16_gap_fullres.json"), "w"), indent=1)
