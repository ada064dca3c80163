"""RMSE of the CPU oracle in fp32 and in bf16 mode against the reference's fp32 output on the 256x416 golden fixture (golden weights):
the bf16 floor of the reference's own arithmetic, next to which the HIP path's gap is read (tests/test_gpu_model.py).
Runs on CPU in the build container; writes tests/golden/oracle_bf16_gap.json."""
import json
import sys, time, math, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camradepth_amd import synth
from camradepth_amd.config import ModelConfig
from tests.util import golden_state_dict, load_npz
from oracle import model as om, losses as ol
torch.set_num_threads(8)
cfg = ModelConfig.variant("base")
g = load_npz("forward256x416_base.npz")
sd = golden_state_dict(cfg)
batch = synth.make_batch(1, 256, 416, seed=1234)
res = {}
for quant in (None, "bf16"):
    t0 = time.time()
    with torch.no_grad():
        o = om.forward(sd, batch["image"], cfg, quant=quant) if quant else om.forward(sd, batch["image"], cfg)
    rmse = float(torch.sqrt(ol.masked_mse(o["depth"]["final_depth"], batch["gt_full"])))
    rel = float((o["depth"]["final_depth"] - torch.from_numpy(g["final_depth"])).norm() / torch.from_numpy(g["final_depth"]).norm())
    res[str(quant)] = (rmse, rel)
    print(quant, "rmse", rmse, "ref rmse", float(g["loss"][5]), "gap", abs(rmse - float(g["loss"][5])), "rel-L2 final vs reference", rel, f"{time.time()-t0:.0f}s", flush=True)

out = {"fixture": "forward256x416_base.npz (golden weights, batch seed 1234)", "reference_rmse_fp32": float(g["loss"][5]),
       "oracle_fp32": {"rmse": res["None"][0], "gap": abs(res["None"][0] - float(g["loss"][5])), "final_depth_rel_l2_vs_reference": res["None"][1]},
       "oracle_bf16": {"rmse": res["bf16"][0], "gap": abs(res["bf16"][0] - float(g["loss"][5])), "final_depth_rel_l2_vs_reference": res["bf16"][1]}}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oracle_bf16_gap.json"), "w"), indent=1)
print(out)
