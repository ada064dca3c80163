"""Epoch-level entry points (camradepth_amd.runner.Trainer: train_one_epoch / eval / test / train, the names of
src/main/runner.py:166-519) over synthetic loaders: accumulation windows and the last-batch flush, both batch layouts,
and the test() metrics against the oracle's restatement of runner.py:443-465 on the same predictions."""
import dataclasses

import numpy as np
import pytest
import torch

from camradepth_amd import synth
from camradepth_amd.config import ModelConfig
from camradepth_amd.params import param_specs

pytestmark = pytest.mark.gpu


def nested(b):
    """The reference dataloader's layout (src/data/dataloader.py:320-333)."""
    return {"image": b["image"], "name": ["sunny"] * b["image"].shape[0],
            "gt": {"depth": {"lidar_depth": b["gt_full"], "lidar_depth_partial": (b["gt_half"], b["gt_quarter"], b["gt_quarter"])},
                   "seg": {"final_seg": b["seg"], "intermediate_seg": b["seg"]}}}


def test_trainer_epoch_entry_points():
    from camradepth_amd.model import CamRaDepth
    from camradepth_amd.runner import Trainer
    from oracle import losses as ol
    cfg = dataclasses.replace(ModelConfig.variant("supervised_seg"), depths=(1, 1, 1, 1))
    sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
    m = CamRaDepth(input_channels=7, depths=cfg.depths, supervised_seg=True)
    m.load_state_dict(sd)
    m = m.cuda().train()
    train = [synth.make_batch(2, 64, 96, seed=10 + i) for i in range(5)]
    train[1], train[3] = nested(train[1]), nested(train[3])            # both layouts in one loader
    val = [synth.make_batch(2, 64, 96, seed=30 + i) for i in range(2)]
    test = [nested(synth.make_batch(1, 64, 96, seed=40 + i)) for i in range(3)]
    tr = Trainer(m, train, val, test, learning_rate=1e-3, num_epochs=2, update_interval=2)
    p0 = m.flat.clone()
    r = tr.train_one_epoch(0)
    torch.cuda.synchronize()
    # 5 batches, update_interval 2: optimizer after batches 2, 4 and (flush) 5  (runner.py:222)
    assert tr.training_steps == 3 and tr.step.step_count == 3 and tr.step.iter_count == 5
    assert not torch.equal(m.flat, p0)
    assert all(np.isfinite(v) for v in r.values()) and r["RMSE"] > 0 and r["seg_mean"] > 0
    val_loss, rmse = tr.eval(0)
    assert np.isfinite(val_loss) and rmse > 0 and m.training
    res = tr.test()
    k100, k50 = "max_depth_100", "max_depth_50"
    assert set(res) >= {"time", k100, k50, "IoU"} and res["time"] > 0
    # against the oracle's metric restatement on the module's own predictions
    exp = []
    m.eval()
    with torch.no_grad():
        for b in test:
            out = m(b["image"].cuda())
            mt = ol.test_metrics(out["depth"]["final_depth"].cpu(), b["gt"]["depth"]["lidar_depth"])
            if mt is not None:
                exp.append(mt)
    m.train()
    for key in ("RMSE", "MAE", "REL"):
        np.testing.assert_allclose(res[k100][key], np.mean([e[key] for e in exp]), rtol=1e-4)
    assert 0.0 <= res["IoU"] <= 1.0 or np.isnan(res["IoU"])
    # second epoch through train(): the schedule continues, best validation loss is returned
    best = tr.train()
    assert np.isfinite(best) and tr.training_steps == 3 + 2 * 3
