"""Child process of the multi-GPU TrainStep tests (tests/test_gpu_train.py); run as

    python tests/dist_child.py force1 <port>           one rank, RCCL group of one (CRD_FORCE_DIST)
    python tests/dist_child.py rank <port> <world> <rank> <outdir>

`force1`: runs the non-distributed graph step, then the distributed control flow (loss all-reduce, per-bucket
asynchronous all-reduce behind the late graphs, optimizer graph) in a process group of one, and checks that both apply
the same update.  `rank`: one rank of a `world`-GPU job; 20 steps on rank-dependent data, then dumps a checksum of the
parameters so the parent can bit-compare the replicas (replicas must not drift through diffGradNorm's
`e > n` branch).
"""
import dataclasses
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def make(cfg, sd):
    from camradepth_amd.model import CamRaDepth
    m = CamRaDepth(input_channels=cfg.input_channels, depths=cfg.depths, supervised_seg=cfg.supervised_seg)
    m.load_state_dict(sd)
    return m.cuda().train()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def one_step(cfg, sd, batch, masks, steps=1):
    from camradepth_amd.trainer import TrainStep
    m = make(cfg, sd)
    ts = TrainStep(m, batch["image"].shape[0], batch["image"].shape[2], batch["image"].shape[3], lr=1e-3, use_graph=True)
    ts.set_batch(batch)
    ts.plan.training_masks_fixed = True
    ts.plan.dp_masks.copy_(torch.stack([t.cuda() for t in masks["drop_path"]]))
    ts.plan.d2_masks.copy_(torch.stack([t.cuda() for t in masks["dropout2d"]]))
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize()
    return m, ts


def main():
    import torch.distributed as dist
    from camradepth_amd import synth
    from camradepth_amd.config import ModelConfig
    from camradepth_amd.params import param_specs
    mode, port = sys.argv[1], sys.argv[2]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    cfg = dataclasses.replace(ModelConfig.variant("supervised_seg"), depths=(1, 1, 1, 1))
    sd = synth.fill_state_dict({n: s for n, s in param_specs(cfg)}, 0)
    if mode == "force1":
        torch.cuda.set_device(0)
        batch = {k: v.cuda() for k, v in synth.make_batch(2, 64, 96, seed=9).items()}
        masks = synth.make_masks(cfg, 2, seed=1)
        m0, ts0 = one_step(cfg, sd, batch, masks)
        assert not ts0.dist_active
        os.environ["CRD_FORCE_DIST"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        m1, ts1 = one_step(cfg, sd, batch, masks)
        assert ts1.dist_active and ts1.late_wgrad
        l0, l1 = ts0.losses(), ts1.losses()
        out = {"loss": [l0["loss"], l1["loss"]], "grad_rel": rel(m1.flat_grad, m0.flat_grad), "param_rel": rel(m1.flat, m0.flat)}
        dist.destroy_process_group()
        print("RESULT " + json.dumps(out), flush=True)
        return
    world, rank, outdir = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from camradepth_amd.trainer import TrainStep
    m = make(cfg, sd)
    ts = TrainStep(m, 2, 64, 96, lr=1e-3, use_graph=True)
    ts.set_batch({k: v.cuda() for k, v in synth.make_batch(2, 64, 96, seed=100 + rank).items()})
    for _ in range(20):
        ts.step()
    torch.cuda.synchronize()
    masks = ts.plan.d2_masks.cpu()
    torch.save({"flat": m.flat.cpu(), "d2_masks": masks, "rng_rank": m.rng_rank}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
