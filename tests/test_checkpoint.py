"""Checkpoint interchange: reference dictionary format, 'module.' prefix, shape-matched transfer between
variants; on the GPU: optimizer state survives a save / load and training resumes with the identical update."""
import io

import pytest
import torch

from camradepth_amd import checkpoint as ck


def _models():
    from camradepth_amd.model import CamRaDepth
    return (CamRaDepth(input_channels=7, supervised_seg=True, seed=1), CamRaDepth(input_channels=7, seed=2))


def test_state_dict_round_trip_and_shape_matched_transfer(tmp_path):
    sup, base = _models()
    state = ck.save_checkpoint(str(tmp_path / "sup.pth"), sup)
    assert set(state) >= {"state_dict", "steps"}
    assert all(v.device.type == "cpu" and v.dtype == torch.float32 for v in state["state_dict"].values())
    # DataParallel-style prefix + a variant with fewer parameters: every common key transfers, nothing else changes
    prefixed = {"state_dict": {"module." + k: v for k, v in state["state_dict"].items()}}
    before = {k: v.clone() for k, v in base.state_dict().items()}
    missing, mismatched, _ = ck.load_checkpoint(prefixed, base)
    after = base.state_dict()
    assert not missing                                  # the base model is a subset of the supervised-seg model
    for k, v in after.items():
        if k in mismatched:
            assert torch.equal(v, before[k])
        else:
            assert torch.equal(v.cpu(), state["state_dict"][k]), k
    # and the other way round: keys the checkpoint lacks are reported and keep their values
    sup2, _ = _models()
    ref = {k: v.clone() for k, v in sup2.state_dict().items()}
    missing, mismatched, _ = ck.load_checkpoint({"state_dict": {k: v.cpu() for k, v in base.state_dict().items()}}, sup2)
    assert missing and all(k.startswith(("seg_", "unsup_")) for k in missing)
    for k in missing:
        assert torch.equal(sup2.state_dict()[k], ref[k])
    # the file on disk is a plain torch.save dictionary with reference key names
    loaded = torch.load(str(tmp_path / "sup.pth"), map_location="cpu", weights_only=False)
    assert list(loaded["state_dict"]) == list(sup.state_dict())


@pytest.mark.gpu
def test_optimizer_state_survives_checkpoint_and_resumes_identically():
    from camradepth_amd.optim import diffGradNorm
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 7, 3, 3), (64,), (33, 5)]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(5)]

    def fresh():
        gg = torch.Generator().manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(s, generator=gg).cuda()) for s in shapes]
        return ps, diffGradNorm(ps, lr=1e-3)

    def run(ps, opt, steps):
        for gs in steps:
            for p, gr in zip(ps, gs):
                p.grad = gr.cuda()
            opt.step()

    ps_a, opt_a = fresh()
    run(ps_a, opt_a, grads[:3])
    buf = io.BytesIO()
    torch.save({"optimizer": opt_a.state_dict(), "p": [p.detach().cpu() for p in ps_a]}, buf)
    run(ps_a, opt_a, grads[3:])
    # resume from the checkpoint in a new process-like state
    buf.seek(0)
    st = torch.load(buf, map_location="cpu", weights_only=False)
    ps_b, opt_b = fresh()
    with torch.no_grad():
        for p, v in zip(ps_b, st["p"]):
            p.copy_(v.cuda())
    opt_b.load_state_dict(st["optimizer"])
    s0 = opt_b.state[ps_b[0]]
    assert int(s0["step"]) == 3 and set(s0) >= {"exp_avg", "exp_avg_sq", "previous_grad", "exp_grad_norm"}
    run(ps_b, opt_b, grads[3:])
    for pa, pb in zip(ps_a, ps_b):
        assert torch.equal(pa, pb), "resumed run diverges from the uninterrupted one"
