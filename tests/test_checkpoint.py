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


# ---- a checkpoint WRITTEN BY THE REFERENCE (tests/golden/make_checkpoint_fixture.py: DataParallel-wrapped reference
# model + the reference's diffGradNorm, saved with the dictionary of runner.py:369-371) --------------------------------
TINY = dict(depths=(1, 1, 1, 1), dims=(16, 16, 16, 16), heads=(1, 1, 1, 1))


def _ref_checkpoint():
    import gzip
    import os
    from tests.util import GOLDEN
    with gzip.open(os.path.join(GOLDEN, "ref_checkpoint_tiny.pth.gz"), "rb") as f:
        return torch.load(io.BytesIO(f.read()), map_location="cpu", weights_only=False)


def test_reference_written_checkpoint_loads_into_the_module():
    from camradepth_amd.model import CamRaDepth
    from tests.util import load_npz
    state = _ref_checkpoint()
    assert set(state) == {"state_dict", "optimizer", "lr", "steps"} and state["steps"] == [2, 0]
    assert all(k.startswith("module.") for k in state["state_dict"])            # nn.DataParallel (runner.py:135-136)
    m = CamRaDepth(input_channels=7, seed=5, **TINY)
    missing, mismatched, steps = ck.load_checkpoint(state, m)
    assert not missing and not mismatched and steps == [2, 0]
    own = m.state_dict()
    assert ["module." + k for k in own] == list(state["state_dict"])            # same keys in the same order
    for k, v in own.items():
        assert torch.equal(v, state["state_dict"]["module." + k]), k
    assert m._flat_ok()                                                          # parameters are still views of one flat buffer
    # the optimizer part is positional (torch.optim state_dict): indices follow the parameter registration order, and
    # only parameters that ever had a gradient own state (diffGradNorm.py:54-55,63-71)
    osd = state["optimizer"]
    names = [n for n, _ in m.named_parameters()]
    assert osd["param_groups"][0]["params"] == list(range(len(names)))
    g = load_npz("ref_checkpoint_tiny.npz")
    frozen = set(g["frozen"].tolist())
    assert sorted(osd["state"]) == [i for i, n in enumerate(names) if n not in frozen]
    assert [names[i] for i in sorted(osd["state"])] == g["names"].tolist()
    st = osd["state"][0]
    assert set(st) == {"step", "exp_avg", "exp_avg_sq", "previous_grad", "exp_grad_norm"} and st["step"] == 2


@pytest.mark.gpu
def test_resume_from_reference_written_checkpoint_takes_the_reference_next_step():
    """Model + optimizer state restored from the reference's file; fed the gradients the reference saw in its next
    iteration, diffGradNorm.step() must land on the reference's next parameters (frozen tensors untouched)."""
    import numpy as np
    from camradepth_amd.model import CamRaDepth
    from camradepth_amd.optim import diffGradNorm
    from tests.util import load_npz
    state, g = _ref_checkpoint(), load_npz("ref_checkpoint_tiny.npz")
    m = CamRaDepth(input_channels=7, seed=5, **TINY).cuda()
    frozen = set(g["frozen"].tolist())
    for n, p in m.named_parameters():
        if n in frozen:
            p.requires_grad_(False)
    opt = diffGradNorm(m.parameters(), lr=state["lr"])
    ck.load_checkpoint(state, m, opt)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    for n, p in m.named_parameters():
        p.grad = None if n in frozen else torch.from_numpy(g["grad:" + n]).cuda()
    opt.step()
    torch.cuda.synchronize()
    for i, (n, p) in enumerate(m.named_parameters()):
        if n in frozen:
            assert torch.equal(p, before[n]), n
        else:
            np.testing.assert_allclose(p.detach().cpu().numpy(), g["after:" + n], rtol=2e-5, atol=1e-7, err_msg=n)
            assert int(opt.state[p]["step"]) == 3
    egn = [float(opt.state[p]["exp_grad_norm"]) for n, p in m.named_parameters() if n not in frozen]
    np.testing.assert_allclose(egn, g["exp_grad_norm"], rtol=1e-5)
