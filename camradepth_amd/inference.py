"""Eval-mode forward replayed from one HIP graph: the reference's own "runtime" metric (the timed forward of
Trainer.test, src/main/runner.py:417-420) without per-kernel launch cost.  SURVEY 8f N4."""
import torch

from . import lib as L


class InferenceGraph:
    """model.eval() forward on static buffers for a fixed (B, H, W), captured once.  run(x) copies x into the plan's input
    buffer, replays the graph and returns the reference's nested output dict (CamRaDepth.py:169-170) -- fresh tensors, like
    the module's forward; run(x, clone=False) returns views of the static output buffers instead (valid until the next
    run(): a caller that keeps two results would see them alias)."""

    def __init__(self, model, B, H, W):
        if model.flat is None or not model.flat.is_cuda:
            raise L.CrdError("InferenceGraph needs the model on an MI355X (no CPU fallback)")
        was_training = model.training
        model.eval()
        self.model, self.B, self.H, self.W = model, B, H, W
        x = torch.zeros((B, model.cfg.input_channels, H, W), device=model.flat.device)
        prev = model.__dict__.get("_need_grad", True)
        model.__dict__["_need_grad"] = False        # inference: nothing is kept for a backward pass
        try:
            self.plan = model._plan_for(x)
        finally:
            model.__dict__["_need_grad"] = prev     # (part of the plan key: a later model._plan_for() must not inherit it)
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.plan.forward()                     # warm-up outside the capture (lazy module loading)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.plan.forward(pack=False)       # weights are packed when they change (replay()), not per frame
        torch.cuda.current_stream().wait_stream(self.stream)
        model.train(was_training)

    def replay(self):
        self.plan.ensure_packed()                   # optimizer step / load_state_dict / mark_params_changed() since the last frame
        self.graph.replay()

    def run(self, x, clone=True):
        p = self.plan
        p.x_in.copy_(x.to(torch.float32))
        self.replay()
        B, H, W = self.B, self.H, self.W
        final = p.out_depth[5].t.view(B, 1, H, W)
        half = p.out_depth[4].t.view(B, 1, H // 2, W // 2)
        quarter = p.out_depth[3].t.view(B, 1, H // 4, W // 4)
        seg = p.seg_out if p.seg_logits is not None else None
        unsup = p.unsup_map
        if clone:
            final, half, quarter = final.clone(), half.clone(), quarter.clone()
            seg = seg.clone() if seg is not None else None
            unsup = unsup.clone() if unsup is not None else None
        return {"depth": {"intermediate_depths": (None, None, quarter, half), "final_depth": final},
                "seg": {"final_seg": seg, "intermediate_seg": None, "unsup_map": unsup}}
