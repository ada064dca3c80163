"""diffGradNorm optimizer on one multi-tensor HIP kernel sequence (3 launches per step, no host sync).

Same constructor, `param_groups` and per-parameter `state` keys as the reference
(src/models/diffGradNorm.py:26-37,63-71) so `OneCycleLR(cycle_momentum=True)` can drive `lr` and
`betas[0]` every iteration (src/main/runner.py:151-152,270) and optimizer state_dicts interchange.
"""
import torch
from torch.optim.optimizer import Optimizer

from . import lib as L

_CHUNK = 4096


class diffGradNorm(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._groups = None

    # ------------------------------------------------------------------ flat layout
    def _build(self, group):
        ps = [p for p in group["params"]]
        dev = ps[0].device
        if not ps[0].is_cuda:
            raise L.CrdError("camradepth_amd.diffGradNorm runs on an MI355X only (no CPU fallback)")
        # parameters that already live in one flat buffer (camradepth_amd.CamRaDepth) are used in place
        base = min(p.data_ptr() for p in ps)
        offs = [(p.data_ptr() - base) // 4 for p in ps]
        order = sorted(range(len(ps)), key=lambda i: offs[i])
        span = max(o + p.numel() for o, p in zip(offs, ps))
        contiguous = span <= 2 * sum(p.numel() for p in ps) + 8 * len(ps) and all(
            offs[order[i]] + ps[order[i]].numel() <= offs[order[i + 1]] for i in range(len(ps) - 1))
        st = {"ps": ps, "adopted": not contiguous}
        if not contiguous:
            # adopt: move the parameters into one flat buffer (their .data become views)
            n, offs = 0, []
            for p in ps:
                offs.append(n)
                n += (p.numel() + 7) // 8 * 8
            flat = torch.zeros(n, dtype=torch.float32, device=dev)
            for p, o in zip(ps, offs):
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[o:o + p.numel()].view(p.shape)
            st["flat_p"], span = flat, n
        else:
            st["flat_p"] = None
        st["offs"], st["span"], st["base"] = offs, span, min(p.data_ptr() for p in ps)
        st["flat_g"] = torch.zeros(span, dtype=torch.float32, device=dev)
        st["m"], st["v"], st["pg"] = (torch.zeros(span, dtype=torch.float32, device=dev) for _ in range(3))
        nt = len(ps)
        st["egn"], st["fac"] = (torch.zeros(nt, dtype=torch.float32, device=dev) for _ in range(2))
        seg = torch.tensor([[o, o + p.numel()] for o, p in zip(offs, ps)], dtype=torch.int64)
        # kernel wants seg_off[t], seg_off[t+1]: store begin/end pairs as 2*t, 2*t+1 and index tensors by 2*t
        b2s, b2c = [], []
        for t, p in enumerate(ps):
            for c in range((p.numel() + _CHUNK - 1) // _CHUNK):
                b2s.append(t)
                b2c.append(c)
        st["seg"], st["nblk"] = seg.to(dev), len(b2s)
        st["nsq"] = torch.zeros(len(b2s), dtype=torch.float32, device=dev)      # per-workgroup parts of ||g||^2
        st["b2s"] = torch.tensor(b2s, dtype=torch.int32, device=dev)
        st["b2c"] = torch.tensor(b2c, dtype=torch.int32, device=dev)
        st["active"] = torch.ones(nt, dtype=torch.uint8, device=dev)
        st["step"] = 0
        for t, (p, o) in enumerate(zip(ps, offs)):
            s = self.state[p]
            s["step"] = 0
            s["exp_avg"] = st["m"][o:o + p.numel()].view(p.shape)
            s["exp_avg_sq"] = st["v"][o:o + p.numel()].view(p.shape)
            s["previous_grad"] = st["pg"][o:o + p.numel()].view(p.shape)
            s["exp_grad_norm"] = st["egn"][t]
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._groups is None:
            self._groups = [self._build(g) for g in self.param_groups]
        lb = L.load()
        for group, st in zip(self.param_groups, self._groups):
            ps, offs = st["ps"], st["offs"]
            # gradient source: in place when the grads are views of one buffer parallel to the params
            act_host = [p.grad is not None for p in ps]
            if not any(act_host):
                continue
            i0 = act_host.index(True)
            g0 = ps[i0].grad
            # The kernel's bias corrections use ONE step count per group (st["step"]).  The reference keeps one per parameter
            # (diffGradNorm.py:66,76-77): they differ only for a parameter that is frozen for some steps and unfrozen later
            # (or a checkpoint with non-uniform steps) -- including the case where the WHOLE active set switches (A frozen after
            # n steps, B unfrozen: B's own count is 0, the group's is n).  Refuse that silently-different case instead of
            # approximating it, BEFORE anything is modified (no kernel launch, no change of st["active"] / st["act_host"]).
            steps = {self.state[p]["step"] + 1 for p, a_ in zip(ps, act_host) if a_}
            if steps != {st["step"] + 1}:
                raise L.CrdError("camradepth_amd.diffGradNorm: the active parameters' step counts "
                                 f"({sorted(steps)}) differ from the group's ({st['step'] + 1}): unfreezing a parameter mid-run (or "
                                 "loading such a checkpoint) needs one param_group per step count")
            # frozen parameters (grad None) are skipped through the `active` mask; the others' grads are used in place
            parallel = all(p.grad is None or (p.grad.data_ptr() - g0.data_ptr()) == 4 * (o - offs[i0]) for p, o in zip(ps, offs))
            if parallel:
                gptr = g0.data_ptr() - 4 * offs[i0]
                if act_host != st.get("act_host"):
                    st["active"].copy_(torch.tensor(act_host, dtype=torch.uint8))
                    st["act_host"] = act_host
            else:
                fg = st["flat_g"]
                for p, o in zip(ps, offs):
                    if p.grad is not None:
                        fg[o:o + p.numel()].copy_(p.grad.reshape(-1))
                gptr = fg.data_ptr()
                st["active"].copy_(torch.tensor(act_host, dtype=torch.uint8))
                st["act_host"] = None
            st["step"] += 1
            beta1, beta2 = group["betas"]
            pbase = st["flat_p"].data_ptr() if st["flat_p"] is not None else st["base"]
            L.check(lb.crd_diffgradnorm_step(pbase, gptr, st["m"].data_ptr(), st["v"].data_ptr(), st["pg"].data_ptr(),
                                             st["egn"].data_ptr(), st["nsq"].data_ptr(), st["fac"].data_ptr(),
                                             st["seg"].data_ptr(), st["b2s"].data_ptr(), st["b2c"].data_ptr(), len(ps),
                                             st["nblk"], None if all(act_host) else st["active"].data_ptr(), float(group["lr"]),
                                             float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]),
                                             st["step"], None, L.stream()), "crd_diffgradnorm_step")
            for p, a_ in zip(ps, act_host):
                if a_:
                    self.state[p]["step"] += 1
            for ow in {getattr(p, "_crd_owner", None) for p in ps}:        # graph-replayed forwards re-pack their weights
                if ow is not None and ow() is not None:
                    ow().mark_params_changed()
        return loss

    def load_state_dict(self, state_dict):
        """Restores a checkpoint written by this class or by the reference's diffGradNorm (same per-parameter keys:
        step, exp_avg, exp_avg_sq, previous_grad, exp_grad_norm; runner.py:369).  torch's loader replaces the state
        tensors; the kernels work on flat buffers the state entries are views of, so the loaded values are copied into
        those buffers and the views re-attached."""
        super().load_state_dict(state_dict)
        loaded = {p: dict(self.state[p]) for g in self.param_groups for p in g["params"] if p in self.state}
        if self._groups is None:
            self._groups = [self._build(g) for g in self.param_groups]
        for group, st in zip(self.param_groups, self._groups):
            for t, (p, o) in enumerate(zip(st["ps"], st["offs"])):
                n = p.numel()
                views = {"exp_avg": st["m"][o:o + n].view(p.shape), "exp_avg_sq": st["v"][o:o + n].view(p.shape),
                         "previous_grad": st["pg"][o:o + n].view(p.shape)}
                src = loaded.get(p)
                if src is not None:
                    for k, v in views.items():
                        if k in src:
                            v.copy_(torch.as_tensor(src[k]).to(v.device, v.dtype).reshape(p.shape))
                    if "exp_grad_norm" in src:
                        st["egn"][t] = float(src["exp_grad_norm"])
                    st["step"] = max(st["step"], int(src.get("step", 0)))
                s = self.state[p]
                s.update(views)
                s["exp_grad_norm"] = st["egn"][t]
                s["step"] = st["step"]

    def zero_grad(self, set_to_none=False):
        """Gradients live in one flat buffer that the backward kernels accumulate into; they are zeroed in
        place (the reference's set_to_none=True would detach the views the kernels write through)."""
        for group in self.param_groups:
            owners = {getattr(p, "_crd_owner", None) for p in group["params"]}
            if len(owners) == 1 and None not in owners and next(iter(owners))() is not None:
                next(iter(owners))().zero_grad()          # one fill of the flat gradient buffer
                continue
            for p in group["params"]:
                if p.grad is not None:
                    p.grad.zero_()
