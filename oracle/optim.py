"""CPU oracle for the diffGradNorm optimizer step. TEST INFRASTRUCTURE ONLY (see oracle/model.py).

Parity status: PINNED by tests/golden/diffgradnorm_*.npz (40 steps of the imported reference).
"""
import math
import torch


def new_state(p):
    """Lazy state init (reference: src/models/diffGradNorm.py:63-71)."""
    return {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p),
            "previous_grad": torch.zeros_like(p), "exp_grad_norm": torch.zeros((), dtype=p.dtype)}


def step_tensor(p, g, st, lr, beta1, beta2, eps=1e-8, weight_decay=0.0):
    """One diffGradNorm update of one tensor, in place (reference: src/models/diffGradNorm.py:73-110)."""
    st["step"] += 1
    if weight_decay != 0:
        g = g + weight_decay * p
    n = torch.linalg.norm(g)
    e = 0.95 * st["exp_grad_norm"] + 0.05 * n
    g1 = g * e / (n + 1e-8) if bool(e > n) else g
    st["exp_grad_norm"] = e.clone()
    st["exp_avg"].mul_(beta1).add_(g1, alpha=1 - beta1)
    st["exp_avg_sq"].mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = st["exp_avg_sq"].sqrt().add_(eps)
    bc1 = 1 - beta1 ** st["step"]
    bc2 = 1 - beta2 ** st["step"]
    dfc = 1.0 / (1.0 + torch.exp(-torch.abs(st["previous_grad"] - g)))
    st["previous_grad"] = g.clone()
    step_size = lr * math.sqrt(bc2) / (bc1 + 1e-8)
    p.addcdiv_(st["exp_avg"] * dfc, denom, value=-step_size)


def one_cycle_schedule(total_steps, max_lr, div_factor=2.0, pct_start=0.15, final_div_factor=1e4,
                       base_momentum=0.85, max_momentum=0.95):
    """(lr, beta1) per step of torch OneCycleLR(cos, cycle_momentum=True) as set up in runner.py:151-152."""
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / final_div_factor
    up_end = float(pct_start * total_steps) - 1
    out = []

    def cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)
    for s in range(total_steps):
        if s <= up_end:
            pct = s / up_end if up_end > 0 else 1.0
            out.append((cos(initial_lr, max_lr, pct), cos(max_momentum, base_momentum, pct)))
        else:
            pct = (s - up_end) / (total_steps - 1 - up_end)
            out.append((cos(max_lr, min_lr, pct), cos(base_momentum, max_momentum, pct)))
    return out
