import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def subsample(t, maxn=8192):
    """Same strided view as tests/golden/make_golden.py:sub."""
    a = t.detach().cpu().float().numpy().reshape(-1)
    stride = max(1, -(-a.size // maxn))
    return a[::stride]


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def param_order(variant):
    with open(os.path.join(GOLDEN, f"param_order_{variant}.json")) as f:
        return json.load(f)


def golden_state_dict(cfg, seed=0):
    """The deterministic weights the fixtures were generated with."""
    from camradepth_amd.params import param_specs
    from camradepth_amd.synth import fill_state_dict
    return fill_state_dict({n: s for n, s in param_specs(cfg)}, seed)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def max_err(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


# ---- crd_sum_t accumulators (include/camradepth_hip.h): 64-bit fixed point, value = integer * 2^-FRAC_BITS ----
STAT_BITS, GRAD_BITS = 20, 44


def zsum(*shape):
    """A zeroed crd_sum_t buffer on the GPU."""
    return torch.zeros(*shape, dtype=torch.int64, device="cuda")


def sval(t):
    """Value of forward-statistic / loss sums as fp32 on the host."""
    return (t.double() * 2.0 ** -STAT_BITS).float().cpu()


def gval(t):
    """Value of gradient sums as fp32 on the host."""
    return (t.double() * 2.0 ** -GRAD_BITS).float().cpu()


def to_stat(x):
    return (x.double() * 2.0 ** STAT_BITS).round().to(torch.int64)


def to_grad(x):
    return (x.double() * 2.0 ** GRAD_BITS).round().to(torch.int64)
