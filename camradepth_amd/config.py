"""Model constants and variants of the CamRaDepth hot path.

The reference reads these from a global argparse namespace evaluated at import time
(reference: src/utils/args.py:11-66,156-166). Here they are an explicit, immutable config.
"""
from dataclasses import dataclass
from typing import Tuple

GROUPNORM_DIVISOR = 16   # reference: src/utils/args.py:38
NUM_CLASSES = 21         # reference: src/utils/args.py:27
UNSUP_CLASSES = 19       # reference: src/models/CamRaDepth.py:92-94
MAX_DEPTH = 100.0        # reference: src/utils/args.py:14
MID_CHANNELS = 128       # reference: src/models/CamRaDepth.py:37
DROPOUT2D_P = 0.2        # reference: src/models/CamRaDepth.py:96
DROP_PATH_RATE = 0.1     # reference: src/models/CamRaDepth.py:57
GN_EPS = 1e-5            # torch.nn.GroupNorm default

# reference: src/utils/args.py:156-159
VARIANTS = {
    "base": (False, False),
    "supervised_seg": (True, False),
    "unsupervised_seg": (False, True),
    "sup_unsup_seg": (True, True),
}


@dataclass(frozen=True)
class ModelConfig:
    """Constructor arguments of CamRaDepth (reference: src/models/CamRaDepth.py:21-31)."""
    input_channels: int = 7
    heads: Tuple[int, ...] = (1, 2, 4, 8)
    ff_expansion: Tuple[int, ...] = (8, 8, 4, 4)
    reduction_ratio: Tuple[int, ...] = (8, 4, 2, 1)
    depths: Tuple[int, ...] = (3, 10, 16, 5)
    dims: Tuple[int, ...] = (64, 128, 160, 256)
    supervised_seg: bool = False
    unsupervised_seg: bool = False
    num_classes: int = NUM_CLASSES

    @staticmethod
    def variant(name: str, **kw) -> "ModelConfig":
        sup, unsup = VARIANTS[name]
        return ModelConfig(supervised_seg=sup, unsupervised_seg=unsup, **kw)

    @property
    def drop_path_rates(self):
        """Stochastic-depth rate per block: linspace(0, 0.1, sum(depths)) (simplified_attention.py:214)."""
        n = sum(self.depths)
        if n == 1:
            return [0.0]
        return [DROP_PATH_RATE * i / (n - 1) for i in range(n)]
