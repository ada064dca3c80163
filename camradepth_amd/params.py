"""Parameter inventory of CamRaDepth: names, shapes and registration order.

The optimizer state of the reference is positional (torch.optim keeps params by index), so a
drop-in must register its tensors under the reference's names and in the reference's order.
Order follows the constructors: src/models/CamRaDepth.py:54-94, src/models/simplified_attention.py
(Block :116-126, Attention_MaxPool :59-70, Mlp :17-24, OverlapPatchEmbed :158-162) and
src/utils/utils.py (ConvLayer :210-215, ShortResBlock :114-124, Depth_Activation :282-283).
"""
from typing import List, Tuple
from .config import ModelConfig, GROUPNORM_DIVISOR, MID_CHANNELS, UNSUP_CLASSES

Spec = Tuple[str, Tuple[int, ...]]


def short_res_block_plan(in_channels: int, out_channels: int, mid: int = MID_CHANNELS):
    """(Cin, Cout) of the three ConvLayers of a ShortResBlock (src/utils/utils.py:107-124)."""
    plan, inp, factor, out = [], in_channels, 0.75, int(mid * 0.75)
    for i in range(3):
        plan.append((inp, out))
        inp += out
        factor -= 0.25
        out = out_channels if i == 1 else int(mid * factor)
    return plan


def decoder_in_channels(cfg: ModelConfig):
    """Input channel count of each depth_upsample stage's ShortResBlock (CamRaDepth.py:67-73)."""
    d = cfg.dims
    return [d[3] + d[2], MID_CHANNELS + d[1], MID_CHANNELS + d[0], MID_CHANNELS + 1,
            MID_CHANNELS + 1 + cfg.input_channels]


def head_in_channels(cfg: ModelConfig):
    extra = int(cfg.supervised_seg) + int(cfg.unsupervised_seg)
    return [MID_CHANNELS, MID_CHANNELS + extra, MID_CHANNELS + extra]  # CamRaDepth.py:75-77


def param_specs(cfg: ModelConfig) -> List[Spec]:
    specs: List[Spec] = []

    def conv(name, cout, cin, kh, kw=None, bias=True):
        if kw is None:
            specs.append((name + ".weight", (cout, cin, kh)))        # Conv1d
        else:
            specs.append((name + ".weight", (cout, cin, kh, kw)))    # Conv2d
        if bias:
            specs.append((name + ".bias", (cout,)))

    def gn(name, c):
        specs.append((name + ".weight", (c,)))
        specs.append((name + ".bias", (c,)))

    enc = "dest_encoder."
    cins = (cfg.input_channels,) + tuple(cfg.dims[:3])
    for s in range(4):
        k = 7 if s == 0 else 3
        conv(f"{enc}patch_embed{s + 1}.proj", cfg.dims[s], cins[s], k, k)
        gn(f"{enc}patch_embed{s + 1}.norm", cfg.dims[s])
    for s in range(4):
        c, sr, hid = cfg.dims[s], cfg.reduction_ratio[s], cfg.dims[s] * cfg.ff_expansion[s]
        for i in range(cfg.depths[s]):
            b = f"{enc}block{s + 1}.{i}."
            gn(b + "norm1", c)
            gn(b + "norm2", c)
            conv(b + "attn.q", c, c, 1)
            conv(b + "attn.k", c, c, 1)
            conv(b + "attn.proj", c, c, 1)
            if sr > 1:
                conv(b + "attn.sr", c, c, sr, sr)
                gn(b + "attn.norm", c)
            conv(b + "mlp1.fc1", hid, c, 1)
            conv(b + "mlp1.dwconv.dwconv", hid, 1, 3, 3)
            conv(b + "mlp1.fc2", c, hid, 1)
            gn(b + "mlp1.norm1", hid)
            gn(b + "mlp1.norm2", hid)

    def conv_layer(name, cin, cout, k):
        conv(name + ".model.0", cout, cin, k, k, bias=False)
        gn(name + ".model.1", cout)

    for j in range(4):
        c = cfg.dims[3 - j]
        conv_layer(f"from_encoder_{j + 1}", c, c, 1)

    def decoder(name, cin):
        for li, (ci, co) in enumerate(short_res_block_plan(cin, MID_CHANNELS)):
            conv_layer(f"{name}.conv.layers.{li}", ci, co, 3)

    for j, cin in enumerate(decoder_in_channels(cfg)):
        decoder(f"depth_upsample.{j}", cin)
    for j, cin in zip((3, 4, 5), head_in_channels(cfg)):
        conv(f"depth_activation_{j}.conv_1", 32, cin, 3, 3)
        conv(f"depth_activation_{j}.conv_2", 1, 32, 3, 3)
    if cfg.supervised_seg or cfg.unsupervised_seg:
        decoder("seg_upsample.0", MID_CHANNELS + 1)
        decoder("seg_upsample.1", MID_CHANNELS + 1 + cfg.input_channels)
    if cfg.supervised_seg:
        conv("seg_conv_stage_4", cfg.num_classes, MID_CHANNELS, 3, 3)
        conv("seg_conv_final", cfg.num_classes, MID_CHANNELS, 3, 3)
    if cfg.unsupervised_seg:
        conv("unsup_stage_4", UNSUP_CLASSES, MID_CHANNELS, 3, 3)
        conv("unsup_final", UNSUP_CLASSES, MID_CHANNELS, 3, 3)
    return specs


def gn_groups(channels: int) -> int:
    return channels // GROUPNORM_DIVISOR
