"""GPU-side batch assembly: the tensor contract of the reference dataloader (src/data/dataloader.py:202-333) for the
radar configuration -- 7-channel input, inverse-normalised LiDAR ground truth and its zero-ignoring min-pool pyramid --
from raw device buffers (uint8 image as cv2 reads it, radar [H,W,3], radial velocity, LiDAR depth in metres).  File
decoding, the nearest-neighbour image resize and the segmentation resize stay on the host (no cv2 / skimage here)."""
import torch

from . import lib as L


def assemble_batch(img_u8, radar, rad_vel, gt_depth, max_depth=100.0, levels=3):
    """img_u8 [B,H,W,3] uint8, radar [B,H,W,3] fp32, rad_vel [B,H,W] fp32 or None, gt_depth [B,H,W] fp32 (metres), all on
    the GPU.  Returns {'image': [B,7,H,W], 'gt_full': [B,1,H,W], 'gt_half', 'gt_quarter'[, 'gt_eighth']} like
    camradepth_amd.synth.make_batch / the reference's batch dictionary."""
    if not img_u8.is_cuda:
        raise L.CrdError("assemble_batch runs on the GPU (no CPU fallback)")
    lib = L.load()
    B, H, W, _ = img_u8.shape
    dev = img_u8.device
    img_u8, radar, gt_depth = img_u8.contiguous(), radar.contiguous().float(), gt_depth.contiguous().float()
    rv = rad_vel.contiguous().float() if rad_vel is not None else None
    x = torch.empty(B, 7 if rv is not None else 6, H, W, device=dev)
    L.check(lib.crd_assemble_input(img_u8.data_ptr(), radar.data_ptr(), rv.data_ptr() if rv is not None else None, B, H, W,
                                   float(max_depth), x.data_ptr(), L.stream()), "crd_assemble_input")
    names = ["gt_full", "gt_half", "gt_quarter", "gt_eighth"][:levels + 1]
    maps, h, w = [], H, W
    for _ in names:
        maps.append(torch.empty(B, 1, h, w, device=dev))
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    ptrs = [m.data_ptr() for m in maps] + [None] * (4 - len(maps))
    L.check(lib.crd_gt_pyramid(gt_depth.data_ptr(), B, H, W, float(max_depth), *ptrs, L.stream()), "crd_gt_pyramid")
    out = {"image": x}
    out.update(dict(zip(names, maps)))
    return out
