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


def resize_image_nearest(img_u8, size):
    """cv2.resize(image, size[::-1], interpolation=cv2.INTER_NEAREST) of dataloader.py:227 for a uint8 cuda batch
    [B,H,W,C]; size = (H_out, W_out) like args.image_dimension."""
    if not img_u8.is_cuda or img_u8.dtype != torch.uint8:
        raise L.CrdError("resize_image_nearest takes a uint8 cuda tensor [B,H,W,C] (no CPU fallback)")
    B, H, W, Cc = img_u8.shape
    out = torch.empty(B, size[0], size[1], Cc, dtype=torch.uint8, device=img_u8.device)
    L.check(L.load().crd_resize_nearest_u8(img_u8.contiguous().data_ptr(), B, H, W, Cc, out.data_ptr(), size[0], size[1], L.stream()),
            "crd_resize_nearest_u8")
    return out


def seg_targets(mseg_u8, rows=416, sizes=((416, 800), (208, 400))):
    """The two segmentation targets of dataloader.py:262-267 from uint8 cuda label maps [B,H,W]: the first `rows` rows,
    nearest-resized to each size (skimage order 0), as int64 -- {'final_seg', 'intermediate_seg'} of the batch dict."""
    if not mseg_u8.is_cuda or mseg_u8.dtype != torch.uint8:
        raise L.CrdError("seg_targets takes a uint8 cuda tensor [B,H,W] (no CPU fallback)")
    B, H, W = mseg_u8.shape
    src = mseg_u8.contiguous()
    outs = []
    for (h, w) in sizes:
        o = torch.empty(B, h, w, dtype=torch.int64, device=src.device)
        L.check(L.load().crd_resize_labels_nearest(src.data_ptr(), B, H, W, rows, o.data_ptr(), h, w, L.stream()),
                "crd_resize_labels_nearest")
        outs.append(o)
    return dict(zip(("final_seg", "intermediate_seg"), outs))
