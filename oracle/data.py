"""CPU oracle for the batch-assembly steps around the hot path. TEST INFRASTRUCTURE ONLY.

Restates the tensor arithmetic of NuscenesDataset.__getitem__ (reference: src/data/dataloader.py:202-333) in numpy.
The reference calls cv2 and scikit-image for the two resizes; neither package is installed in the build container.

Parity status:
  * resize_labels: PINNED to scipy.ndimage.zoom(order=0, grid_mode=True) -- the routine scikit-image 0.19.3's
    `resize(order=0, anti_aliasing=False)` delegates to (requirements.txt pins scikit-image==0.19.3; scipy is here);
    tests/test_oracle_golden.py checks this restatement against scipy itself.
  * resize_image_nearest: PARITY UNPINNED -- restated from OpenCV's published resizeNN (imgproc/resize.cpp:
    x_ofs[x] = min(cvFloor(x * ifx), ssize.width - 1), ifx = 1 / inv_scale_x, inv_scale_x = dsize.width / ssize.width).
"""
import numpy as np


def resize_image_nearest(img, size):
    """cv2.resize(image, size[::-1], interpolation=cv2.INTER_NEAREST) (dataloader.py:227). img [H,W,C] uint8."""
    SH, SW = img.shape[:2]
    DH, DW = size
    ifx, ify = 1.0 / (DW / SW), 1.0 / (DH / SH)
    sx = np.minimum(np.floor(np.arange(DW) * ifx).astype(np.int64), SW - 1)
    sy = np.minimum(np.floor(np.arange(DH) * ify).astype(np.int64), SH - 1)
    return img[sy][:, sx]


def resize_labels(mseg, size, rows=416):
    """skimage.transform.resize(mseg[:rows], size, order=0, preserve_range=True, anti_aliasing=False)
    (dataloader.py:262-267), as integer labels."""
    src = mseg[:rows]
    SH, SW = src.shape
    DH, DW = size
    sy = np.floor(((np.arange(DH) + 0.5) * (SH / DH) - 0.5) + 0.5).astype(np.int64).clip(0, SH - 1)
    sx = np.floor(((np.arange(DW) + 0.5) * (SW / DW) - 0.5) + 0.5).astype(np.int64).clip(0, SW - 1)
    return src[sy][:, sx].astype(np.int64)


def normalise_image(img_u8):
    """ToTensor + Normalize(mean, std) applied to the BGR image as read (dataloader.py:228-233). [H,W,3] -> [3,H,W] fp32."""
    mean = np.array([0.485, 0.456, 0.406], np.float32)
    std = np.array([0.229, 0.224, 0.225], np.float32)
    x = img_u8.astype(np.float32) / np.float32(255.0)
    return np.moveaxis((x - mean) / std, -1, 0)
