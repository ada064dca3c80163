// Batch assembly on the device: the tensor contract of NuscenesDataset.__getitem__ (src/data/dataloader.py:202-333)
// for the radar configuration the reference trains (image + radar depth + radar flow + radial velocity).  SURVEY 8f N1:
// at a few hundred images/s per GPU the reference's 8 CPU workers become the bottleneck; these are small HBM-bound passes.
#include "common.h"

namespace {

constexpr int TPB = 256;

// out[b][0..2] = (img/255 - mean[c]) / std[c] in the channel order the image was read (cv2: BGR -- the reference applies
// the RGB ImageNet constants to it as is, dataloader.py:226-233); out[3] = clip(radar[...,0], 0, max_depth) / max_depth
// (:304-306); out[4..5] = radar[...,1..2] (:309-310); out[6] = rad_vel (:315-318).  img: uint8 [B][H][W][3], radar: fp32
// [B][H][W][3], rad_vel: fp32 [B][H][W] or NULL (then 6 channels).
__global__ __launch_bounds__(TPB) void k_assemble_input(const unsigned char* img, const float* radar, const float* rad_vel,
                                                        long long HW, float max_depth, int channels, float* out) {
  const int b = blockIdx.y;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  for (long long p = (long long)blockIdx.x * TPB + threadIdx.x; p < HW; p += (long long)gridDim.x * TPB) {
    const unsigned char* ip = img + ((long long)b * HW + p) * 3;
    const float* rp = radar + ((long long)b * HW + p) * 3;
    float* o = out + (long long)b * channels * HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * HW] = ((float)ip[c] / 255.f - mean[c]) / stdv[c];
    o[3 * HW] = fminf(fmaxf(rp[0], 0.f), max_depth) / max_depth;
    o[4 * HW] = rp[1];
    o[5 * HW] = rp[2];
    if (rad_vel) o[6 * HW] = rad_vel[(long long)b * HW + p];
  }
}

// Inverse-normalised ground truth (dataloader.py:241-247): g = clip(d, 0, max); g > 0 -> (max - g) / max
__global__ __launch_bounds__(TPB) void k_gt_inverse(const float* depth, long long n, float max_depth, float* out) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    float g = fminf(fmaxf(depth[i], 0.f), max_depth);
    if (g > 0.f) g = (max_depth - g) * (1.f / max_depth);
    out[i] = g;
  }
}

// The reference's `minpool` (dataloader.py:213-222): zeros become 255, -maxpool(-x) with kernel 3, stride 2, padding 1,
// then 255 back to zero -- the minimum over the valid (non-zero) entries of each 3x3 window, 0 if there are none.
__global__ __launch_bounds__(TPB) void k_gt_minpool(const float* src, int H, int W, float* dst) {
  const int b = blockIdx.y;
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const float* s = src + (long long)b * H * W;
  float* d = dst + (long long)b * OH * OW;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < OH * OW; i += gridDim.x * TPB) {
    const int oy = i / OW, ox = i - oy * OW;
    float m = 255.f;
#pragma unroll
    for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
      for (int kx = -1; kx <= 1; ++kx) {
        const int y = 2 * oy + ky, x = 2 * ox + kx;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
          float v = s[(long long)y * W + x];
          if (v == 0.f) v = 255.f;
          m = fminf(m, v);
        }
      }
    d[i] = m == 255.f ? 0.f : m;
  }
}

// cv2.resize(img, (DW, DH), interpolation=cv2.INTER_NEAREST) on interleaved uint8 pixels (dataloader.py:227): source
// column = min(floor(dx * ifx), SW - 1) with ifx = 1 / (DW / SW) in double, rows alike (OpenCV resizeNN).
__global__ __launch_bounds__(TPB) void k_resize_nearest_u8(const unsigned char* src, int SH, int SW, int C, unsigned char* dst,
                                                           int DH, int DW) {
  const int b = blockIdx.y;
  const double ifx = 1.0 / ((double)DW / (double)SW), ify = 1.0 / ((double)DH / (double)SH);
  const long long total = (long long)DH * DW;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int dy = (int)(i / DW), dx = (int)(i - (long long)dy * DW);
    int sy = (int)floor(dy * ify), sx = (int)floor(dx * ifx);
    sy = sy < SH - 1 ? sy : SH - 1;
    sx = sx < SW - 1 ? sx : SW - 1;
    const unsigned char* s = src + (((long long)b * SH + sy) * SW + sx) * C;
    unsigned char* d = dst + ((long long)b * total + i) * C;
    for (int c = 0; c < C; ++c) d[c] = s[c];
  }
}

// skimage.transform.resize(mseg[:rows], (DH, DW), order=0, preserve_range=True, anti_aliasing=False) (dataloader.py:262-267;
// scikit-image 0.19.3 = scipy.ndimage.zoom(order=0, grid_mode=True)): source index = floor(((o + 0.5) * (S / D) - 0.5) + 0.5)
// in double.  uint8 label maps in, int64 labels out (what the loss consumes, runner.py:189-190).
__global__ __launch_bounds__(TPB) void k_resize_labels(const unsigned char* src, int SH_full, int SH, int SW, long long* dst,
                                                       int DH, int DW) {
  const int b = blockIdx.y;
  const double zy = (double)SH / (double)DH, zx = (double)SW / (double)DW;
  const long long total = (long long)DH * DW;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int dy = (int)(i / DW), dx = (int)(i - (long long)dy * DW);
    int sy = (int)floor(((dy + 0.5) * zy - 0.5) + 0.5), sx = (int)floor(((dx + 0.5) * zx - 0.5) + 0.5);
    sy = sy < 0 ? 0 : (sy < SH ? sy : SH - 1);
    sx = sx < 0 ? 0 : (sx < SW ? sx : SW - 1);
    dst[(long long)b * total + i] = src[((long long)b * SH_full + sy) * SW + sx];
  }
}

inline int blocks_for(long long total, int cap = 1024) {
  long long n = (total + TPB - 1) / TPB;
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return (int)n;
}

}  // namespace

extern "C" int crd_assemble_input(const void* img_u8, const float* radar, const float* rad_vel, int32_t B, int32_t H, int32_t W,
                                  float max_depth, float* out, crd_stream_t stream) {
  CRD_CHECK_ARG(img_u8 && radar && out && B > 0 && H > 0 && W > 0 && max_depth > 0.f, "crd_assemble_input: bad argument");
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(k_assemble_input, dim3(blocks_for(HW), B), dim3(TPB), 0, as_stream(stream),
                     reinterpret_cast<const unsigned char*>(img_u8), radar, rad_vel, HW, max_depth, rad_vel ? 7 : 6, out);
  CRD_LAUNCH_CHECK("crd_assemble_input");
  return CRD_OK;
}

extern "C" int crd_gt_pyramid(const float* depth, int32_t B, int32_t H, int32_t W, float max_depth, float* full, float* half,
                              float* quarter, float* eighth, crd_stream_t stream) {
  CRD_CHECK_ARG(depth && full && B > 0 && H > 0 && W > 0 && max_depth > 0.f, "crd_gt_pyramid: bad argument");
  CRD_CHECK_ARG(!(quarter && !half) && !(eighth && !quarter), "crd_gt_pyramid: a level needs the one above it");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_gt_inverse, dim3(blocks_for((long long)B * H * W)), dim3(TPB), 0, st, depth, (long long)B * H * W, max_depth, full);
  const float* src = full;
  float* lv[3] = {half, quarter, eighth};
  int h = H, w = W;
  for (int i = 0; i < 3 && lv[i]; ++i) {
    const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(k_gt_minpool, dim3(blocks_for((long long)oh * ow, 256), B), dim3(TPB), 0, st, src, h, w, lv[i]);
    src = lv[i]; h = oh; w = ow;
  }
  CRD_LAUNCH_CHECK("crd_gt_pyramid");
  return CRD_OK;
}

extern "C" int crd_resize_nearest_u8(const void* src, int32_t B, int32_t SH, int32_t SW, int32_t C, void* dst, int32_t DH, int32_t DW,
                                     crd_stream_t stream) {
  CRD_CHECK_ARG(src && dst && B > 0 && SH > 0 && SW > 0 && C > 0 && DH > 0 && DW > 0, "crd_resize_nearest_u8: bad argument");
  hipLaunchKernelGGL(k_resize_nearest_u8, dim3(blocks_for((long long)DH * DW), B), dim3(TPB), 0, as_stream(stream),
                     reinterpret_cast<const unsigned char*>(src), SH, SW, C, reinterpret_cast<unsigned char*>(dst), DH, DW);
  CRD_LAUNCH_CHECK("crd_resize_nearest_u8");
  return CRD_OK;
}

extern "C" int crd_resize_labels_nearest(const void* src_u8, int32_t B, int32_t SH, int32_t SW, int32_t rows, int64_t* dst, int32_t DH,
                                         int32_t DW, crd_stream_t stream) {
  CRD_CHECK_ARG(src_u8 && dst && B > 0 && SH > 0 && SW > 0 && rows > 0 && DH > 0 && DW > 0, "crd_resize_labels_nearest: bad argument");
  hipLaunchKernelGGL(k_resize_labels, dim3(blocks_for((long long)DH * DW), B), dim3(TPB), 0, as_stream(stream),
                     reinterpret_cast<const unsigned char*>(src_u8), SH, rows < SH ? rows : SH, SW, reinterpret_cast<long long*>(dst), DH, DW);
  CRD_LAUNCH_CHECK("crd_resize_labels_nearest");
  return CRD_OK;
}
