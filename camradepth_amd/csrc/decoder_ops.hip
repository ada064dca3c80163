// Decoder-side streaming kernels for gfx950: bicubic x2 (forward / transpose), layout conversion at
// the module boundary, Seg_Block argmax, slice copies and small elementwise helpers.
#include "common.h"

namespace {

constexpr int TPB = 256;
// PyTorch upsample_bicubic2d, A = -0.75, scale 2, align_corners=False:
// odd output o=2k+1 reads inputs k-1..k+2 with WO, even output o=2k reads k-2..k+1 with WE.
__device__ __constant__ float WO[4] = {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f};
__device__ __constant__ float WE[4] = {-0.03515625f, 0.26171875f, 0.87890625f, -0.10546875f};

__device__ __forceinline__ void taps_of(int o, int n, int (&idx)[4], float (&w)[4]) {
  const int k = o >> 1;
  const bool odd = o & 1;
  const int base = odd ? k - 1 : k - 2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int i = base + t;
    idx[t] = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    w[t] = odd ? WO[t] : WE[t];
  }
}

// One thread produces the 2 x 2 outputs above input pixel (k, m) for 8 channels: they share the clamped 5 x 5 input
// neighbourhood (25 loads instead of 4 x 16), horizontal pass first (even / odd column results of the five rows), then
// the vertical one -- the same order of operations as the one-output-per-thread form it replaces.
// F8 = 1: the output is the e4m3 copy (y = bytes, y_ld in bytes) of the bf16 result, scaled by inv_scale; 2: that and the
// bf16 result itself (y2)
template <int F8>
__global__ __launch_bounds__(TPB) void k_bicubic(const bf16_t* x, int x_ld, int H, int W, int C, void* y, int y_ld, float inv_scale,
                                                 void* y2, int y2_ld) {
  const int b = blockIdx.y;
  const int CG = C >> 3;
  const int OW = 2 * W;
  const long long total = (long long)H * W * CG;
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const int pix = (int)(i / CG);
  const int k = pix / W, m = pix - k * W;
  const bf16_t* xb = x + (long long)b * H * W * x_ld + cg * 8;
  int cx[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) { const int ix = m - 2 + c; cx[c] = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix); }
  float he[5][8], ho[5][8];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    int iy = k - 2 + r;
    iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
    float v[5][8];
#pragma unroll
    for (int c = 0; c < 5; ++c) load8(xb, ((long long)iy * W + cx[c]) * x_ld, 0, v[c]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { he[r][j] = 0.f; ho[r][j] = 0.f; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) { he[r][j] += WE[t] * v[t][j]; ho[r][j] += WO[t] * v[t + 1][j]; }
  }
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int ox = 0; ox < 2; ++ox) {
      float out[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) out[j] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float wy = oy ? WO[t] : WE[t];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] += wy * (ox ? ho[t + oy][j] : he[t + oy][j]);
      }
      const long long opix = (long long)b * 2 * H * OW + (long long)(2 * k + oy) * OW + 2 * m + ox;
      if (F8) store8_fp8(y, opix * y_ld + cg * 8, out, inv_scale);
      else store8_bf16(y, opix * y_ld + cg * 8, out);
      if (F8 == 2) store8_bf16(y2, opix * y2_ld + cg * 8, out);
    }
}

// transpose: dx[iy][ix] (+)= sum_{oy,ox} By[oy][iy]*Bx[ox][ix]*dy[oy][ox]; candidates o in [2i-4, 2i+5]
__device__ __forceinline__ void bwd_weights(int i, int n, float (&w)[10]) {
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const int o = 2 * i - 4 + k;
    float acc = 0.f;
    if (o >= 0 && o < 2 * n) {
      int idx[4];
      float ww[4];
      taps_of(o, n, idx, ww);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (idx[t] == i) acc += ww[t];
    }
    w[k] = acc;
  }
}

__global__ __launch_bounds__(TPB) void k_bicubic_bwd(const bf16_t* dy, int dy_ld, int H, int W, int C, bf16_t* dx, int dx_ld,
                                                     int accumulate) {
  const int b = blockIdx.y;
  const int CG = C >> 3;
  const int OH = 2 * H, OW = 2 * W;
  const long long total = (long long)H * W * CG;
  const long long i = (long long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const int pix = (int)(i / CG);
  const int iy = pix / W, ix = pix - iy * W;
  float wy[10], wx[10];
  bwd_weights(iy, H, wy);
  bwd_weights(ix, W, wx);
  const bf16_t* db = dy + (long long)b * OH * OW * dy_ld + cg * 8;
  float out[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = 0.f;
  // border pixels collect clamped taps from further away: widen the window there
  const bool border = iy < 2 || iy > H - 3 || ix < 2 || ix > W - 3;
  if (!border) {
#pragma unroll
    for (int a = 1; a < 9; ++a) {
      const int oy = 2 * iy - 4 + a;
#pragma unroll
      for (int c = 1; c < 9; ++c) {
        const int ox = 2 * ix - 4 + c;
        float v[8];
        load8(db, ((long long)oy * OW + ox) * dy_ld, 0, v);
        const float w = wy[a] * wx[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] += w * v[j];
      }
    }
  } else {
    for (int a = 0; a < 10; ++a) {
      const int oy = 2 * iy - 4 + a;
      if (oy < 0 || oy >= OH || wy[a] == 0.f) continue;
      for (int c = 0; c < 10; ++c) {
        const int ox = 2 * ix - 4 + c;
        if (ox < 0 || ox >= OW || wx[c] == 0.f) continue;
        float v[8];
        load8(db, ((long long)oy * OW + ox) * dy_ld, 0, v);
        const float w = wy[a] * wx[c];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] += w * v[j];
      }
    }
  }
  const long long off = ((long long)b * H * W + pix) * dx_ld + cg * 8;
  if (accumulate) {
    float o[8];
    load8(dx, off, 0, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] += o[j];
  }
  store8_bf16(dx, off, out);
}

__global__ __launch_bounds__(TPB) void k_nchw_to_pm(const float* x, int C, long long HW, bf16_t* y, int y_ld, int Cpad) {
  const int b = blockIdx.y;
  const int CG = Cpad >> 3;
  const long long total = HW * CG;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long p = i % HW;          // pixel fastest: coalesced reads of each channel plane
    const int cg = (int)(i / HW);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cg * 8 + j;
      v[j] = c < C ? x[((long long)b * C + c) * HW + p] : 0.f;
    }
    store8_bf16(y, ((long long)b * HW + p) * y_ld + cg * 8, v);
  }
}

__global__ __launch_bounds__(TPB) void k_pm_to_nchw(const void* x, int x_f32, int x_ld, int C, long long HW, float* y) {
  const int b = blockIdx.y;
  const long long total = HW * C;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long p = i % HW;
    const int c = (int)(i / HW);
    const long long off = ((long long)b * HW + p) * x_ld + c;
    y[((long long)b * C + c) * HW + p] = x_f32 ? reinterpret_cast<const float*>(x)[off] : bf2f(reinterpret_cast<const bf16_t*>(x)[off]);
  }
}

__global__ __launch_bounds__(TPB) void k_scale_f32(const float* src, float* dst, long long n, float scale) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) dst[i] = scale * src[i];
}

__global__ __launch_bounds__(TPB) void k_seg_argmax(const float* logits, int ld, long long rows, int C, int num_classes,
                                                    void* y, int y_f32, int y_ld) {
  for (long long r = (long long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long long)gridDim.x * TPB) {
    const float* p = logits + r * ld;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      float v = p[c];
      if (v > best) { best = v; bi = c; }
    }
    const float v = (float)bi / (float)num_classes;
    if (y_f32) reinterpret_cast<float*>(y)[r * y_ld] = v;
    else reinterpret_cast<bf16_t*>(y)[r * y_ld] = f2bf(v);
  }
}

__global__ __launch_bounds__(TPB) void k_slice_copy(const bf16_t* src, int s_ld, bf16_t* dst, int d_ld, long long rows, int C,
                                                    int accumulate) {
  const int CG = C >> 3;
  const long long total = rows * CG;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int cg = (int)(i % CG);
    const long long r = i / CG;
    uint4 u = *reinterpret_cast<const uint4*>(src + r * s_ld + cg * 8);
    if (accumulate) {
      float a[8], o[8];
      load8(src, r * s_ld + cg * 8, 0, a);
      load8(dst, r * d_ld + cg * 8, 0, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += o[j];
      store8_bf16(dst, r * d_ld + cg * 8, a);
    } else {
      *reinterpret_cast<uint4*>(dst + r * d_ld + cg * 8) = u;
    }
  }
}

// dst[r][c] = bf16(scale[r / rows_per_sample] * src[r*s_ld + c] + add[r*add_ld + c]) for c < C
__global__ __launch_bounds__(TPB) void k_f32_to_bf16_rows(const float* src, int s_ld, bf16_t* dst, int d_ld, long long rows,
                                                          int C, const float* scale, long long rows_per_sample,
                                                          const bf16_t* add, int add_ld, int vec) {
  if (vec) {  // C % 8 == 0 and all strides / offsets 8-aligned
    const int CG = C >> 3;
    const long long total = rows * CG;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
      const int cg = (int)(i % CG);
      const long long r = i / CG;
      const float s = scale ? scale[r / rows_per_sample] : 1.f;
      float v[8];
      load8(src, r * s_ld + cg * 8, 1, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= s;
      if (add) {
        float a[8];
        load8(add, r * add_ld + cg * 8, 0, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += a[j];
      }
      store8_bf16(dst, r * d_ld + cg * 8, v);
    }
    return;
  }
  const long long total = rows * C;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int c = (int)(i % C);
    const long long r = i / C;
    float s = scale ? scale[r / rows_per_sample] : 1.f;
    float v = s * src[r * s_ld + c];
    if (add) v += bf2f(add[r * add_ld + c]);
    dst[r * d_ld + c] = f2bf(v);
  }
}

// dz = da * a * (1-a), in place on da (bf16 contiguous)
__global__ __launch_bounds__(TPB) void k_sigmoid_bwd(const bf16_t* a, bf16_t* da, long long n8) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n8; i += (long long)gridDim.x * TPB) {
    float av[8], dv[8];
    load8(a, i * 8, 0, av);
    load8(da, i * 8, 0, dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) dv[j] *= av[j] * (1.f - av[j]);
    store8_bf16(da, i * 8, dv);
  }
}

// ------------------------------------------------------------------------------------------------
// Depth_Activation.conv_2 (utils.py:283,288): 3x3, 32 -> 1 channel.  A GEMM tile would be 1/32 full, so this is a
// stencil-reduce: 4 lanes per pixel (8 channels each), weights in registers, HBM-bound on the 32-channel input.
// w is the reference layout fp32 [1][32][3][3] (index c*9 + tap), rounded to bf16 on load like autocast does.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_head2_fwd(const bf16_t* a, const float* w, const float* bias, int H, int W,
                                                   float* depth, bf16_t* copy, int copy_ld) {
  const int b = blockIdx.y, q = threadIdx.x & 3;
  // the 72 weights of this lane's 8 channels: loaded once, the thread then walks over many pixels (one pixel per thread
  // spent 8x more loads on the weights than on the data)
  float wv[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[t][j] = bf_round(w[(q * 8 + j) * 9 + t]);
  const float bias0 = bias[0];
  const bf16_t* ab = a + (long long)b * H * W * 32 + q * 8;
  const int npix = H * W, stride = (gridDim.x * TPB) >> 2;
  for (int pix = (blockIdx.x * TPB + threadIdx.x) >> 2; pix < npix; pix += stride) {     // uniform per 4-lane pixel group
    const int py = pix / W, px = pix - py * W;
    // the nine taps from clamped addresses, all in flight, zeroed by a select outside the image (under `if (inside) load` every
    // tap's load was waited for before the next was issued: nine dependent memory latencies per pixel)
    uint4 tv[9];
    bool in[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
      in[t] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int cy = iy < 0 ? 0 : (iy < H ? iy : H - 1), cx = ix < 0 ? 0 : (ix < W ? ix : W - 1);
      tv[t] = *reinterpret_cast<const uint4*>(ab + ((long long)cy * W + cx) * 32);
    }
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const uint4 u = in[t] ? tv[t] : make_uint4(0, 0, 0, 0);
      acc += bf_lo(u.x) * wv[t][0]; acc += bf_hi(u.x) * wv[t][1]; acc += bf_lo(u.y) * wv[t][2]; acc += bf_hi(u.y) * wv[t][3];
      acc += bf_lo(u.z) * wv[t][4]; acc += bf_hi(u.z) * wv[t][5]; acc += bf_lo(u.w) * wv[t][6]; acc += bf_hi(u.w) * wv[t][7];
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (q == 0) {
      const float r = bf_round(acc + bias0);
      depth[(long long)b * npix + pix] = r;
      if (copy) copy[((long long)b * npix + pix) * copy_ld] = f2bf(r);
    }
  }
}

// dz[p][c] = a(1-a) * sum_tap dy[p - off(tap)] * w[c][tap],  dy = gd (+ add)
__global__ __launch_bounds__(TPB) void k_head2_bwd_data(const float* gd, const bf16_t* add, int add_ld, const bf16_t* a,
                                                        const float* w, int H, int W, bf16_t* dz) {
  const int b = blockIdx.y, q = threadIdx.x & 3;
  float wv[9][8];                                           // loaded once per thread, as in k_head2_fwd
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[t][j] = bf_round(w[(q * 8 + j) * 9 + t]);
  const float* gb = gd + (long long)b * H * W;
  const bf16_t* addb = add ? add + (long long)b * H * W * add_ld : nullptr;
  const int npix = H * W, stride = (gridDim.x * TPB) >> 2;
  for (int pix = (blockIdx.x * TPB + threadIdx.x) >> 2; pix < npix; pix += stride) {
    const int py = pix / W, px = pix - py * W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // (the nine taps from clamped addresses, all in flight, zeroed by a select outside the image: as in k_head2_fwd)
    float dyv[9];
    long long oc[9];
    bool in[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int oy = py + 1 - t / 3, ox = px + 1 - t % 3;
      in[t] = (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
      const int cy = oy < 0 ? 0 : (oy < H ? oy : H - 1), cx = ox < 0 ? 0 : (ox < W ? ox : W - 1);
      oc[t] = (long long)cy * W + cx;
      dyv[t] = gb[oc[t]];
    }
    if (addb) {                                             // (kernel argument: uniform)
      bf16_t av9[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) av9[t] = addb[oc[t] * add_ld];
#pragma unroll
      for (int t = 0; t < 9; ++t) dyv[t] += bf2f(av9[t]);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float dv = in[t] ? bf_round(dyv[t]) : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += dv * wv[t][j];
    }
    float av[8];
    const long long off = ((long long)b * npix + pix) * 32 + q * 8;
    load8(a, off, 0, av);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= av[j] * (1.f - av[j]);
    store8_bf16(dz, off, acc);
  }
}

// rows[r][c*9 + tap] += sum_p dy[p] * a[p + off(tap)][c] ; rows[r][288] += sum_p dy[p], r = workgroup % replicas.
// The 289 sums of a workgroup are folded with shuffles (pixel lanes of a wave) and four LDS rounds (waves); with one
// accumulator for all 2048 workgroups the chain of contended atomics (2048 x 32 per 128-byte line x 2.6 ns) was the
// whole 170 us of this kernel, hence the copies (crd_wgrad_unpack sums them).
__global__ __launch_bounds__(TPB) void k_head2_wgrad(const float* gd, const bf16_t* add, int add_ld, const bf16_t* a, int H, int W,
                                                     int chunk, crd_sum_t* rows, int replicas) {
  __shared__ float sm[9 * 32 + 1];
  const int b = blockIdx.y, q = threadIdx.x & 3, pl = threadIdx.x >> 2;
  const float* gb = gd + (long long)b * H * W;
  const bf16_t* addb = add ? add + (long long)b * H * W * add_ld : nullptr;
  const bf16_t* ab = a + (long long)b * H * W * 32 + q * 8;
  float acc[9][8], bs = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  int p0 = blockIdx.x * chunk, p1 = p0 + chunk;
  if (p1 > H * W) p1 = H * W;
  for (int pix = p0 + pl; pix < p1; pix += TPB / 4) {
    const int py = pix / W, px = pix - py * W;
    float dyv = gb[pix];
    if (addb) dyv += bf2f(addb[(long long)pix * add_ld]);
    dyv = bf_round(dyv);
    bs += dyv;
    // all nine taps requested before the first is used: unconditional loads from a clamped address, the tap's weight selected
    // (under `if (in the image) load` every tap was waited for before the next was issued: 94 us for 27 MB at 256 x 416 x 8)
    uint4 raw[9];
    float wt[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = py + ky - 1, ix = px + kx - 1;
        const int cy = iy < 0 ? 0 : (iy < H ? iy : H - 1), cx = ix < 0 ? 0 : (ix < W ? ix : W - 1);
        raw[ky * 3 + kx] = *reinterpret_cast<const uint4*>(ab + ((long long)cy * W + cx) * 32);
        wt[ky * 3 + kx] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? dyv : 0.f;
      }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const uint4 u = raw[t];
      const float v[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] += wt[t] * v[j];
    }
  }
  // pixel lanes of the wave (lane bits 2..5)
  for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][j] += __shfl_xor(acc[t][j], o);
    bs += __shfl_xor(bs, o);
  }
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int w = 0; w < TPB / 64; ++w) {
    if (wave == w && l < 4) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float* p = &sm[t * 32 + q * 8 + j];
          *p = w == 0 ? acc[t][j] : *p + acc[t][j];
        }
      if (l == 0) sm[288] = w == 0 ? bs : sm[288] + bs;
    }
    __syncthreads();
  }
  crd_sum_t* dst = rows + (long long)((blockIdx.y * gridDim.x + blockIdx.x) % replicas) * 289;
  for (int i = threadIdx.x; i < 288; i += TPB) grad_add(&dst[(i & 31) * 9 + (i >> 5)], sm[i]);
  if (threadIdx.x == 0) grad_add(&dst[288], sm[288]);
}

// LDS-tiled form of k_bicubic_bwd: a workgroup owns an 8 x 16 tile of input pixels and a 16-channel window (32 channels: 60 KB
// of LDS, two workgroups per CU, 206 instead of 168 us on the 256 x 416 level); the (2*8+8) x (2*16+8) region of dy that feeds it
// is loaded once (coalesced, all loads in flight).  The gather kernel above fetched each dy element ~16 times through L1/L2 and
// 7 times from HBM (rocprofv3 FETCH_SIZE: 421 MB per launch for 233 MB of dy).
constexpr int BTH = 8, BTW = 16, BCG = 2;                 // tile rows / columns, granules per window
constexpr int BRH = 2 * BTH + 8, BRW = 2 * BTW + 8;       // dy region
__global__ __launch_bounds__(TPB) void k_bicubic_bwd_tile(const bf16_t* dy, int dy_ld, int H, int W, int C, bf16_t* dx, int dx_ld,
                                                          int accumulate, int tiles_x) {
  __shared__ __attribute__((aligned(16))) uint4 sdy[BRH * BRW * BCG];        // 30 KB: five workgroups per CU
  const int b = blockIdx.z;
  const int g0 = blockIdx.y * BCG;                          // first granule of the window
  const int CG = C >> 3;
  const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
  const int y0 = tyi * BTH, x0 = txi * BTW;
  const int OH = 2 * H, OW = 2 * W;
  const bf16_t* db = dy + (long long)b * OH * OW * dy_ld;
  const int t = threadIdx.x;
  const int g = t & (BCG - 1), xc = (t >> 1) & (BTW - 1), rgp = t >> 5;      // 2 granules x 16 columns x 8 rows
  const int ix = x0 + xc, iy = y0 + rgp;
  const bool colok = g0 + g < CG && ix < W;
  // the tap weights first, pinned in registers (left to the scheduler, the branchy weight code ended up interleaved with the passes)
  float wx[10], wy[10];
  bwd_weights(ix < W ? ix : W - 1, W, wx);
  bwd_weights(iy < H ? iy : H - 1, H, wy);
#pragma unroll
  for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(wx[i]), "+v"(wy[i]));
  {
    constexpr int NP = (BRH * BRW * BCG + TPB - 1) / TPB;
    uint4 r[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = t + k * TPB;
      const int px = i / BCG, gg = i - px * BCG;
      const int ry = px / BRW, rx = px - ry * BRW;
      const int oy = 2 * y0 - 4 + ry, ox = 2 * x0 - 4 + rx;
      // (unconditional loads from a clamped address + select: no early wait behind the first piece)
      const bool ok = i < BRH * BRW * BCG && g0 + gg < CG && (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
      const int cy = oy < 0 ? 0 : (oy < OH ? oy : OH - 1), cx = ox < 0 ? 0 : (ox < OW ? ox : OW - 1), cgr = g0 + gg < CG ? g0 + gg : g0;
      const uint4 u = *reinterpret_cast<const uint4*>(db + ((long long)cy * OW + cx) * dy_ld + cgr * 8);
      r[k] = ok ? u : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int i = t + k * TPB;
      if (i < BRH * BRW * BCG) sdy[i] = r[k];
    }
  }
  __syncthreads();
  // The transposed stencil is separable -- dx = By^T (dy Bx) -- and is evaluated that way: a horizontal pass over the region's 24
  // rows (10 taps each, results kept in registers), then, with the fp32 row sums written back over the dy tile in LDS, a vertical
  // pass (10 taps).  The dense 10 x 10 form did 100 LDS reads + ~1700 vector operations per (pixel, granule): 276 us on the
  // 256 x 416 level against ~85 us of memory time.  (Compiled with -fno-slp-vectorize, build.py: the SLP vectoriser pairs the
  // per-channel FMAs and keeps every unpacked operand alive for it -- 452 registers instead of ~100.)
  constexpr int RPT = BRH / 8;                                               // region rows per thread in the horizontal pass: 3
  float hs[RPT][8];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int ry = k * 8 + rgp;
#pragma unroll
    for (int j = 0; j < 8; ++j) hs[k][j] = 0.f;
#pragma unroll
    for (int c = 0; c < 10; ++c) {
      const uint4 u = sdy[(ry * BRW + 2 * xc + c) * BCG + g];
      const float w = wx[c];
      hs[k][0] += w * bf_lo(u.x); hs[k][1] += w * bf_hi(u.x); hs[k][2] += w * bf_lo(u.y); hs[k][3] += w * bf_hi(u.y);
      hs[k][4] += w * bf_lo(u.z); hs[k][5] += w * bf_hi(u.z); hs[k][6] += w * bf_lo(u.w); hs[k][7] += w * bf_hi(u.w);
    }
  }
  __syncthreads();                                                           // every read of the dy tile is done
  float4* srow = reinterpret_cast<float4*>(sdy);                             // [2 halves][BRH][BTW][BCG] float4: 24 KB of the tile's 30
  constexpr int HALF = BRH * BTW * BCG;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int ry = k * 8 + rgp;
    float4* p = srow + (ry * BTW + xc) * BCG + g;
    p[0] = make_float4(hs[k][0], hs[k][1], hs[k][2], hs[k][3]);
    p[HALF] = make_float4(hs[k][4], hs[k][5], hs[k][6], hs[k][7]);
  }
  __syncthreads();
  if (!colok || iy >= H) return;
  float out[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = 0.f;
#pragma unroll
  for (int a = 0; a < 10; ++a) {
    const float4* p = srow + ((2 * rgp + a) * BTW + xc) * BCG + g;
    const float4 lo = p[0], hi = p[HALF];
    const float w = wy[a];
    out[0] += w * lo.x; out[1] += w * lo.y; out[2] += w * lo.z; out[3] += w * lo.w;
    out[4] += w * hi.x; out[5] += w * hi.y; out[6] += w * hi.z; out[7] += w * hi.w;
  }
  const long long off = ((long long)b * H * W + (long long)iy * W + ix) * dx_ld + (g0 + g) * 8;
  if (accumulate) {
    float o[8];
    load8(dx, off, 0, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] += o[j];
  }
  store8_bf16(dx, off, out);
}

// workgroups per sample of the head stencil kernels: 4 lanes per pixel, ~8+ pixels per thread on large grids
inline int head2_blocks(int H, int W, int B) {
  long long n = cdiv(4ll * H * W, TPB);
  long long cap = 4096 / (B > 0 ? B : 1);
  if (cap < 1) cap = 1;
  return (int)(n > cap ? cap : n);
}

inline int blocks_for(long long total) {
  long long n = (total + TPB - 1) / TPB;
  if (n > 4096) n = 4096;
  if (n < 1) n = 1;
  return (int)n;
}

}  // namespace

extern "C" int crd_bicubic2x(const void* x, int32_t x_ld, int32_t x_coff, int32_t B, int32_t H, int32_t W, int32_t C, void* y,
                             int32_t y_ld, int32_t y_coff, crd_stream_t stream) {
  CRD_CHECK_ARG(x && y, "crd_bicubic2x: null pointer");
  CRD_CHECK_ARG(C % 8 == 0 && x_ld % 8 == 0 && x_coff % 8 == 0 && y_ld % 8 == 0 && y_coff % 8 == 0, "crd_bicubic2x: alignment");
  const long long total = 4ll * H * W * (C / 8);
  hipLaunchKernelGGL(k_bicubic<0>, dim3((unsigned)cdiv(total / 4, TPB), B), dim3(TPB), 0, as_stream(stream),
                     reinterpret_cast<const bf16_t*>(x) + x_coff, x_ld, H, W, C, reinterpret_cast<bf16_t*>(y) + y_coff, y_ld, 1.f, nullptr, 0);
  CRD_LAUNCH_CHECK("crd_bicubic2x");
  return CRD_OK;
}

extern "C" int crd_bicubic2x_fp8(const void* x, int32_t x_ld, int32_t x_coff, int32_t B, int32_t H, int32_t W, int32_t C, void* y_fp8,
                                 int32_t y_ld, int32_t y_coff, float y_scale, void* y_bf16, int32_t yb_ld, int32_t yb_coff,
                                 crd_stream_t stream) {
  CRD_CHECK_ARG(x && y_fp8 && y_scale > 0.f, "crd_bicubic2x_fp8: null pointer / bad scale");
  CRD_CHECK_ARG(C % 8 == 0 && x_ld % 8 == 0 && x_coff % 8 == 0 && y_ld % 8 == 0 && y_coff % 8 == 0 && yb_ld % 8 == 0 && yb_coff % 8 == 0,
                "crd_bicubic2x_fp8: alignment");
  const long long total = 4ll * H * W * (C / 8);
  const dim3 grid((unsigned)cdiv(total / 4, TPB), B);
  if (y_bf16)
    hipLaunchKernelGGL(k_bicubic<2>, grid, dim3(TPB), 0, as_stream(stream), reinterpret_cast<const bf16_t*>(x) + x_coff, x_ld, H, W, C,
                       reinterpret_cast<unsigned char*>(y_fp8) + y_coff, y_ld, 1.f / y_scale, reinterpret_cast<bf16_t*>(y_bf16) + yb_coff, yb_ld);
  else
    hipLaunchKernelGGL(k_bicubic<1>, grid, dim3(TPB), 0, as_stream(stream), reinterpret_cast<const bf16_t*>(x) + x_coff, x_ld, H, W, C,
                       reinterpret_cast<unsigned char*>(y_fp8) + y_coff, y_ld, 1.f / y_scale, nullptr, 0);
  CRD_LAUNCH_CHECK("crd_bicubic2x_fp8");
  return CRD_OK;
}

extern "C" int crd_bicubic2x_bwd(const void* dy, int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                                 void* dx, int32_t dx_ld, int32_t dx_coff, int32_t accumulate, crd_stream_t stream) {
  CRD_CHECK_ARG(dy && dx, "crd_bicubic2x_bwd: null pointer");
  CRD_CHECK_ARG(C % 8 == 0 && dy_ld % 8 == 0 && dy_coff % 8 == 0 && dx_ld % 8 == 0 && dx_coff % 8 == 0, "crd_bicubic2x_bwd: alignment");
  if (H >= BTH && W >= BTW) {      // LDS-tiled kernel; tiny maps keep the gather kernel
    const int tiles_x = cdiv(W, BTW), tiles_y = cdiv(H, BTH);
    hipLaunchKernelGGL(k_bicubic_bwd_tile, dim3(tiles_x * tiles_y, cdiv(C / 8, BCG), B), dim3(TPB), 0, as_stream(stream),
                       reinterpret_cast<const bf16_t*>(dy) + dy_coff, dy_ld, H, W, C, reinterpret_cast<bf16_t*>(dx) + dx_coff, dx_ld,
                       accumulate, tiles_x);
    CRD_LAUNCH_CHECK("crd_bicubic2x_bwd");
    return CRD_OK;
  }
  const long long total = (long long)H * W * (C / 8);
  hipLaunchKernelGGL(k_bicubic_bwd, dim3((unsigned)cdiv(total, TPB), B), dim3(TPB), 0, as_stream(stream),
                     reinterpret_cast<const bf16_t*>(dy) + dy_coff, dy_ld, H, W, C, reinterpret_cast<bf16_t*>(dx) + dx_coff, dx_ld,
                     accumulate);
  CRD_LAUNCH_CHECK("crd_bicubic2x_bwd");
  return CRD_OK;
}

extern "C" int crd_nchw_to_pm(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, void* y, int32_t y_ld, int32_t y_coff,
                              int32_t Cpad, crd_stream_t stream) {
  CRD_CHECK_ARG(x && y, "crd_nchw_to_pm: null pointer");
  CRD_CHECK_ARG(Cpad % 8 == 0 && Cpad >= C && y_ld % 8 == 0 && y_coff % 8 == 0, "crd_nchw_to_pm: alignment");
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(k_nchw_to_pm, dim3(blocks_for(HW * (Cpad / 8)), B), dim3(TPB), 0, as_stream(stream), x, C, HW,
                     reinterpret_cast<bf16_t*>(y) + y_coff, y_ld, Cpad);
  CRD_LAUNCH_CHECK("crd_nchw_to_pm");
  return CRD_OK;
}

extern "C" int crd_pm_to_nchw(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t C, int32_t H,
                              int32_t W, float* y, crd_stream_t stream) {
  CRD_CHECK_ARG(x && y, "crd_pm_to_nchw: null pointer");
  const long long HW = (long long)H * W;
  const void* xp = x_f32 ? (const void*)(reinterpret_cast<const float*>(x) + x_coff) : (const void*)(reinterpret_cast<const bf16_t*>(x) + x_coff);
  hipLaunchKernelGGL(k_pm_to_nchw, dim3(blocks_for(HW * C), B), dim3(TPB), 0, as_stream(stream), xp, x_f32, x_ld, C, HW, y);
  CRD_LAUNCH_CHECK("crd_pm_to_nchw");
  return CRD_OK;
}

extern "C" int crd_seg_argmax(const void* logits, int32_t ld, int32_t B, int32_t P, int32_t C, int32_t num_classes, void* y,
                              int32_t y_f32, int32_t y_ld, int32_t y_coff, crd_stream_t stream) {
  CRD_CHECK_ARG(logits && y && C >= 1 && num_classes >= 1, "crd_seg_argmax: bad argument");
  const long long rows = (long long)B * P;
  void* yp = y_f32 ? (void*)(reinterpret_cast<float*>(y) + y_coff) : (void*)(reinterpret_cast<bf16_t*>(y) + y_coff);
  hipLaunchKernelGGL(k_seg_argmax, dim3(blocks_for(rows)), dim3(TPB), 0, as_stream(stream), reinterpret_cast<const float*>(logits),
                     ld, rows, C, num_classes, yp, y_f32, y_ld);
  CRD_LAUNCH_CHECK("crd_seg_argmax");
  return CRD_OK;
}

extern "C" int crd_scale_f32(const float* src, float* dst, int64_t n, float scale, crd_stream_t stream) {
  CRD_CHECK_ARG(src && dst && n > 0, "crd_scale_f32: bad argument");
  hipLaunchKernelGGL(k_scale_f32, dim3(blocks_for(n)), dim3(TPB), 0, as_stream(stream), src, dst, (long long)n, scale);
  CRD_LAUNCH_CHECK("crd_scale_f32");
  return CRD_OK;
}

extern "C" int crd_slice_copy(const void* src, int32_t s_ld, int32_t s_coff, void* dst, int32_t d_ld, int32_t d_coff,
                              int64_t rows, int32_t C, int32_t accumulate, crd_stream_t stream) {
  CRD_CHECK_ARG(src && dst, "crd_slice_copy: null pointer");
  CRD_CHECK_ARG(C % 8 == 0 && s_ld % 8 == 0 && s_coff % 8 == 0 && d_ld % 8 == 0 && d_coff % 8 == 0, "crd_slice_copy: alignment");
  hipLaunchKernelGGL(k_slice_copy, dim3(blocks_for(rows * (C / 8))), dim3(TPB), 0, as_stream(stream),
                     reinterpret_cast<const bf16_t*>(src) + s_coff, s_ld, reinterpret_cast<bf16_t*>(dst) + d_coff, d_ld,
                     (long long)rows, C, accumulate);
  CRD_LAUNCH_CHECK("crd_slice_copy");
  return CRD_OK;
}

extern "C" int crd_f32_to_bf16_rows(const float* src, int32_t s_ld, void* dst, int32_t d_ld, int32_t d_coff, int64_t rows,
                                    int32_t C, const float* scale, int64_t rows_per_sample, const void* add, int32_t add_ld,
                                    int32_t add_coff, crd_stream_t stream) {
  CRD_CHECK_ARG(src && dst && rows_per_sample > 0, "crd_f32_to_bf16_rows: bad argument");
  const int vec = (C % 8 == 0) && (s_ld % 8 == 0) && (d_ld % 8 == 0) && (d_coff % 8 == 0) &&
                  (!add || (add_ld % 8 == 0 && add_coff % 8 == 0)) && ((reinterpret_cast<uintptr_t>(src) & 31) == 0);
  hipLaunchKernelGGL(k_f32_to_bf16_rows, dim3(blocks_for(vec ? rows * (C / 8) : rows * C)), dim3(TPB), 0, as_stream(stream), src,
                     s_ld, reinterpret_cast<bf16_t*>(dst) + d_coff, d_ld, (long long)rows, C, scale, (long long)rows_per_sample,
                     add ? reinterpret_cast<const bf16_t*>(add) + add_coff : nullptr, add_ld, vec);
  CRD_LAUNCH_CHECK("crd_f32_to_bf16_rows");
  return CRD_OK;
}

extern "C" int crd_head_conv2_fwd(const void* a, const float* w, const float* bias, int32_t B, int32_t H, int32_t W, float* depth,
                                  void* copy, int32_t copy_ld, int32_t copy_coff, crd_stream_t stream) {
  CRD_CHECK_ARG(a && w && bias && depth, "crd_head_conv2_fwd: null pointer");
  hipLaunchKernelGGL(k_head2_fwd, dim3((unsigned)head2_blocks(H, W, B), B), dim3(TPB), 0, as_stream(stream),
                     reinterpret_cast<const bf16_t*>(a), w, bias, H, W, depth,
                     copy ? reinterpret_cast<bf16_t*>(copy) + copy_coff : nullptr, copy_ld);
  CRD_LAUNCH_CHECK("crd_head_conv2_fwd");
  return CRD_OK;
}

static int head2_bwd_launch(const float* gd, const void* add, int32_t add_ld, int32_t add_coff, const void* a, const float* w,
                            int32_t B, int32_t H, int32_t W, void* dz, crd_sum_t* dw_rows, int32_t replicas, int parts,
                            const char* who, crd_stream_t stream) {
  CRD_CHECK_ARG(gd && a && (!(parts & 1) || (w && dz)) && (!(parts & 2) || (dw_rows && replicas >= 1)), "%s: null pointer / replicas < 1", who);
  const bf16_t* addp = add ? reinterpret_cast<const bf16_t*>(add) + add_coff : nullptr;
  hipStream_t st = as_stream(stream);
  if (parts & 1)
    hipLaunchKernelGGL(k_head2_bwd_data, dim3((unsigned)head2_blocks(H, W, B), B), dim3(TPB), 0, st, gd, addp, add_ld,
                       reinterpret_cast<const bf16_t*>(a), w, H, W, reinterpret_cast<bf16_t*>(dz));
  if (parts & 2) {
    const int P = H * W;
    int nblk = cdiv(P, 64 * 4);
    int cap = 2048 / (B > 0 ? B : 1); if (cap < 1) cap = 1;
    if (nblk > cap) nblk = cap;
    const int chunk = cdiv(P, nblk);
    nblk = cdiv(P, chunk);
    hipLaunchKernelGGL(k_head2_wgrad, dim3(nblk, B), dim3(TPB), 0, st, gd, addp, add_ld, reinterpret_cast<const bf16_t*>(a), H, W, chunk,
                       dw_rows, replicas);
  }
  CRD_LAUNCH_CHECK(who);
  return CRD_OK;
}

extern "C" int crd_head_conv2_bwd(const float* gd, const void* add, int32_t add_ld, int32_t add_coff, const void* a, const float* w,
                                  int32_t B, int32_t H, int32_t W, void* dz, crd_sum_t* dw_rows, int32_t replicas,
                                  crd_stream_t stream) {
  return head2_bwd_launch(gd, add, add_ld, add_coff, a, w, B, H, W, dz, dw_rows, replicas, 3, "crd_head_conv2_bwd", stream);
}

extern "C" int crd_head_conv2_bwd_data(const float* gd, const void* add, int32_t add_ld, int32_t add_coff, const void* a,
                                       const float* w, int32_t B, int32_t H, int32_t W, void* dz, crd_stream_t stream) {
  return head2_bwd_launch(gd, add, add_ld, add_coff, a, w, B, H, W, dz, nullptr, 0, 1, "crd_head_conv2_bwd_data", stream);
}

extern "C" int crd_head_conv2_wgrad(const float* gd, const void* add, int32_t add_ld, int32_t add_coff, const void* a, int32_t B,
                                    int32_t H, int32_t W, crd_sum_t* dw_rows, int32_t replicas, crd_stream_t stream) {
  return head2_bwd_launch(gd, add, add_ld, add_coff, a, nullptr, B, H, W, nullptr, dw_rows, replicas, 2, "crd_head_conv2_wgrad", stream);
}

extern "C" int crd_sigmoid_bwd(const void* a, void* da, int64_t n, crd_stream_t stream) {
  CRD_CHECK_ARG(a && da && n % 8 == 0, "crd_sigmoid_bwd: bad argument");
  hipLaunchKernelGGL(k_sigmoid_bwd, dim3(blocks_for(n / 8)), dim3(TPB), 0, as_stream(stream), reinterpret_cast<const bf16_t*>(a),
                     reinterpret_cast<bf16_t*>(da), (long long)(n / 8));
  CRD_LAUNCH_CHECK("crd_sigmoid_bwd");
  return CRD_OK;
}
