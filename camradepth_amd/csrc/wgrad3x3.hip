// Streaming weight gradient of the 3x3 / stride-1 / pad-1 convolutions for gfx950 (the decoder's ShortResBlock convs).
//
//   dw[co][tap][ci] += sum_{b,y,x} dy[b,y,x,co] * x[b, y+ky-1, x+kx-1, ci]
//
// The generic split-K kernel (wgrad.hip) re-gathers the input once per tap and re-reads dy once per 128-wide (tap,ci)
// tile: ~9 GB of L2 traffic per launch on the big layers, and it is bound by that.  Here a workgroup owns
//   * one 64-channel chunk of ci, ALL nine taps and all of Cout  -> 9 x (Cout x 64) fp32 accumulators in registers
//     (wave w: 16 output channels x 64 ci x 9 taps = 36 v_mfma_f32_16x16x32_bf16 tiles = 144 VGPRs),
//   * a contiguous range of "strip rows": 32-pixel-wide column strips of the images, walked top to bottom.
// Each step consumes one 32-pixel output row: its dy row and the three input rows around it come from LDS rings that
// are filled by LDS-DMA three steps ahead (HBM latency), so every input row is fetched once per strip instead of
// nine times, and dy once per ci chunk.  Both operands are pixel-major, so MFMA fragments are read with the
// transposing LDS read ds_read_b64_tr_b16.  Partial results leave the workgroup once, as fp32 atomics.
#include <stdlib.h>
#include "common.h"

namespace {

struct Wg3K {
  const bf16_t* x; int x_ld; int Cin;           // input activations (Cin padded to 8)
  const bf16_t* dy; int dy_ld; int Cout;
  int B, H, W;
  int strips_x;                                  // ceil(W / 32)
  int rows_per_wg;                               // strip rows per workgroup
  int S, nchunks;                                // row splits per channel chunk, channel chunks of 64
  long long total_rows;                          // B * strips_x * H
  long long x_bytes, dy_bytes;
  crd_sum_t* dw; crd_sum_t* dbias;
  float* dw_part;                                // per-row-split copies [gridDim.x][Cout][9][Cin] (plain stores) or nullptr -> atomics into dw
};

constexpr int XS = 6, YS = 4;                    // ring slots: input rows / dy rows
constexpr int XPX = 40;                          // pixels per staged input row (34 used, 5 DMA groups of 8)
constexpr int XLD = 64;                          // channels per chunk (128-byte LDS rows)
typedef __attribute__((address_space(3))) void* lds_ptr;
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Source-side swizzle of the lane-linear LDS images so that the transposing reads are bank-conflict free: a 16-lane
// group of ds_read_b64_tr_b16 fetches 4 pixel rows x 32 bytes, a half-wave two such groups 8 rows apart.  The 32-byte
// unit u of row r is stored at unit u ^ f(r)  (found by exhaustive search over the access pattern; 64 banks x 4 B).
__device__ __forceinline__ int swz128(int r) { return (r & 3) ^ ((r >> 3) & 1); }            // 128-byte rows (4 units)
__device__ __forceinline__ int swz256(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }     // 256-byte rows (8 units)
// 192-byte rows (6 units): four consecutive rows already start 48 banks apart; only the second 16-lane group of a
// half-wave (8 rows further, the same banks again) has to move, by one unit
__device__ __forceinline__ int swz192(int r) { return (r >> 3) & 1; }
template <int COT>
__device__ __forceinline__ int swz_y(int r) { return COT == 128 ? swz256(r) : (COT == 96 ? swz192(r) : (COT == 64 ? swz128(r) : 0)); }

__device__ __forceinline__ s16x4 tr_read3(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// WCO x WCI = 8 waves; a wave owns TCO 16-channel co tiles and NT 16-channel ci tiles of the 64-channel chunk
// (fragment reads per strip row and wave: 2 TCO of dy, 18 NT of the input, for 9 TCO NT MFMAs: many co tiles x ONE ci tile is best)
// RPS = image rows per step (per barrier): with one row a wave issues 9 x TCO x NT = 18-36 MFMAs between two barriers; two rows
// halve the barriers and the counted waits per MFMA (the rings are twice as deep: 12 input rows, 8 dy rows).
template <int WCO, int WCI, int TCO, int RPS>
__global__ __launch_bounds__(512) void k_wgrad3x3(Wg3K a) {
  constexpr int XSr = XS * RPS, YSr = YS * RPS;
  static_assert(WCO * WCI == 8, "8 waves");
  constexpr int NT = 4 / WCI;
  constexpr int COT = WCO * TCO * 16;            // channels of dy staged per row (>= Cout)
  constexpr int YGRP = 32 * COT * 2 / 1024;      // DMA wave-instructions per dy row (1 KiB each)
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sX = lds;                              // [XSr][XPX][XLD]
  bf16_t* sY = lds + XSr * XPX * XLD;            // [YSr][32][COT]

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wco = wv / WCI, wci = wv % WCI;
  // workgroup = chunk * S + split.  (Placing the chunks of a row split on one XCD, so that their re-reads of the same dy rows hit
  // that XCD's L2, measured the same -- 304 -> 128 at 256 x 416: 0.706 vs 0.713 ms: the kernel is bound by its LDS fragment reads and
  // MFMA issue inside the CU, about 1 us per strip row, not by the 2.5 TB/s it fetches; tools/sweep_w3.sh)
  const int chunk = (int)blockIdx.x / a.S, split = (int)blockIdx.x - chunk * a.S;
  const int c0 = chunk * XLD;
  const unsigned OOB = 0x80000000u;

  f32x4 acc[9][TCO][NT];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[tp][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loop-invariant (swizzled) fragment offsets
  const int q = l >> 4, prow = (l & 15) >> 2, pcol = (l & 3) * 4;
  int yoff[TCO][2], xoff[3][NT][2];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 8 * q + prow + 4 * h;
      yoff[i][h] = r * COT + (((wco * TCO + i) ^ swz_y<COT>(r)) << 4) + pcol;
    }
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 8 * q + prow + kx + 4 * h;
        xoff[kx][j][h] = r * XLD + (((wci * NT + j) ^ swz128(r)) << 4) + pcol;
      }
  const bool do_bias = a.dbias != nullptr && chunk == 0 && t < a.Cout;
  float bsum = 0.f;

  long long sr = (long long)split * a.rows_per_wg;
  long long sr_end = sr + a.rows_per_wg;
  if (sr_end > a.total_rows) sr_end = a.total_rows;

  // DMA lane roles: input row = 5 wave-instructions (8 pixels x 128 B), issued by waves 0..4;
  // dy row = YGRP wave-instructions, issued by waves 0..YGRP-1.  lane -> (pixel in group, 16-byte granule)
  const int xpix = 8 * wv + (l >> 3);            // staged pixel index 0..39 (image x = x0 - 1 + xpix)
  const int xgr = l & 7;                         // channel granule inside the chunk
  // dy row: 32 pixels x COT/8 granules, lane-linear over the YGRP wave-instructions (a 96-channel row is 12 granules:
  // an instruction then covers 5 1/3 pixels)
  const int ylin = 64 * wv + l;
  const int ypix = ylin / (COT / 8);
  const int ygr = ylin % (COT / 8);

  while (sr < sr_end) {
    // ---- one segment: rows [y0, y1) of strip (b, sx) ----
    const long long strip = sr / a.H;
    const int y0 = (int)(sr - strip * a.H);
    long long seg_end = (strip + 1) * a.H;
    if (seg_end > sr_end) seg_end = sr_end;
    const int y1 = y0 + (int)(seg_end - sr);
    const int b = __builtin_amdgcn_readfirstlane((int)(strip / a.strips_x));
    const int sx = (int)(strip - (long long)b * a.strips_x);
    const int x0 = sx * 32;
    // one buffer descriptor per image: byte offsets stay below 2^31 whatever the batch (4 x 928 x 1600 x 304 channels is 3.6 GB)
    // (requests through lds_dma16, common.h: with the builtin the compiler drains the rings in front of every step's reads)
    const crd_rsrc_t rx = make_rsrc(a.x + (long long)b * a.H * a.W * a.x_ld, (unsigned)a.x_bytes);
    const crd_rsrc_t ry = make_rsrc(a.dy + (long long)b * a.H * a.W * a.dy_ld, (unsigned)a.dy_bytes);

    auto issue_x = [&](int h) {                  // input row h (image coords) -> ring slot (h + 1) % XS
#if defined(__HIP_DEVICE_COMPILE__)
      if (wv < 5) {
        const int ix = x0 - 1 + xpix, ch = c0 + ((((xgr >> 1) ^ swz128(xpix)) << 1) | (xgr & 1)) * 8;
        const bool ok = (unsigned)h < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && xpix < 34 && ch < a.Cin;
        const unsigned off = ok ? (unsigned)((((long long)h * a.W + ix) * a.x_ld + ch) * 2) : OOB;
        lds_dma16(rx, (unsigned)(uintptr_t)(lds_ptr)(sX + (((h + 1) % XSr) * XPX + 8 * wv) * XLD), off);
      }
#else
      (void)h;
#endif
    };
    auto issue_y = [&](int r) {                  // dy row r -> ring slot r % YS (rows >= y1 are zero-filled)
#if defined(__HIP_DEVICE_COMPILE__)
      if (wv < YGRP) {
        const int ix = x0 + ypix, co = ((((ygr >> 1) ^ swz_y<COT>(ypix)) << 1) | (ygr & 1)) * 8;
        const bool ok = r < y1 && ix < a.W && co < a.Cout;
        const unsigned off = ok ? (unsigned)((((long long)r * a.W + ix) * a.dy_ld + co) * 2) : OOB;
        lds_dma16(ry, (unsigned)(uintptr_t)(lds_ptr)(sY + (r % YSr) * 32 * COT + 512 * wv), off);
      }
#else
      (void)r;
#endif
    };

    // all waves must be done with the previous segment's LDS data before it is overwritten
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // prologue: input rows y0-1, y0 first, then the three "virtual" steps before y0 (each: its dy rows, the next input rows)
    issue_x(y0 - 1);
    issue_x(y0);
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < RPS; ++r) { issue_y(y0 + s * RPS + r); issue_x(y0 + s * RPS + 1 + r); }

    for (int y = y0; y < y1; y += RPS) {
      // this step's dy rows and last input rows were issued three steps ago: at most two steps' worth of DMA may still be in flight
      if (wv < 5 && wv < YGRP) wait_vm<4 * RPS>();
      else if (wv < 5 || wv < YGRP) wait_vm<2 * RPS>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int r = 0; r < RPS; ++r) { issue_y(y + 3 * RPS + r); issue_x(y + 3 * RPS + 1 + r); }

      // a wave whose 16 x NT input channels all lie beyond Cin (the tail chunk of Cin = 136 / 144 / 240: 3, 3, 1 of the 4
      // ci-waves) only takes part in the DMA and the barriers: its MFMAs would multiply zeros -- at the chip's power limit
      // they still cost time
      // (the dy column sums of the bias gradient below are still theirs to add: threads t < Cout of chunk 0)
      const bool mfma_on = c0 + wci * NT * 16 < a.Cin;
      if (!mfma_on && !do_bias) continue;
#pragma unroll
      for (int rr = 0; rr < RPS; ++rr) {
        const int yy = y + rr;
        if (rr > 0 && yy >= y1) break;                 // odd segment length: the last step has one row
        const bf16_t* yrow = sY + (yy % YSr) * 32 * COT;
        if (mfma_on) {
        bf16x8 af[TCO];
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
          s16x4 lo = tr_read3(yrow + yoff[i][0]), hi = tr_read3(yrow + yoff[i][1]);
          union { bf16x8 v; s16x4 h[2]; } u;
          u.h[0] = lo; u.h[1] = hi;
          af[i] = u.v;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const bf16_t* xrow = sX + ((yy + ky) % XSr) * XPX * XLD;      // input row yy+ky-1 lives in slot (yy+ky) % XSr
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              s16x4 lo = tr_read3(xrow + xoff[kx][j][0]), hi = tr_read3(xrow + xoff[kx][j][1]);
              union { bf16x8 v; s16x4 h[2]; } u;
              u.h[0] = lo; u.h[1] = hi;
#pragma unroll
              for (int i = 0; i < TCO; ++i)
                acc[ky * 3 + kx][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], u.v, acc[ky * 3 + kx][i][j], 0, 0, 0);
            }
          }
        }
        }   // mfma_on
        if (do_bias) {
#pragma unroll 8
          for (int r = 0; r < 32; ++r) bsum += bf2f(yrow[r * COT + ((((t >> 4) ^ swz_y<COT>(r)) << 4) | (t & 15))]);
        }
      }
    }
    sr = seg_end;
  }

  // ---- flush: dw[co][tap][ci] ; D layout: col = lane&15 -> ci, row = (lane>>4)*4 + r -> co ----
  const int Ktot = 9 * a.Cin;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int ci = c0 + (wci * NT + j) * 16 + (l & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = (wco * TCO + i) * 16 + (l >> 4) * 4 + r;
          if (co < a.Cout && ci < a.Cin) {
            const long long off = (long long)co * Ktot + tp * a.Cin + ci;
            // every (row split, chunk) workgroup owns its block of its split's copy: plain stores, the caller sums the
            // copies (crd_wgrad_unpack); otherwise Cout x 9 x Cin atomics per row split (19 M per launch, ~0.1 ms)
            if (a.dw_part) a.dw_part[(long long)split * a.Cout * Ktot + off] = acc[tp][i][j][r];
            else grad_add(a.dw + off, acc[tp][i][j][r]);
          }
        }
      }
  if (do_bias) grad_add(a.dbias + t, bsum);
}

// Row splits (= partial copies) and strip rows per workgroup.  budget: workgroups the launch may use (0 = one per CU: every row
// split adds Cout x 9 x Cin fp32 values to sum, which a second workgroup per CU does not win back: 30.0 vs 30.3 ms/step); cap > 0:
// the number of partial copies the caller provides.
int plan_rows(long long total_rows, int Cin, int& rows_per_wg, int budget, int cap) {
  const int chunks = cdiv(Cin, XLD);
  const int W = budget > 0 ? budget : 256;
  long long S = W / chunks;
  if (S < 1) S = 1;
  if (cap > 0 && S > cap) S = cap;
  if (S > total_rows / 8) S = total_rows / 8 > 0 ? total_rows / 8 : 1;
  rows_per_wg = (int)((total_rows + S - 1) / S);
  return (int)((total_rows + rows_per_wg - 1) / rows_per_wg);
}

template <int WCO, int WCI, int TCO, int RPS>
int launch_w3r(const Wg3K& k0, hipStream_t st, int budget, int partial_capacity) {
  Wg3K k = k0;
  constexpr int COT = WCO * TCO * 16;
  const size_t lds = (size_t)RPS * (XS * XPX * XLD + YS * 32 * COT) * sizeof(bf16_t);
  k.nchunks = cdiv(k.Cin, XLD);
  k.S = plan_rows(k.total_rows, k.Cin, k.rows_per_wg, budget, k.dw_part ? partial_capacity : 0);
  static bool attr_done = false;
  if (!attr_done) {
    crd_reserve_lds(reinterpret_cast<const void*>(&k_wgrad3x3<WCO, WCI, TCO, RPS>), (int)lds, "k_wgrad3x3");
    attr_done = true;
  }
  hipLaunchKernelGGL((k_wgrad3x3<WCO, WCI, TCO, RPS>), dim3(k.nchunks * k.S), dim3(512), lds, st, k);
  CRD_LAUNCH_CHECK("crd_conv_wgrad(3x3 streaming)");
  return CRD_OK;
}

template <int WCO, int WCI, int TCO>
int launch_w3(const Wg3K& k, hipStream_t st, int budget, int partial_capacity) {
  static int rps = -1;
  if (rps < 0) rps = crd_dev_int("CRD_W3_RPS", 2);
  return rps == 1 ? launch_w3r<WCO, WCI, TCO, 1>(k, st, budget, partial_capacity) : launch_w3r<WCO, WCI, TCO, 2>(k, st, budget, partial_capacity);
}

}  // namespace

// Called from crd_conv_wgrad for 3x3 / stride 1 / pad 1 layers on grids at least one strip wide.
int crd_wgrad3x3_stream(const crd_wgrad_desc* d, hipStream_t st) {
  Wg3K k;
  k.x = reinterpret_cast<const bf16_t*>(d->x) + d->x_coff; k.x_ld = d->x_ld; k.Cin = d->Cin;
  k.dy = reinterpret_cast<const bf16_t*>(d->dy) + d->dy_coff; k.dy_ld = d->dy_ld; k.Cout = d->Cout;
  k.B = d->B; k.H = d->IH; k.W = d->IW;
  k.strips_x = cdiv(d->IW, 32);
  k.total_rows = (long long)d->B * k.strips_x * d->IH;
  k.rows_per_wg = 0;
  k.x_bytes = (long long)d->IH * d->IW * d->x_ld * 2;          // per image (the kernel builds one descriptor per image)
  k.dy_bytes = (long long)d->IH * d->IW * d->dy_ld * 2;
  CRD_UNSUPPORTED(k.x_bytes < (1ll << 31) && k.dy_bytes < (1ll << 31), "crd_conv_wgrad: image too large for 32-bit byte offsets");
  k.dw = d->dw; k.dbias = d->dbias;
  k.dw_part = d->dw_partials;
  CRD_CHECK_ARG(d->dw_partials == nullptr || d->dw_partial_capacity >= 1, "crd_conv_wgrad: dw_partials needs a capacity >= 1");
  CRD_CHECK_ARG(d->wg_budget >= 0, "crd_conv_wgrad: wg_budget must be >= 0");
  if (d->Cout <= 32) return launch_w3<2, 4, 1>(k, st, d->wg_budget, d->dw_partial_capacity);
  if (d->Cout <= 64) return launch_w3<2, 4, 2>(k, st, d->wg_budget, d->dw_partial_capacity);
  if (d->Cout <= 96) return launch_w3<2, 4, 3>(k, st, d->wg_budget, d->dw_partial_capacity);     // 96-channel dy rows: no padded MFMA tiles
  // 128 output channels: FOUR co tiles x one ci tile per wave.  Per strip row a wave then reads 8 dy + 18 input fragments for its 36
  // MFMAs; with 2 x 2 tiles (round 2-3) it was 4 + 36 -- the kernel is bound by those LDS reads (0.716 -> 0.658 ms at 304 -> 128,
  // 0.293 -> 0.262 ms at 128 -> 128, 256 x 416 x 8)
  if (crd_dev_int("CRD_W3_T22", 0)) return launch_w3<4, 2, 2>(k, st, d->wg_budget, d->dw_partial_capacity);
  return launch_w3<2, 4, 4>(k, st, d->wg_budget, d->dw_partial_capacity);
}

int crd_wgrad3x3_splits(const crd_wgrad_desc* d) {
  int rows = 0;
  return plan_rows((long long)d->B * cdiv(d->IW, 32) * d->IH, d->Cin, rows, d->wg_budget, d->dw_partial_capacity > 0 ? d->dw_partial_capacity : 0);
}
