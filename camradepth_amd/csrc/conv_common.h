// Shared pieces of the MFMA convolution kernels (igemm.hip, conv3x3.hip): kernel argument block and the fused epilogue.
#pragma once
#include "common.h"

namespace crdk {

struct ConvK {
  const bf16_t* x; int x_ld; int IH, IW, Cin; long long x_bstride;
  const bf16_t* w; int Cout, KW, stride, pad, Ktot;
  int OW, OHW; int gather_mode;
  void* y; int y_ld; int y_f32; long long y_bstride;
  int out_mode, patch_k, patch_c, YW;
  const float* bias; int bias_bstride; int act;
  const float* res; int res_ld; long long res_bstride; const float* res_scale;
  int accumulate; crd_sum_t* stats; int G16;
  float* stats_partial; int n_tiles;   // per-tile partial sums [B][n_tiles][G16][2] (plain stores) or nullptr -> atomics
  int vec_ok;   // y base/offset 16-byte aligned: the vectorised epilogue may be used
  int vecf_ok;  // fp32 output (+ residual) rows are 16-byte aligned as well: LDS-staged float4 epilogue
  int lds_bytes;   // dynamic LDS of the launch (the fp32 staging tile needs BM x (BN+4) floats)
  // Optional: reduce phase of the backward of the GroupNorm (+GELU) whose dy this launch produces (vector path only).
  // red_x = that GroupNorm's raw bf16 input [pixels][red_x_ld], batch stride red_x_bstride; r as in crd_gn_bwd_reduce.
  const void* red_x; int red_x_f32; int red_x_ld; long long red_x_bstride;   // bf16, or fp32 when red_x_f32 (the residual stream)
  const crd_sum_t* red_stats; const float* red_gamma; const float* red_beta; int red_gmul, red_act;
  crd_sum_t* red_r;
  crd_sum_t* chan;   // optional per-channel (sum, sumsq) of the stored output [B][Cout][2] (scalar epilogue path only)
  int col0;  // first output column of this launch (the 3x3 halo kernel covers wide layers with two tile widths)
  int dbg;   // developer experiments only (CRD_DBG env): 1 no weight-DMA wait, 2 no DMA at all, 4 no epilogue
  int ny_tiles;   // column tiles of the launch, for kernels that decode (row tile, column tile) from a 1-D XCD-aware grid (xfgemm.hip)
};


// fp32 LDS-staged epilogue (conv_epilogue's second branch) applies?  One definition for conv_epilogue and conv_epilogue_idle.
template <int BM, int BN, int NT>
__device__ __forceinline__ bool conv_use_vecf(const ConvK& a) {
  return a.vecf_ok && a.out_mode == 0 && a.y_f32 && !a.stats_partial && (NT % (BN / 4) == 0) &&
         (size_t)BM * (BN + 4) * 4 + (size_t)NT * 8 * 4 <= (size_t)a.lds_bytes;
}

// Fused epilogue of one wave's TM x TN grid of 32x32 accumulator tiles (v_mfma_f32_32x32x16 C/D layout: lane l holds
// column l&31 and rows (r&3) + 8*(r>>2) + 4*(l>>5)).  `pix(i, rr, valid, p)` maps accumulator row rr of row-tile i to the
// linear output-pixel index p of the image (or valid=false).  Fuses: bias (optionally per image), sigmoid, residual add
// with per-image scale, fp32/bf16 store or read-modify-write accumulate, patch scatter, GroupNorm partial statistics.
// `smem` is the kernel's LDS (free after the K loop).  Statistics are reduced across the WM waves that share a column
// range before they leave the workgroup.  `pix(i, rr, valid, p)` maps accumulator row rr of row-tile i to the output
// pixel p; `pixrow(row_local, valid, p)` does the same for a row of the whole BM x BN tile (vector store path).
// Vector path (bf16 output, plain layout): the tile is transposed through LDS so that every thread stores -- or
// read-modify-writes, for gradient accumulation -- 16 contiguous bytes instead of 64 scattered 2-byte elements.
template <int TM, int TN, int WM, int WN, int HALVES = 1, typename PixFn, typename PixRowFn>
__device__ __forceinline__ void conv_epilogue(const ConvK& a, f32x16 (&acc)[TM][TN], int b, int l, int wm, int wn, int n0,
                                              int tile, void* smem, PixFn pix, PixRowFn pixrow) {
  // HALVES = 2: the staging tile holds half of the rows at a time (two passes; the waves of the other half wait), for
  // kernels that budget their LDS for two workgroups per CU
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64, LDT = BN + 8, BMH = BM / HALVES;
  static_assert(HALVES == 1 || (HALVES == 2 && WM % 2 == 0), "row halves are whole waves");
  float* red = reinterpret_cast<float*>(smem) + (BMH * LDT) / 2;     // behind the bf16 staging tile
  const bool vec = (a.out_mode == 0 || (a.patch_c & 7) == 0) && !a.y_f32 && !a.res && (a.Cout & 7) == 0 && (a.y_ld & 7) == 0 && a.vec_ok;
  if (vec) {
    bf16_t* T = reinterpret_cast<bf16_t*>(smem);
    bf16_t* yb = reinterpret_cast<bf16_t*>(a.y) + (long long)b * a.y_bstride;
    constexpr int GPR = BN / 8;
    // fused GroupNorm-backward reduce: a thread keeps the same 8 columns for all its rows (NT % GPR == 0)
    const bool redo = (NT % GPR == 0) && a.red_x != nullptr;
    float rs0[8], rs1[8], rga[8], rbe[8], rmean = 0.f, rrstd = 0.f;
    const int rcol = n0 + (threadIdx.x % GPR) * 8;
    if (redo) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { rs0[j] = rs1[j] = 0.f; rga[j] = 1.f; rbe[j] = 0.f; }
      if (rcol < a.Cout) {
        const int cpg = 16 * a.red_gmul;
        gn_mean_rstd(a.red_stats + (long long)b * (a.Cout >> 4) * 2, (rcol / cpg) * a.red_gmul, a.red_gmul,
                     (float)a.OHW * cpg, rmean, rrstd);
#pragma unroll
        for (int j = 0; j < 8; ++j) { rga[j] = a.red_gamma[rcol + j]; rbe[j] = a.red_beta[rcol + j]; }
      }
    }
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
    if (h > 0) __syncthreads();            // the previous half has been stored
    if (HALVES == 1 || (wm * TM * 32) / BMH == h) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cl = (wn * TN + j) * 32 + (l & 31), col = n0 + cl;
      const bool colok = col < a.Cout;
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
          bool valid;
          int row;
          pix(i, rr, valid, row);
          float v = acc[i][j][r];          // bias is already in the accumulator (see the kernels' accumulator init)
          if (a.act == 1) v = sigmoidf_(v);
          const bf16_t q = f2bf(v);
          T[((wm * TM + i) * 32 + rr - h * BMH) * LDT + cl] = q;
          if (valid && colok) { v = bf2f(q); s += v; ss += v * v; }
        }
      }
      if (a.stats) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
        s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
        if ((l & 15) == 0 && l < 32) {
          float* r = red + (((wm * WN + wn) * TN + j) * 2 + (l >> 4)) * 2;
          r[0] = s; r[1] = ss;
        }
      }
    }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < BMH * GPR; idx += NT) {
      const int rh = idx / GPR, g = idx - rh * GPR, rl = rh + h * BMH;
      bool valid;
      int p;
      pixrow(rl, valid, p);
      const int col = n0 + g * 8;
      if (valid && col < a.Cout) {
        uint4 u = *reinterpret_cast<const uint4*>(T + rh * LDT + g * 8);
        bf16_t* dst = yb + (long long)p * a.y_ld + col;
        if (a.out_mode == 1) {               // patch scatter: column = (tap, channel), 8 columns stay inside one tap
          const int tap = col / a.patch_c, pci = col - tap * a.patch_c;
          const int pky = tap / a.patch_k, pkx = tap - pky * a.patch_k;
          const int oy = p / a.OW, ox = p - oy * a.OW;
          dst = yb + ((long long)(oy * a.patch_k + pky) * a.YW + (ox * a.patch_k + pkx)) * a.y_ld + pci;
        }
        if (a.accumulate) {
          const uint4 o = *reinterpret_cast<const uint4*>(dst);
          u.x = pack_bf2(bf_lo(u.x) + bf_lo(o.x), bf_hi(u.x) + bf_hi(o.x));
          u.y = pack_bf2(bf_lo(u.y) + bf_lo(o.y), bf_hi(u.y) + bf_hi(o.y));
          u.z = pack_bf2(bf_lo(u.z) + bf_lo(o.z), bf_hi(u.z) + bf_hi(o.z));
          u.w = pack_bf2(bf_lo(u.w) + bf_lo(o.w), bf_hi(u.w) + bf_hi(o.w));
        }
        *reinterpret_cast<uint4*>(dst) = u;
        if (redo) {
          float xq[8];
          load8(a.red_x, (long long)b * a.red_x_bstride + (long long)p * a.red_x_ld + col, a.red_x_f32, xq);
          const float dq[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = (xq[j] - rmean) * rrstd;
            float gg = dq[j];
            if (a.red_act == 1) gg *= gelu_grad(xh * rga[j] + rbe[j]);
            rs0[j] += gg; rs1[j] += gg * xh;
          }
        }
      }
    }
    }   // row halves
    if (redo) {
      // threads t, t + GPR, ... share a column group: fold them through LDS (the staging tile is free again)
      __syncthreads();
      float* fr = reinterpret_cast<float*>(smem);            // [NT][16], then [GPR*16] gamma-weighted
#pragma unroll
      for (int j = 0; j < 8; ++j) { fr[threadIdx.x * 16 + j * 2] = rs0[j]; fr[threadIdx.x * 16 + j * 2 + 1] = rs1[j]; }
      __syncthreads();
      float* fw = fr + NT * 16;
      if ((int)threadIdx.x < GPR * 16) {
        const int g = threadIdx.x >> 4, jk = threadIdx.x & 15;
        float v = 0.f;
        for (int m = 0; m < NT / GPR; ++m) v += fr[(g + m * GPR) * 16 + jk];
        const int c = n0 + g * 8 + (jk >> 1);
        float wv = 0.f;
        if (c < a.Cout) {
          grad_add(&a.red_r[((long long)b * a.Cout + c) * 2 + (jk & 1)], v);
          wv = v * a.red_gamma[c];
        }
        fw[threadIdx.x] = wv;
      }
      __syncthreads();
      if ((int)threadIdx.x < (BN / 16) * 2) {                   // gamma-weighted sums of the tile's 16-channel slabs
        const int slab = threadIdx.x >> 1, which = threadIdx.x & 1;
        const int c0s = n0 + slab * 16;
        if (c0s < a.Cout) {
          float s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) s2 += fw[(slab * 16 + j) * 2 + which];
          const int cpg = 16 * a.red_gmul;
          grad_add(&a.red_r[(long long)gridDim.z * a.Cout * 2 + ((long long)b * (a.Cout / cpg) + c0s / cpg) * 2 + which], s2);
        }
      }
    }
  } else if (conv_use_vecf<BM, BN, NT>(a)) {
    // fp32 output (+ residual, + GroupNorm sums of the stored values): the tile goes through LDS as floats and leaves
    // with 16-byte loads / stores (the per-lane 4-byte form below cost +5..11 us on the encoder's fc2 convs)
    constexpr int LDF = BN + 4, GPR4 = BN / 4;
    float* T = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cl = (wn * TN + j) * 32 + (l & 31);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
          float v = acc[i][j][r];
          if (a.act == 1) v = sigmoidf_(v);
          T[((wm * TM + i) * 32 + rr) * LDF + cl] = v;
        }
    }
    __syncthreads();
    const float rscale = (a.res && a.res_scale) ? a.res_scale[b] : 1.f;
    float* yb = reinterpret_cast<float*>(a.y) + (long long)b * a.y_bstride;
    const float* resb = a.res ? a.res + (long long)b * a.res_bstride : nullptr;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, css[4] = {0.f, 0.f, 0.f, 0.f};
    const int g4 = threadIdx.x % GPR4, col = n0 + g4 * 4;      // a thread keeps its 4 columns (NT % GPR4 == 0)
    if (col < a.Cout) {
      for (int rl = threadIdx.x / GPR4; rl < BM; rl += NT / GPR4) {
        bool valid;
        int p;
        pixrow(rl, valid, p);
        if (!valid) continue;
        float4 v = *reinterpret_cast<const float4*>(T + rl * LDF + g4 * 4);
        if (resb) {
          const float4 q = *reinterpret_cast<const float4*>(resb + (long long)p * a.res_ld + col);
          v.x = q.x + rscale * bf_round(v.x); v.y = q.y + rscale * bf_round(v.y);
          v.z = q.z + rscale * bf_round(v.z); v.w = q.w + rscale * bf_round(v.w);
        }
        float4* dst = reinterpret_cast<float4*>(yb + (long long)p * a.y_ld + col);
        if (a.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
        cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w;
        css[0] += v.x * v.x; css[1] += v.y * v.y; css[2] += v.z * v.z; css[3] += v.w * v.w;
      }
    }
    if (a.stats) {       // fold the threads that share a column group, then one atomic per column / 16-column slab
      __syncthreads();
      float* fr = reinterpret_cast<float*>(smem);              // [NT][8]; the staging tile is free again
#pragma unroll
      for (int j = 0; j < 4; ++j) { fr[threadIdx.x * 8 + j * 2] = cs[j]; fr[threadIdx.x * 8 + j * 2 + 1] = css[j]; }
      __syncthreads();
      float* fc = fr + NT * 8;                                 // [BN][2] column sums
      if ((int)threadIdx.x < BN * 2) {
        const int c = threadIdx.x >> 1, which = threadIdx.x & 1;
        float v = 0.f;
        for (int m = 0; m < NT / GPR4; ++m) v += fr[((c >> 2) + m * GPR4) * 8 + (c & 3) * 2 + which];
        fc[threadIdx.x] = v;
        if (a.chan && n0 + c < a.Cout) stat_add(&a.chan[((long long)b * a.Cout + n0 + c) * 2 + which], v);
      }
      __syncthreads();
      if ((int)threadIdx.x < (BN / 16) * 2) {
        const int slab = threadIdx.x >> 1, which = threadIdx.x & 1;
        const int gidx = (n0 >> 4) + slab;
        if (gidx < a.G16) {
          float v = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) v += fc[(slab * 16 + j) * 2 + which];
          stat_add(a.stats + ((long long)b * a.G16 + gidx) * 2 + which, v);
        }
      }
    }
    return;
  } else {
  const int col0 = n0 + wn * TN * 32;
  const float rscale = (a.res && a.res_scale) ? a.res_scale[b] : 1.f;
  char* yb = reinterpret_cast<char*>(a.y) + (long long)b * a.y_bstride * (a.y_f32 ? 4 : 2);
  const float* resb = a.res ? a.res + (long long)b * a.res_bstride : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32 + (l & 31);
    const bool colok = col < a.Cout;
    int pky = 0, pkx = 0, pci = col;
    if (a.out_mode == 1) {
      int tap = col / a.patch_c;
      pci = col - tap * a.patch_c;
      pky = tap / a.patch_k;
      pkx = tap - pky * a.patch_k;
    }
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        bool valid;
        int row;
        pix(i, rr, valid, row);
        if (valid && colok) {
          float v = acc[i][j][r];          // bias is already in the accumulator
          if (a.act == 1) v = sigmoidf_(v);
          long long off;
          if (a.out_mode == 0) off = (long long)row * a.y_ld + col;
          else {
            int oy = row / a.OW, ox = row - oy * a.OW;
            off = ((long long)(oy * a.patch_k + pky) * a.YW + (ox * a.patch_k + pkx)) * a.y_ld + pci;
          }
          if (resb) v = resb[(long long)row * a.res_ld + col] + rscale * bf_round(v);
          if (a.y_f32) {
            float* p = reinterpret_cast<float*>(yb) + off;
            if (a.accumulate) v += *p;
            *p = v;
          } else {
            bf16_t* p = reinterpret_cast<bf16_t*>(yb) + off;
            if (a.accumulate) v += bf2f(*p);
            bf16_t q = f2bf(v);
            *p = q;
            v = bf2f(q);
          }
          s += v; ss += v * v;
        }
      }
    }
    if (a.chan) {                               // lanes l and l^32 hold the same column: one atomic pair per column and wave
      const float cs = s + __shfl_xor(s, 32), css = ss + __shfl_xor(ss, 32);
      if (l < 32 && colok) {
        crd_sum_t* cp = a.chan + ((long long)b * a.Cout + col) * 2;
        stat_add(cp, cs);
        stat_add(cp + 1, css);
      }
    }
    if (a.stats) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if ((l & 15) == 0 && l < 32) {            // lanes 0 and 16: the two 16-channel slabs of this 32-column tile
        float* r = red + (((wm * WN + wn) * TN + j) * 2 + (l >> 4)) * 2;
        r[0] = s; r[1] = ss;
      }
    }
  }
  }   // scalar path
  if (a.stats) {
    // one value per (slab, sum|sumsq) of the workgroup's column range: reduce over the WM waves that share it
    __syncthreads();
    const int t = threadIdx.x;
    if (t < WN * TN * 4) {
      const int which = t & 1, slab = (t >> 1) % (TN * 2), wn_ = (t >> 1) / (TN * 2);
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(((w * WN + wn_) * TN) * 2 + slab) * 2 + which];
      const int g = (n0 >> 4) + wn_ * TN * 2 + slab;
      if (g < a.G16) {
        if (a.stats_partial) a.stats_partial[(((long long)b * a.n_tiles + tile) * a.G16 + g) * 2 + which] = v;
        else stat_add(a.stats + ((long long)b * a.G16 + g) * 2 + which, v);
      }
    }
  }
}

// ---- register epilogue of the 64 x 64 tile kernels (round 5) --------------------------------------------------------------------
// The small encoder GEMMs (k_igemm<2,2,1,1>: 210 launches per step) spend more time in the LDS-staged epilogue above than in their K
// loop (160 -> 160 at 16 x 26: launch 1.7 us, K loop 2.2, store path 1.6, statistics 0.9; with the fused reduce three more barriers --
// tools/sweep_igemm.sh).  With the MFMA operands SWAPPED (A = weight rows, B = pixels) a lane holds ONE pixel (l & 31) and, of its
// wave's 32 columns, the channels (r & 3) + 8 (r >> 2) + 4 (l >> 5): v_permlane32_swap between the half-waves turns them into two runs
// of 8 consecutive channels -- two 16-byte stores per lane straight from the accumulators (k_conv3x3p's epilogue).  Everything the
// epilogue READS (the old values of an accumulating store, the GroupNorm input and moments of the fused backward reduce, gamma / beta)
// does not depend on the GEMM: it is requested at kernel start and arrives under the K loop.  The fused reduce's 32 per-lane sums
// (2 pieces x 8 channels x 2 moments) are folded over the 32 pixels of a half-wave with a halving butterfly (31 shuffles: after
// the step with mask m a lane keeps the half of its values selected by its bit m), then over the WM waves through 1.5 KB of LDS.
// Arithmetic identical to conv_epilogue's vector path: bf16(v) stored, accumulate = bf16(bf16(v) + old), GroupNorm sums of the rounded
// values before accumulation, the reduce on what was stored.
typedef __attribute__((ext_vector_type(4))) unsigned crd_u32x4;
struct RegEpiState {
  int row, col[2];
  bool ok[2];
  crd_u32x4 old[2];
  float xq[2][8], rmean[2], rrstd[2], rga[2][8], rbe[2][8];
};

template <int WM, int WN>
__device__ __forceinline__ void reg_epi_prefetch(const ConvK& a, int b, int l, int wm, int wn, int m0, int n0, RegEpiState& st) {
  st.row = m0 + wm * 32 + (l & 31);
  const bool rowok = st.row < a.OHW;
  const int rowc = rowok ? st.row : a.OHW - 1;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    st.col[k] = n0 + wn * 32 + 16 * k + 8 * (l >> 5);
    st.ok[k] = rowok && st.col[k] < a.Cout;
    const int colc = st.col[k] < a.Cout ? st.col[k] : 0;                 // (unconditional loads from a clamped address)
    st.old[k] = crd_u32x4{0u, 0u, 0u, 0u};
    if (a.accumulate)
      st.old[k] = *reinterpret_cast<const crd_u32x4*>(reinterpret_cast<const bf16_t*>(a.y) + (long long)b * a.y_bstride + (long long)rowc * a.y_ld + colc);
    if (a.red_x) {
      load8(a.red_x, (long long)b * a.red_x_bstride + (long long)rowc * a.red_x_ld + colc, a.red_x_f32, st.xq[k]);
      const int cpg = 16 * a.red_gmul;
      gn_mean_rstd(a.red_stats + (long long)b * (a.Cout >> 4) * 2, (colc / cpg) * a.red_gmul, a.red_gmul, (float)a.OHW * cpg, st.rmean[k], st.rrstd[k]);
      if (a.red_act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { st.rga[k][j] = a.red_gamma[colc + j]; st.rbe[k][j] = a.red_beta[colc + j]; }
      }
    }
  }
}

// acc: the wave's 32 x 32 tile in the SWAPPED layout (lane = pixel, registers = channels).  smem: the kernel's LDS, free after the K loop.
template <int WM, int WN>
__device__ __forceinline__ void conv_epilogue_reg(const ConvK& a, f32x16& acc, int b, int l, int wm, int wn, int n0, int tile, void* smem,
                                                  const RegEpiState& st) {
  constexpr int BN = WN * 32, NT = WM * WN * 64;
  float* fr = reinterpret_cast<float*>(smem);              // [WM][BN * 2]  per-channel (sum g, sum g xhat) of each wave row
  float* fw = fr + WM * BN * 2;                            // [BN * 2]      gamma-weighted, for the group sums
  float* sred = fw + BN * 2;                               // [WM][WN][2 slabs][2]
  uint32_t d[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v0 = acc[4 * g], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
    if (a.act == 1) { v0 = sigmoidf_(v0); v1 = sigmoidf_(v1); v2 = sigmoidf_(v2); v3 = sigmoidf_(v3); }
    d[g][0] = pack_bf2(v0, v1);
    d[g][1] = pack_bf2(v2, v3);
  }
  bf16_t* yb = reinterpret_cast<bf16_t*>(a.y) + (long long)b * a.y_bstride;
  float rv[32];
  float s[2], ss[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    // runs 2k (channels 16k + 4 half ..) and 2k + 1 (16k + 8 + 4 half ..): after the swaps the lower half-wave holds channels
    // 16k .. 16k + 7 of its pixel, the upper one 16k + 8 .. 16k + 15
    auto r0 = __builtin_amdgcn_permlane32_swap(d[2 * k][0], d[2 * k + 1][0], false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(d[2 * k][1], d[2 * k + 1][1], false, false);
    crd_u32x4 u = {r0[0], r1[0], r0[1], r1[1]};
    {
      const float q[8] = {bf_lo(u[0]), bf_hi(u[0]), bf_lo(u[1]), bf_hi(u[1]), bf_lo(u[2]), bf_hi(u[2]), bf_lo(u[3]), bf_hi(u[3])};
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { a0 += q[j]; a1 += q[j] * q[j]; }
      s[k] = st.ok[k] ? a0 : 0.f;
      ss[k] = st.ok[k] ? a1 : 0.f;
    }
    if (a.accumulate) {
      const crd_u32x4 o = st.old[k];
#pragma unroll
      for (int w = 0; w < 4; ++w) u[w] = pack_bf2(bf_lo(u[w]) + bf_lo(o[w]), bf_hi(u[w]) + bf_hi(o[w]));
    }
    if (st.ok[k]) *reinterpret_cast<crd_u32x4*>(yb + (long long)st.row * a.y_ld + st.col[k]) = u;
    if (a.red_x) {
      const float dq[8] = {bf_lo(u[0]), bf_hi(u[0]), bf_lo(u[1]), bf_hi(u[1]), bf_lo(u[2]), bf_hi(u[2]), bf_lo(u[3]), bf_hi(u[3])};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (st.xq[k][j] - st.rmean[k]) * st.rrstd[k];
        float gg = st.ok[k] ? dq[j] : 0.f;
        if (a.red_act == 1) gg *= gelu_grad(xh * st.rga[k][j] + st.rbe[k][j]);
        rv[(k * 8 + j) * 2] = gg;
        rv[(k * 8 + j) * 2 + 1] = gg * xh;
      }
    }
  }
  if (!a.stats && !a.red_x) return;
  if (a.red_x) {
    // halving butterfly over the 32 pixels of the half-wave: lane x (0..31) ends up with the total of value x
#pragma unroll
    for (int m = 16, n = 32; m >= 1; m >>= 1, n >>= 1) {
      const bool up = (l & m) != 0;
#pragma unroll
      for (int i = 0; i < n / 2; ++i) {
        const float send = up ? rv[i] : rv[i + n / 2];
        const float keep = up ? rv[i + n / 2] : rv[i];
        rv[i] = keep + __shfl_xor(send, m);
      }
    }
    const int x = l & 31;                                   // value index: piece x >> 4, channel (x >> 1) & 7, moment x & 1
    const int cl = wn * 32 + 16 * (x >> 4) + 8 * (l >> 5) + ((x >> 1) & 7);
    fr[wm * BN * 2 + cl * 2 + (x & 1)] = rv[0];
  }
  if (a.stats) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float sv = wave_sum(s[k]), sq = wave_sum(ss[k]);
      if (l == 0) { sred[(((wm * WN + wn) * 2 + k) * 2)] = sv; sred[(((wm * WN + wn) * 2 + k) * 2) + 1] = sq; }
    }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (a.stats && t < WN * 2 * 2) {
    const int which = t & 1, slab = (t >> 1) & 1, wn_ = t >> 2;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) v += sred[(((w * WN + wn_) * 2 + slab) * 2) + which];
    const int g = (n0 >> 4) + wn_ * 2 + slab;
    if (g < a.G16) {
      if (a.stats_partial) a.stats_partial[(((long long)b * a.n_tiles + tile) * a.G16 + g) * 2 + which] = v;
      else stat_add(a.stats + ((long long)b * a.G16 + g) * 2 + which, v);
    }
  }
  if (!a.red_x) return;
  if (t < BN * 2) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) v += fr[w * BN * 2 + t];
    const int c = n0 + (t >> 1);
    float wv = 0.f;
    if (c < a.Cout) {
      grad_add(&a.red_r[((long long)b * a.Cout + c) * 2 + (t & 1)], v);
      wv = v * a.red_gamma[c];
    }
    fw[t] = wv;
  }
  __syncthreads();
  if (t < (BN / 16) * 2) {
    const int slab = t >> 1, which = t & 1;
    const int c0s = n0 + slab * 16;
    if (c0s < a.Cout) {
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) s2 += fw[(slab * 16 + j) * 2 + which];
      const int cpg = 16 * a.red_gmul;
      grad_add(&a.red_r[(long long)gridDim.z * a.Cout * 2 + ((long long)b * (a.Cout / cpg) + c0s / cpg) * 2 + which], s2);
    }
  }
  (void)NT;
}

// Waves of a workgroup that hold no output tile (split-K groups 1.. of k_igemm) must still meet the barriers of
// conv_epilogue: one in the LDS-staged vector path, one before the statistics fold.  Keep in step with conv_epilogue.
template <int BM, int BN, int NT>
__device__ __forceinline__ void conv_epilogue_idle(const ConvK& a) {
  const bool vec = (a.out_mode == 0 || (a.patch_c & 7) == 0) && !a.y_f32 && !a.res && (a.Cout & 7) == 0 && (a.y_ld & 7) == 0 && a.vec_ok;
  if (!vec && conv_use_vecf<BM, BN, NT>(a)) {      // fp32 staging path: one barrier, three more with statistics
    __syncthreads();
    if (a.stats) { __syncthreads(); __syncthreads(); __syncthreads(); }
    return;
  }
  if (vec) __syncthreads();
  if (vec && a.red_x) { __syncthreads(); __syncthreads(); __syncthreads(); }
  if (a.stats) __syncthreads();
}

// The narrow pointwise layers (pw_narrow.hip: Mlp.fc2 / fc1's data gradient at encoder stages 1-2).  NarrowGn: an optional
// GroupNorm + exact GELU applied to the bf16 input rows on their way in (Mlp.norm2 in front of fc2); xn: optional bf16 copy of the
// activated rows (what fc2's weight gradient reads).
struct NarrowGn {
  const crd_sum_t* stats; int gmul; const float* gamma; const float* beta; float count;
  bf16_t* xn; int xn_ld; long long xn_bstride;
};

// stats[b][g][which] += sum over tiles of the per-tile partials written by conv_epilogue
__global__ __launch_bounds__(256) void k_stats_finalize(const float* partial, int n_tiles, int G16, crd_sum_t* stats);


}  // namespace crdk

bool crd_pw_narrow_applicable(const crdk::ConvK& k, const crdk::NarrowGn* gn);      // pw_narrow.hip
int crd_pw_narrow(const crdk::ConvK& k, const crdk::NarrowGn* gn, int B, hipStream_t st);
