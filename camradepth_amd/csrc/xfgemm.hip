// Backward twin of gngemm.hip (round 6): the APPLY phase of a GroupNorm (+ exact GELU) backward folded into the A-operand path of
// the pointwise MFMA GEMM that consumes the gradient (gfx950, bf16 operands, fp32 accumulate).
//
//   y = epilogue( W * dx ),   dx = (gamma * g - S1 - xhat * S2) * rstd,   g = dy * mask * act'(xhat * gamma + beta)
//
// i.e. exactly what crd_gn_bwd_apply stores (the autograd of nn.GroupNorm + nn.GELU at
// src/models/simplified_attention.py:23-24,38-41,117-118).  In the encoder's backward chain every GroupNorm cost two grid-wide
// dependencies: producer (+ fused reduce) -> crd_gn_bwd_apply -> consumer.  The reduce needs the whole sample before the apply can start,
// so ONE boundary is inherent; the second is not: the consumer of dx is a pointwise data-gradient GEMM (Mlp.fc1's behind Mlp.norm1,
// attn.sr's patch scatter behind attn.norm) whose A rows can be produced on the fly from (dy, x) and the per-sample reduce sums.
// Per block and GroupNorm this removes one launch from the dependency chain and one read of the gradient tensor.
//
// Structure = k_gngemm_reg: both operands by buffer_load to registers two K-steps ahead (dy AND the GroupNorm's input x for the A rows),
// transformed in registers on their way into a two-stage XOR-swizzled LDS tile; per-sample coefficient table in LDS:
//   ACT 0:  dx = dy * ca + (x * cb + cc)                       ca = gamma mask rstd, cb = -rstd^2 S2, cc = -rstd S1 + mean rstd^2 S2
//   ACT 1:  dx = dy * GELU'(x * za + zb) * ca + (x * cb + cc)   za = gamma rstd, zb = beta - mean za
// dx is stored once (bf16, by the workgroups of column tile 0) when a weight gradient reads it; the GroupNorm's parameter gradients
// (sums over the samples of r) are added by the workgroups of sample 0 / column tile 0, as crd_gn_bwd_apply does.
// The epilogue (bias, accumulate, patch scatter, fused reduce of the NEXT GroupNorm's backward, GroupNorm sums) is conv_common.h's.
#include <stdlib.h>
#include "conv_common.h"

using namespace crdk;

namespace {

constexpr int BK = 64;

struct XfIn {
  const void* gx; int gx_f32; int gx_ld; long long gx_bstride;      // the GroupNorm's input [B][P][gx_ld] (bf16 or fp32)
  const crd_sum_t* stats; int gmul; const float* gamma; const float* beta; const float* mask;
  const crd_sum_t* r; float count; int B;                           // reduce sums [B][C][2] then [B][C/(16 gmul)][2]
  bf16_t* dx; int dx_ld; long long dx_bstride;                      // optional store of dx
  float* dgamma; float* dbeta;                                      // optional parameter gradients (+=)
};

// coefficient table of sample b: t4[c] = (ca, cb, cc, za), tz[c] = zb.  The per-GROUP quantities (mean, rstd from the forward sums in
// fp64, S1, S2 from the reduce sums) are computed once per group into `grp` (LDS) -- per channel they were C / cpg times the fp64 work and a
// dependent load chain per loop iteration at the head of every workgroup -- then a channel is two parameter loads and four FMAs.
__device__ __forceinline__ void build_table(const ConvK& a, const XfIn& xi, int b, f32x4* t4, float* tz, f32x4* grp) {
  const int C = a.Cin, cpg = 16 * xi.gmul, G = C / cpg;
  const crd_sum_t* stb = xi.stats + (long long)b * (C >> 4) * 2;
  const crd_sum_t* rgb = xi.r + (long long)xi.B * C * 2 + (long long)b * G * 2;
  const float inv_m = 1.f / xi.count;
  // parameters of this thread's first channels requested before the group pass waits for anything
  float ga0 = 0.f, be0 = 0.f, mk0 = 1.f;
  if ((int)threadIdx.x < C) {
    ga0 = xi.gamma[threadIdx.x]; be0 = xi.beta[threadIdx.x];
    if (xi.mask) mk0 = xi.mask[(long long)b * C + threadIdx.x];
  }
  for (int gI = threadIdx.x; gI < G; gI += 256) {
    float mean, rstd;
    gn_mean_rstd(stb, gI * xi.gmul, xi.gmul, xi.count, mean, rstd);
    grp[gI] = f32x4{mean, rstd, grad_get(rgb + gI * 2) * inv_m, grad_get(rgb + gI * 2 + 1) * inv_m};
  }
  lds_barrier();
  for (int c = threadIdx.x; c < C; c += 256) {
    const f32x4 gv = grp[c / cpg];
    const float mean = gv[0], rstd = gv[1], S1 = gv[2], S2 = gv[3];
    const bool first = c < 256;
    const float ga = first ? ga0 : xi.gamma[c], be = first ? be0 : xi.beta[c];
    const float mk = first ? mk0 : (xi.mask ? xi.mask[(long long)b * C + c] : 1.f);
    const float za = ga * rstd, q = rstd * rstd * S2;
    t4[c] = f32x4{za * mk, -q, mean * q - rstd * S1, za};
    tz[c] = be - mean * za;
  }
}

template <int TN>
__device__ __forceinline__ void mfma_slab(const bf16_t* sa, const bf16_t* sb, f32x16 (&acc)[1][TN], int wm, int wn, int l) {
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    bf16x8 af, bfr[TN];
    const int gi2 = ks * 2 + (l >> 5);
    {
      const int row = wm * 32 + (l & 31);
      af = *reinterpret_cast<const bf16x8*>(&sa[row * BK + ((gi2 ^ ((row >> 1) & 7)) << 3)]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = (wn * TN + j) * 32 + (l & 31);
      bfr[j] = *reinterpret_cast<const bf16x8*>(&sb[row * BK + ((gi2 ^ ((row >> 1) & 7)) << 3)]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[0][j], 0, 0, 0);
  }
}

template <int BM, int BN>
constexpr size_t epilogue_bytes() { return (size_t)BM * (BN + 8) * 4 + 256 * 16 * 4 + 2048; }

// 4 waves as 2 x 2, a 64 x (64 TN) tile per workgroup, one workgroup per (row tile, column tile, sample)
template <int TN, int ACT, int GXF32>
__global__ __launch_bounds__(256) void k_gnbwd_gemm(ConvK a, XfIn xi) {
  constexpr int WM = 2, WN = 2, TM = 1;
  constexpr int BM = 64, BN = WN * TN * 32;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4r;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sA = lds;                                   // [2][BM][BK]   (the epilogue's staging area aliases the tiles)
  bf16_t* sB = sA + 2 * BM * BK;                      // [2][BN][BK]
  f32x4* t4 = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(lds) + a.lds_bytes);   // [Cin], behind tiles / staging
  float* tz = reinterpret_cast<float*>(t4 + a.Cin);                                   // [Cin]
  f32x4* grp = reinterpret_cast<f32x4*>(tz + a.Cin);                                  // [Cin / 16] at most

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WN, wn = wv % WN;
  // workgroup -> (row tile, column tile): the column tiles of one row tile re-read (and re-transform) the same dy / x rows; workgroups go
  // to the XCDs round-robin by linear index, so indices i and i + 8 share an XCD and its L2.  With a.ny_tiles != 0 the launch is a
  // (8 ny ceil(row tiles / 8), 1, B) grid in which the column tiles of a row tile are 8 apart (a multiple of 8 per sample, so every sample
  // starts on XCD 0).  The plain (row tile, column tile, sample) grid put them on different XCDs and each fetched its rows from HBM:
  // stage 2's fc1 34.5 -> 30.0 us, stage 3's 17.5 -> 16.7 with this order.  Launches with fewer than 7 row tiles per sample (the sr
  // scatter: 2 row tiles x 32-64 column tiles) keep the plain grid -- their row tiles alone cannot cover the XCDs (8.5 -> 23 us when forced).
  // (The same order in k_igemm and k_gngemm_reg, whose operands arrive by LDS-DMA / are read once per launch: no gain, 13.8 -> 14.7 us on
  // fc2's data gradient at stage 3 -- not kept there.  profiles/r06_ab_xcd_tile_order.txt)
  const int ny = a.ny_tiles;
  const int bx = ny ? (int)(blockIdx.x / (8 * ny)) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ny ? (int)(blockIdx.x % (8 * ny)) >> 3 : (int)blockIdx.y;
  if (bx >= a.n_tiles) return;
  const int b = blockIdx.z, m0 = bx * BM, n0 = by * BN;
  const int r0 = 8 * wv + (l >> 3);                   // this thread's rows: r0 + 32 i; LDS slot l & 7 <- K granule g (k_igemm's swizzle)
  const int g = (l & 7) ^ ((r0 >> 1) & 7);
  const unsigned OOB = 0x80000000u;
  constexpr int esz = GXF32 ? 4 : 2;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + (long long)b * a.x_bstride), 0, (int)(a.x_bstride * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(xi.gx) + (long long)b * xi.gx_bstride * esz), 0, (int)(xi.gx_bstride * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(xi.dx ? xi.dx + (long long)b * xi.dx_bstride : reinterpret_cast<bf16_t*>(const_cast<void*>(xi.gx))), 0,
      xi.dx ? (int)(xi.dx_bstride * 2) : 0, 0x00020000);
  const int nK = (a.Ktot + BK - 1) / BK;
  const bool store_dx = xi.dx != nullptr && by == 0;

  // Loop-invariant BYTE offsets of this thread's rows in dy / x / dx, with the row's validity folded in (an invalid row starts at the
  // out-of-range offset, and stays out of range whatever is added): every load and store of the K loop is then UNCONDITIONAL.  Written
  // as `ok ? computed : OOB` inside the loop the compiler sank the multiply into a branch around the load -- two loads into the same
  // registers on two exec-mask paths -- and, no longer able to count what was in flight, waited vmcnt(0) in front of every use: the
  // two-slab prefetch was one slab deep (ISA of the first build; the conditional xn stores of k_gngemm_reg had the same effect).
  unsigned dyoff[A_IT], gxoff[A_IT], dxoff[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool ok = m < a.OHW;
    dyoff[i] = ok ? (unsigned)(m * a.x_ld * 2) : OOB;
    gxoff[i] = ok ? (unsigned)(m * xi.gx_ld * esz) : OOB;
    dxoff[i] = (ok && store_dx) ? (unsigned)(m * xi.dx_ld * 2) : OOB;
  }
  unsigned woff[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int ng = n0 + r0 + 32 * j;
    woff[j] = ng < a.Cout ? (unsigned)(ng * a.Ktot * 2) : OOB;
  }
  struct Regs { u32x4r d[A_IT]; u32x4r x[A_IT][GXF32 ? 2 : 1]; u32x4r w[B_IT]; };
  auto load_slab = [&](int kt, Regs& r) {
    const int kc = kt * BK + g * 8;
    const unsigned km = kc < a.Ktot ? 0u : OOB;          // K tail: out of range as well
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      r.d[i] = __builtin_amdgcn_raw_buffer_load_b128(rd, (dyoff[i] + (unsigned)(kc * 2)) | km, 0, 0);
      const unsigned xo = (gxoff[i] + (unsigned)(kc * esz)) | km;
      r.x[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo, 0, 0);
      if (GXF32) r.x[i][GXF32 ? 1 : 0] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo + 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j)
      r.w[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (woff[j] + (unsigned)(kc * 2)) | km, 0, 0);
  };
  auto store_slab = [&](int kt, int stage, const Regs& r) {
    const int kc = kt * BK + g * 8;
    const bool kok = kc < a.Ktot;
    const int kt_ = kok ? kc : 0;                      // (table reads stay inside the table on the K tail; the values are masked below)
    float ca[8], cb[8], cc[8], za[8], zb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const f32x4 v = t4[kt_ + j]; ca[j] = v[0]; cb[j] = v[1]; cc[j] = v[2]; za[j] = v[3]; }
    if (ACT == 1) {
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(tz + kt_), z1 = *reinterpret_cast<const f32x4*>(tz + kt_ + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { zb[j] = z0[j]; zb[4 + j] = z1[j]; }
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      float dv[8], xv[8], o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { dv[2 * j] = bf_lo(r.d[i][j]); dv[2 * j + 1] = bf_hi(r.d[i][j]); }
      if (GXF32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[j] = __uint_as_float(r.x[i][0][j]); xv[4 + j] = __uint_as_float(r.x[i][GXF32 ? 1 : 0][j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[2 * j] = bf_lo(r.x[i][0][j]); xv[2 * j + 1] = bf_hi(r.x[i][0][j]); }
      }
      const bool ok = kok && dyoff[i] != OOB;            // (out-of-range loads returned zeros; the constant term cc must go as well)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float gq = dv[j];
        if (ACT == 1) gq *= gelu_grad(xv[j] * za[j] + zb[j]);
        const float y = gq * ca[j] + (xv[j] * cb[j] + cc[j]);
        o[j] = ok ? y : 0.f;
      }
      const u32x4r q = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7])};
      *reinterpret_cast<u32x4r*>(sA + stage * BM * BK + (r0 + 32 * i) * BK + (l & 7) * 8) = q;
      // unconditional: rows / K granules that are not stored carry the out-of-range offset and the hardware drops them
      __builtin_amdgcn_raw_buffer_store_b128(q, rdx, (dxoff[i] + (unsigned)(kc * 2)) | (kok ? 0u : OOB), 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) *reinterpret_cast<u32x4r*>(sB + stage * BN * BK + (r0 + 32 * j) * BK + (l & 7) * 8) = r.w[j];
  };

  // everything the first two K-steps need is requested before anything is waited for: the table's inputs first, then slabs 0 and 1
  build_table(a, xi, b, t4, tz, grp);
  Regs r0s, r1s;
  load_slab(0, r0s);
  load_slab(1, r1s);                                  // (slabs past the end: every offset out of range -- zeros, no traffic)
  // the GroupNorm's parameter gradients: the workgroups of sample 0 / column tile 0 share the channels (crd_gn_bwd_apply's rule)
  if (b == 0 && by == 0 && xi.dgamma) {
    for (int c = bx * 256 + t; c < a.Cin; c += a.n_tiles * 256) {
      const float ob = xi.dbeta[c], og = xi.dgamma[c];
      long long g0, g1;
      sum_samples(xi.r, xi.B, a.Cin, c, g0, g1);
      xi.dbeta[c] = ob + (float)g0 * (1.f / GRAD_ONE);
      xi.dgamma[c] = og + (float)g1 * (1.f / GRAD_ONE);
    }
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (l & 31);
    const float bias_v = (a.bias && col < a.Cout) ? a.bias[(long long)b * a.bias_bstride + col] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc[0][j][rr] = bias_v;
  }
  lds_barrier();                                      // the table
  store_slab(0, 0, r0s);
  load_slab(2, r0s);
  lds_barrier();
  // Steady state, two slabs per trip, NO conditional memory operation inside: a load or store under `if (kt + 3 < nK)` -- even a
  // wave-uniform one -- makes the number of requests in flight path-dependent, and the compiler then waits as if the skipped path had
  // been taken: every use drained the younger slab's requests as well (vmcnt(3..0) where vmcnt(8) was meant).  Slabs past the end carry
  // out-of-range offsets instead (zeros in, nothing out), an odd slab count ends in a single-slab tail.  (Four slabs in flight were
  // tried: 256 registers or 36-92 spilled, one workgroup per CU -- not kept.)
  int kt = 0;
  for (; kt + 2 <= nK; kt += 2) {
    store_slab(kt + 1, 1, r1s);
    load_slab(kt + 3, r1s);
    mfma_slab<TN>(sA, sB, acc, wm, wn, l);
    lds_barrier();
    store_slab(kt + 2, 0, r0s);
    load_slab(kt + 4, r0s);
    mfma_slab<TN>(sA + BM * BK, sB + BN * BK, acc, wm, wn, l);
    lds_barrier();
  }
  if (kt < nK) {
    mfma_slab<TN>(sA, sB, acc, wm, wn, l);
    lds_barrier();
  }
  conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, bx, lds,
                                [&](int i, int rr, bool& valid, int& row) { row = m0 + (wm * TM + i) * 32 + rr; valid = row < a.OHW; },
                                [&](int rl, bool& valid, int& row) { row = m0 + rl; valid = row < a.OHW; });
}

template <int TN, int ACT, int GXF32>
int launch(const ConvK& k0, const XfIn& xi, int B, hipStream_t st) {
  constexpr int BM = 64, BN = 64 * TN;
  ConvK k = k0;
  k.n_tiles = cdiv(k.OHW, BM);
  size_t tiles = (size_t)2 * (BM + BN) * BK * 2;
  if (tiles < epilogue_bytes<BM, BN>()) tiles = epilogue_bytes<BM, BN>();
  tiles = (tiles + 255) / 256 * 256;
  const size_t lds = tiles + (size_t)k.Cin * 20 + (size_t)(k.Cin / 16) * 16;
  CRD_UNSUPPORTED(lds <= 160 * 1024, "crd_gn_bwd_conv: coefficient table does not fit in LDS");
  static bool attr_done = false;
  if (!attr_done) {
    crd_reserve_lds(reinterpret_cast<const void*>(&k_gnbwd_gemm<TN, ACT, GXF32>), 160 * 1024, "k_gnbwd_gemm");
    attr_done = true;
  }
  k.lds_bytes = (int)tiles;
  const int ny = cdiv(k.Cout, BN);
  const bool xcd = ny >= 2 && k.n_tiles >= 7;
  k.ny_tiles = xcd ? ny : 0;
  if (xcd) hipLaunchKernelGGL((k_gnbwd_gemm<TN, ACT, GXF32>), dim3(8 * ny * cdiv(k.n_tiles, 8), 1, B), dim3(256), lds, st, k, xi);
  else hipLaunchKernelGGL((k_gnbwd_gemm<TN, ACT, GXF32>), dim3(k.n_tiles, ny, B), dim3(256), lds, st, k, xi);
  CRD_LAUNCH_CHECK("crd_gn_bwd_conv");
  return CRD_OK;
}

template <int ACT, int GXF32>
int dispatch(const ConvK& k, const XfIn& xi, int B, hipStream_t st) {
  // 64-column tiles when 128-column ones would not cover the chip, and wherever the fused reduce of the epilogue needs threads that
  // keep their columns on a ragged last tile (crd_conv_igemm's rule: 64 < Cout <= 96 and 128 < Cout <= 160 have no 128-column form)
  const long long big_tiles = (long long)cdiv(k.OHW, 64) * cdiv(k.Cout, 128) * B;
  const bool red_narrow = k.red_x && !((k.Cout > 96 && k.Cout <= 128) || k.Cout > 160);
  if (k.Cout <= 64 || big_tiles < 256 || red_narrow || k.out_mode == 1) return launch<1, ACT, GXF32>(k, xi, B, st);
  return launch<2, ACT, GXF32>(k, xi, B, st);
}

}  // namespace

// ConvK of a pointwise data-gradient GEMM from the descriptor
static int fill_convk(const crd_conv_desc* d, ConvK& k, const char* who) {
  k.x = reinterpret_cast<const bf16_t*>(d->x) + d->x_coff; k.x_ld = d->x_ld;
  k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin; k.x_bstride = (long long)d->IH * d->IW * d->x_ld;
  k.w = reinterpret_cast<const bf16_t*>(d->w);
  k.Cout = d->Cout; k.KW = 1; k.stride = 1; k.pad = 0; k.Ktot = d->Cin;
  k.OW = d->OW; k.OHW = d->OH * d->OW; k.gather_mode = 0;
  k.y_ld = d->y_ld; k.y_f32 = 0;
  k.out_mode = d->out_mode; k.patch_k = d->patch_k; k.patch_c = d->patch_c;
  int YH = d->OH, YW = d->OW;
  if (d->out_mode == 1) { YH = d->OH * d->patch_k; YW = d->OW * d->patch_k; }
  k.YW = YW;
  k.y_bstride = (long long)YH * YW * d->y_ld;
  k.y = (void*)(reinterpret_cast<bf16_t*>(d->y) + d->y_coff);
  k.bias = d->bias; k.bias_bstride = d->bias_bstride; k.act = 0;
  k.res = nullptr; k.res_ld = 0; k.res_bstride = 0; k.res_scale = nullptr;
  k.accumulate = d->accumulate; k.stats = d->stats; k.G16 = d->Cout / 16;
  k.stats_partial = nullptr; k.n_tiles = 0; k.col0 = 0; k.chan = nullptr;
  k.vec_ok = (d->y_coff % 8 == 0) && ((reinterpret_cast<uintptr_t>(d->y) & 15) == 0);
  k.vecf_ok = 0;
  k.lds_bytes = 0;
  k.red_x = d->red_x; k.red_x_f32 = d->red_x_f32; k.red_x_ld = d->red_x_ld;
  k.red_x_bstride = (long long)YH * YW * d->red_x_ld;
  k.red_stats = d->red_stats; k.red_gamma = d->red_gamma; k.red_beta = d->red_beta; k.red_gmul = d->red_gmul;
  k.red_act = d->red_act; k.red_r = d->red_r;
  k.dbg = 0;
  if (d->red_x) {
    CRD_CHECK_ARG(d->red_stats && d->red_gamma && d->red_beta && d->red_r && d->red_gmul >= 1 && d->red_x_ld % 8 == 0 &&
                  (d->Cout / 16) % d->red_gmul == 0, "%s: incomplete fused-reduce arguments", who);
    CRD_UNSUPPORTED(d->Cout % 16 == 0 && d->out_mode == 0 && d->y_ld % 8 == 0 && k.vec_ok,
                    "%s: the fused GroupNorm-backward reduce needs a plain-layout bf16 vector-path output", who);
  }
  return CRD_OK;
}

extern "C" int crd_gn_bwd_conv(const crd_conv_desc* d, const crd_gn_bwd_input* n, crd_stream_t stream) {
  CRD_CHECK_ARG(d && n && d->x && d->w && d->y && n->gx && n->stats && n->gamma && n->beta && n->r, "crd_gn_bwd_conv: null pointer");
  CRD_CHECK_ARG(d->Cin % 16 == 0 && d->x_ld % 8 == 0 && d->x_coff % 8 == 0 && n->gx_ld % 8 == 0,
                "crd_gn_bwd_conv: Cin must be a multiple of 16, x_ld / x_coff / gx_ld of 8");
  CRD_CHECK_ARG(d->B > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0, "crd_gn_bwd_conv: bad dims");
  CRD_UNSUPPORTED(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->IH == d->OH && d->IW == d->OW,
                  "crd_gn_bwd_conv: pointwise data gradients only (patch scatter through out_mode = 1)");
  CRD_CHECK_ARG(n->gmul >= 1 && (d->Cin / 16) % n->gmul == 0 && (n->act == 0 || n->act == 1), "crd_gn_bwd_conv: bad GroupNorm arguments");
  CRD_CHECK_ARG(!d->y_f32 && !d->res && !d->act && !d->chan_sums && !d->stats_partial,
                "crd_gn_bwd_conv: bf16 output without residual / activation / channel sums");
  CRD_CHECK_ARG(d->out_mode == 0 || (d->out_mode == 1 && d->patch_k > 0 && d->patch_c > 0 && d->Cout == d->patch_k * d->patch_k * d->patch_c),
                "crd_gn_bwd_conv: bad patch-scatter dims");
  CRD_CHECK_ARG(!d->stats || d->Cout % 16 == 0, "crd_gn_bwd_conv: stats need Cout %% 16 == 0");
  CRD_CHECK_ARG(!n->dx || (n->dx_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(n->dx) & 15) == 0), "crd_gn_bwd_conv: dx rows must be 16-byte aligned");
  CRD_CHECK_ARG((n->dgamma == nullptr) == (n->dbeta == nullptr), "crd_gn_bwd_conv: dgamma and dbeta come together");
  CRD_UNSUPPORTED((long long)d->Cout * d->Cin < (1ll << 30) && d->Cin <= 4096 && (long long)d->IH * d->IW * d->x_ld * 2 < (1ll << 31) &&
                  (long long)d->IH * d->IW * n->gx_ld * (n->gx_f32 ? 4 : 2) < (1ll << 31), "crd_gn_bwd_conv: tensor too large for 32-bit byte offsets");
  ConvK k;
  { const int rc = fill_convk(d, k, "crd_gn_bwd_conv"); if (rc != CRD_OK) return rc; }
  CRD_CHECK_ARG((reinterpret_cast<uintptr_t>(k.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(n->gx) & 15) == 0 && (!n->gx_f32 || n->gx_ld % 4 == 0),
                "crd_gn_bwd_conv: dy / x rows must be 16-byte aligned");
  XfIn xi;
  xi.gx = n->gx; xi.gx_f32 = n->gx_f32; xi.gx_ld = n->gx_ld; xi.gx_bstride = (long long)d->IH * d->IW * n->gx_ld;
  xi.stats = n->stats; xi.gmul = n->gmul; xi.gamma = n->gamma; xi.beta = n->beta; xi.mask = n->mask;
  xi.r = n->r; xi.count = (float)d->IH * (float)d->IW * 16.f * (float)n->gmul; xi.B = d->B;
  xi.dx = reinterpret_cast<bf16_t*>(n->dx); xi.dx_ld = n->dx_ld; xi.dx_bstride = (long long)d->IH * d->IW * n->dx_ld;
  xi.dgamma = n->dgamma; xi.dbeta = n->dbeta;
  hipStream_t st = as_stream(stream);
  if (n->gx_f32) return n->act ? dispatch<1, 1>(k, xi, d->B, st) : dispatch<0, 1>(k, xi, d->B, st);
  return n->act ? dispatch<1, 0>(k, xi, d->B, st) : dispatch<0, 0>(k, xi, d->B, st);
}
