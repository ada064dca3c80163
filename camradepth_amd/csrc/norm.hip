// GroupNorm family for gfx950: statistics, apply(+GELU,+dropout mask), backward.
// All kernels are HBM-bound streaming passes: one thread owns 8 consecutive channels (16-byte bf16
// / 32-byte fp32 accesses) of a pixel and keeps its channel group fixed while striding over pixels,
// so per-channel partial sums stay in registers; workgroup partials are merged in LDS and flushed
// with one global atomic per value.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int U = 4;   // independent 16-byte loads in flight per thread in the streaming loops

// thread -> (channel granule cg, pixel lane pl); CG = C/8 granules; PL = TPB/CG pixel lanes
struct Map {
  int cg, pl, PL; bool active;
  __device__ Map(int C) {
    int CG = C >> 3;
    PL = TPB / CG;
    if (PL < 1) PL = 1;
    active = threadIdx.x < PL * CG;
    cg = threadIdx.x % CG;
    pl = threadIdx.x / CG;
  }
};

// sm holds `rows` rows of n floats (one per pixel lane); leaves their sum in row 0.  LDS float atomics for this merge
// cost 18 us of a 40 us kernel (ablation in tools/bench_gn.py): every lane of a wave hits one of four banks.
__device__ __forceinline__ void fold_rows(float* sm, int n, int rows) {
  for (int i = threadIdx.x; i < n; i += TPB) {
    float a = sm[i];
    for (int r = 1; r < rows; ++r) a += sm[(long long)r * n + i];
    sm[i] = a;
  }
  __syncthreads();
}

template <int XF>
__global__ __launch_bounds__(TPB) void k_gn_stats(const void* x, int x_f32, int x_ld, long long P, int C, int chunk,
                                                  crd_sum_t* stats, crd_sum_t* chan) {
  extern __shared__ float sm[];  // [PL][C][2]: one row of channel sums per pixel lane, folded after the barrier
  const int b = blockIdx.y;
  Map m(C);
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
  if (m.active) {
    long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    for (long long p = p0 + m.pl; p < p1; p += (long long)U * m.PL) {   // U independent loads in flight per thread
      float v[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long long pp = p + (long long)u * m.PL;
        if (pp >= p1) pp = p1 - 1;
        load8t<XF>(x, ((long long)b * P + pp) * x_ld + m.cg * 8, v[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + (long long)u * m.PL < p1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { s[j] += v[u][j]; ss[j] += v[u][j] * v[u][j]; }
        }
    }
    float4* row = reinterpret_cast<float4*>(sm + (long long)m.pl * 2 * C + m.cg * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) row[j] = make_float4(s[2 * j], ss[2 * j], s[2 * j + 1], ss[2 * j + 1]);
  }
  __syncthreads();
  fold_rows(sm, 2 * C, m.PL);
  if (chan)
    for (int i = threadIdx.x; i < 2 * C; i += TPB) stat_add(&chan[(long long)b * C * 2 + i], sm[i]);
  if (stats)
    for (int g = threadIdx.x; g < (C >> 4) * 2; g += TPB) {
      int slab = g >> 1, which = g & 1;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) a += sm[(slab * 16 + j) * 2 + which];
      stat_add(&stats[((long long)b * (C >> 4) + slab) * 2 + which], a);
    }
}

// crd_attn_out_residual with the GroupNorm statistics of its output (the block's norm2 input) folded in:
// x1 = x + dp[b]*bf16(u[b][c]*S[b][n] + bp[c]);  stats[b][c/16] += (sum, sumsq) of x1.  Thread mapping and fold as in
// k_gn_stats.
__global__ __launch_bounds__(TPB) void k_attn_out_residual_stats(const float* x, const float* u, const float* S, const float* bp,
                                                                 const float* dp, long long P, int C, int chunk, float* x1,
                                                                 crd_sum_t* stats) {
  extern __shared__ float sm[];  // [PL][C][2]
  const int b = blockIdx.y;
  Map m(C);
  if (m.active) {
    float s[8], ss[8], uu[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
    const float dps = dp ? dp[b] : 1.f;
    load8(u, (long long)b * C + m.cg * 8, 1, uu);
    load8(bp, m.cg * 8, 1, bb);
    long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    for (long long p = p0 + m.pl; p < p1; p += (long long)U * m.PL) {
      float v[U][8], sv[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        long long pp = p + (long long)q * m.PL;
        if (pp >= p1) pp = p1 - 1;
        load8(x, ((long long)b * P + pp) * C + m.cg * 8, 1, v[q]);
        sv[q] = S[(long long)b * P + pp];
      }
#pragma unroll
      for (int q = 0; q < U; ++q)
        if (p + (long long)q * m.PL < p1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[q][j] += dps * bf_round(uu[j] * sv[q] + bb[j]);
            s[j] += v[q][j]; ss[j] += v[q][j] * v[q][j];
          }
          store8_f32(x1, ((long long)b * P + p + (long long)q * m.PL) * C + m.cg * 8, v[q]);
        }
    }
    float4* row = reinterpret_cast<float4*>(sm + (long long)m.pl * 2 * C + m.cg * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) row[j] = make_float4(s[2 * j], ss[2 * j], s[2 * j + 1], ss[2 * j + 1]);
  }
  __syncthreads();
  fold_rows(sm, 2 * C, m.PL);
  for (int g = threadIdx.x; g < (C >> 4) * 2; g += TPB) {
    int slab = g >> 1, which = g & 1;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) a += sm[(slab * 16 + j) * 2 + which];
    stat_add(&stats[((long long)b * (C >> 4) + slab) * 2 + which], a);
  }
}

template <int XF, int YF, int ACT>
__global__ __launch_bounds__(TPB) void k_gn_apply(const void* x, int x_f32, int x_ld, long long P, int C, int chunk,
                                                  const crd_sum_t* stats, int gmul, const float* gamma, const float* beta,
                                                  int act, const float* mask, void* y, int y_ld, int y_f32, float y_inv_scale,
                                                  void* y2, int y2_ld) {
  const int b = blockIdx.y;
  Map m(C);
  if (!m.active) return;
  const int c0 = m.cg * 8;
  long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
  if (p1 > P) p1 = P;
  long long p = p0 + m.pl;
  // Every load the first batch needs is issued before anything is waited for -- the pixels, then the parameters, then
  // the statistics: four dependent round trips (kernel arguments, statistics, parameters, data) were most of the ~5 us
  // a launch on a small grid takes.
  float v[U][8];
  auto load_batch = [&](long long q) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long pp = q + (long long)u * m.PL;
      if (pp >= p1) pp = p1 - 1;
      load8t<XF>(x, ((long long)b * P + pp) * x_ld + c0, v[u]);
    }
  };
  if (p < p1) load_batch(p);
  float ga[8], be[8], mk[8];
  load8(gamma, c0, 1, ga);
  load8(beta, c0, 1, be);
  if (mask) load8(mask, (long long)b * C + c0, 1, mk);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) mk[j] = 1.f;
  }
  float mean, rstd;
  const int grp = (c0 >> 4) / gmul;
  gn_mean_rstd(stats + (long long)b * (C >> 4) * 2, grp * gmul, gmul, (float)P * 16.f * gmul, mean, rstd);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] *= rstd;
    be[j] -= mean * ga[j];
  }
  for (bool first = true; p < p1; p += (long long)U * m.PL, first = false) {
    if (!first) load_batch(p);                 // (the first batch was requested at the top of the kernel)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long pp = p + (long long)u * m.PL;
      if (pp < p1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float w = v[u][j] * ga[j] + be[j];
          if (ACT == 1) w = gelu_exact(w);
          v[u][j] = w * mk[j];
        }
        if (YF == 3) {                                   // YF: 0 bf16, 1 fp32, 2 e4m3, 3 e4m3 (y) and bf16 (y2)
          store8_fp8(y, ((long long)b * P + pp) * y_ld + c0, v[u], y_inv_scale);
          store8_bf16(y2, ((long long)b * P + pp) * y2_ld + c0, v[u]);
        } else if (YF == 2) store8_fp8(y, ((long long)b * P + pp) * y_ld + c0, v[u], y_inv_scale);
        else if (YF) store8_f32(reinterpret_cast<float*>(y), ((long long)b * P + pp) * y_ld + c0, v[u]);
        else store8_bf16(y, ((long long)b * P + pp) * y_ld + c0, v[u]);
      }
    }
  }
}

// r[b][c] = (sum_p g, sum_p g*xhat), g = dy*mask*act'(u)
// gridDim.z = channel slices of Cs channels each (a multiple of the group width): a workgroup then covers Cs channels of
// gridDim.z times the pixels instead of all C channels -- the same number of workgroups, 2 Cs instead of 2 C atomics each
// (Mlp.norm2 of stages 1-2: 1-1.7 M 64-bit atomics per launch were ~10 of its 32-36 us).
template <int XF, int DF, int ACT>
__global__ __launch_bounds__(TPB) void k_gn_bwd_reduce(const void* x, int x_f32, int x_ld, const void* dy, int dy_f32,
                                                       int dy_ld, long long P, int C, int Cs, int chunk, const crd_sum_t* stats,
                                                       int gmul, const float* gamma, const float* beta, int act,
                                                       const float* mask, crd_sum_t* r, float* partial) {
  extern __shared__ float sm[];  // [PL][Cs][2]
  const int b = blockIdx.y, cs = blockIdx.z * Cs;
  Map m(Cs);
  if (m.active) {
    const int c0 = cs + m.cg * 8;
    long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    long long p = p0 + m.pl;
    // first batch, parameters and statistics all in flight before the first wait (see k_gn_apply)
    float v[U][8], d[U][8];
    auto load_batch = [&](long long q) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        long long pp = q + (long long)u * m.PL;
        if (pp >= p1) pp = p1 - 1;
        load8t<XF>(x, ((long long)b * P + pp) * x_ld + c0, v[u]);
        load8t<DF>(dy, ((long long)b * P + pp) * dy_ld + c0, d[u]);
      }
    };
    if (p < p1) load_batch(p);
    float ga[8], be[8], mk[8], s0[8], s1[8];
    load8(gamma, c0, 1, ga);
    load8(beta, c0, 1, be);
    if (mask) load8(mask, (long long)b * C + c0, 1, mk);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!mask) mk[j] = 1.f;
      s0[j] = s1[j] = 0.f;
    }
    float mean, rstd;
    const int grp = (c0 >> 4) / gmul;
    gn_mean_rstd(stats + (long long)b * (C >> 4) * 2, grp * gmul, gmul, (float)P * 16.f * gmul, mean, rstd);
    for (bool first = true; p < p1; p += (long long)U * m.PL, first = false) {
      if (!first) load_batch(p);               // (the first batch was requested at the top)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (p + (long long)u * m.PL < p1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float xh = (v[u][j] - mean) * rstd;
            float g = d[u][j] * mk[j];
            if (ACT == 1) g *= gelu_grad(xh * ga[j] + be[j]);
            s0[j] += g; s1[j] += g * xh;
          }
        }
    }
    float4* row = reinterpret_cast<float4*>(sm + (long long)m.pl * 2 * Cs + m.cg * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) row[j] = make_float4(s0[2 * j], s1[2 * j], s0[2 * j + 1], s1[2 * j + 1]);
  }
  __syncthreads();
  fold_rows(sm, 2 * Cs, m.PL);
  if (partial) {   // plain stores of this workgroup's sums; k_gn_bwd_finalize folds them (no contended atomics)
    float* dst = partial + ((long long)b * gridDim.x + blockIdx.x) * 2 * C + cs * 2;
    for (int i = threadIdx.x; i < 2 * Cs; i += TPB) dst[i] = sm[i];
    return;
  }
  // per-channel sums out; in the same pass weight them with gamma in place (all threads, coalesced gamma loads) for the
  // group sums below -- a serial walk of one thread over the 64..128 channels of a group, with a dependent gamma load
  // per channel, was 5 us of this kernel's 13 us floor
  for (int i = threadIdx.x; i < 2 * Cs; i += TPB) {
    const float v = sm[i];
    grad_add(&r[((long long)b * C + cs) * 2 + i], v);
    sm[i] = v * gamma[cs + (i >> 1)];
  }
  __syncthreads();
  // per-group sums S1 = sum_c gamma_c r0, S2 = sum_c gamma_c r1 (stored after the [B][C][2] block of r)
  const int cpg = 16 * gmul, G = C / cpg, Gs = Cs / cpg;
  crd_sum_t* rg = r + (long long)gridDim.y * C * 2 + ((long long)b * G + cs / cpg) * 2;
  for (int gi = threadIdx.x; gi < 2 * Gs; gi += TPB) {
    const int grp = gi >> 1, which = gi & 1;
    float acc = 0.f;
    for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) acc += sm[c * 2 + which];
    grad_add(&rg[grp * 2 + which], acc);
  }
}

// r[b][c][0..1] = sum over workgroup partials; rg[b][grp] += sum_c gamma_c r[b][c]  (64 channels per workgroup)
__global__ __launch_bounds__(TPB) void k_gn_bwd_finalize(const float* partial, int nblk, int C, int gmul, const float* gamma,
                                                         crd_sum_t* r, int B) {
  __shared__ float sm[4][64][2];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (c < C)
    for (int k = lane; k < nblk; k += 4) {
      const float* p = partial + ((long long)b * nblk + k) * 2 * C + c * 2;
      s0 += p[0]; s1 += p[1];
    }
  sm[lane][threadIdx.x & 63][0] = s0; sm[lane][threadIdx.x & 63][1] = s1;
  __syncthreads();
  if (lane == 0 && c < C) {
    s0 = sm[0][threadIdx.x][0] + sm[1][threadIdx.x][0] + sm[2][threadIdx.x][0] + sm[3][threadIdx.x][0];
    s1 = sm[0][threadIdx.x][1] + sm[1][threadIdx.x][1] + sm[2][threadIdx.x][1] + sm[3][threadIdx.x][1];
    r[((long long)b * C + c) * 2] = to_fx(s0, GRAD_ONE);          // fixed summation order above: reproducible
    r[((long long)b * C + c) * 2 + 1] = to_fx(s1, GRAD_ONE);
    const float g = gamma[c];
    sm[0][threadIdx.x][0] = g * s0; sm[0][threadIdx.x][1] = g * s1;
  }
  __syncthreads();
  // group sums: 16-channel slabs of this 64-channel window, added into the group's accumulator
  if (threadIdx.x < 8) {
    const int slab = threadIdx.x >> 1, which = threadIdx.x & 1, c0 = blockIdx.x * 64 + slab * 16;
    if (c0 < C) {
      float a = 0.f;
      for (int j = 0; j < 16; ++j) a += sm[0][slab * 16 + j][which];
      const int cpg = 16 * gmul;
      grad_add(r + (long long)B * C * 2 + ((long long)b * (C / cpg) + c0 / cpg) * 2 + which, a);
    }
  }
}

template <int XF, int DF, int ACT>
__global__ __launch_bounds__(TPB) void k_gn_bwd_apply(const void* x, int x_f32, int x_ld, const void* dy, int dy_f32,
                                                      int dy_ld, long long P, int C, int chunk, const crd_sum_t* stats,
                                                      int gmul, const float* gamma, const float* beta, int act,
                                                      const float* mask, const crd_sum_t* r, float* dgamma, float* dbeta,
                                                      void* dx, int dx_f32, int dx_ld, int dx_acc, int B, void* dx2, int dx2_ld,
                                                      const float* scale2, unsigned char* dx8, int dx8_ld, const float* scale8,
                                                      unsigned* amax8) {
  const int b = blockIdx.y;
  Map m(C);
  const int c0 = m.cg * 8;
  // dx8 (config 5, fp8 data gradients): an e4m3 copy of the bf16 gradient just stored, e4m3(bf16(dx) / *scale8), and the running
  // max |bf16(dx)| of the launch into one of CRD_FP8_AMAX_SLOTS slots (u32 max on the bit pattern of a non-negative float)
  __shared__ float s_amax[TPB / 64];
  const float inv8 = (dx8 && scale8) ? 1.f / fmaxf(*scale8, 1e-30f) : 0.f;
  float amax = 0.f;
  long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
  if (p1 > P) p1 = P;
  long long p = p0 + m.pl;
  // first batch, parameters, reduce sums and statistics all in flight before the first wait (see k_gn_apply)
  float v[U][8], d[U][8];
  auto load_batch = [&](long long q) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long pp = q + (long long)u * m.PL;
      if (pp >= p1) pp = p1 - 1;
      load8t<XF>(x, ((long long)b * P + pp) * x_ld + c0, v[u]);
      load8t<DF>(dy, ((long long)b * P + pp) * dy_ld + c0, d[u]);
    }
  };
  if (m.active && p < p1) load_batch(p);
  if (b == 0 && dgamma) {     // parameter gradients: the workgroups of sample 0 share the channels (one workgroup doing all
                              // of them put B x C dependent loads in front of its own pixels: +5 us on the small grids)
    for (int c = blockIdx.x * TPB + threadIdx.x; c < C; c += gridDim.x * TPB) {
      const float ob = dbeta[c], og = dgamma[c];            // (requested with the sums, not behind them)
      long long g0, g1;
      sum_samples(r, B, C, c, g0, g1);
      dbeta[c] = ob + (float)g0 * (1.f / GRAD_ONE);
      dgamma[c] = og + (float)g1 * (1.f / GRAD_ONE);
    }
  }
  if (!m.active && !amax8) return;
  if (m.active) {
  float ga[8], be[8], mk[8];
  load8(gamma, c0, 1, ga);
  load8(beta, c0, 1, be);
  if (mask) load8(mask, (long long)b * C + c0, 1, mk);
  else {
#pragma unroll
    for (int j = 0; j < 8; ++j) mk[j] = 1.f;
  }
  const int grp = (c0 >> 4) / gmul;
  const int cpg = 16 * gmul;
  const float inv_m = 1.f / ((float)P * cpg);
  const crd_sum_t* rg = r + (long long)B * C * 2 + ((long long)b * (C / cpg) + grp) * 2;
  const float S1 = grad_get(rg) * inv_m, S2 = grad_get(rg + 1) * inv_m;
  float mean, rstd;
  gn_mean_rstd(stats + (long long)b * (C >> 4) * 2, grp * gmul, gmul, (float)P * cpg, mean, rstd);
  const float sc2 = (dx2 && scale2) ? scale2[b] : 1.f;
  for (bool first = true; p < p1; p += (long long)U * m.PL, first = false) {
    if (!first) load_batch(p);                 // (the first batch was requested at the top of the kernel)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long pp = p + (long long)u * m.PL;
      if (pp >= p1) continue;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xh = (v[u][j] - mean) * rstd;
        float g = d[u][j] * mk[j];
        if (ACT == 1) g *= gelu_grad(xh * ga[j] + be[j]);
        o[j] = (ga[j] * g - S1 - xh * S2) * rstd;
      }
      const long long off = ((long long)b * P + pp) * dx_ld + c0;
      if (dx_f32) {
        float* q = reinterpret_cast<float*>(dx);
        if (dx_acc) {
          float w[8];
          load8(dx, off, 1, w);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += w[j];
        }
        store8_f32(q, off, o);
      } else {
        if (dx_acc) {
          float w[8];
          load8(dx, off, 0, w);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += w[j];
        }
        store8_bf16(dx, off, o);
      }
      if (dx8) {          // e4m3 copy of the stored bf16 gradient + its running amax
        // The bf16 values come from the hardware converter (v_cvt_pk_bf16_f32, what store8_bf16 above wrote) and are unpacked once for
        // BOTH the amax and the e4m3 conversion: as software roundings per use (bf_round = ~5 integer operations, twice per element)
        // they were 80 of this path's ~110 vector operations per 8 elements and made crd_gn_bwd_apply_fp8 459 us where the plain apply takes
        // 318 (B = 16, 128 channels): three quarters of what the e4m3 data gradients gain (profiles/r06_c5_decoder_backward_chain_*).
        const uint32_t w0 = pack_bf2(o[0], o[1]), w1 = pack_bf2(o[2], o[3]), w2 = pack_bf2(o[4], o[5]), w3 = pack_bf2(o[6], o[7]);
        const float r[8] = {bf_lo(w0), bf_hi(w0), bf_lo(w1), bf_hi(w1), bf_lo(w2), bf_hi(w2), bf_lo(w3), bf_hi(w3)};
        // a NaN / infinite gradient would vanish here -- fmaxf drops NaN, the e4m3 conversion clamps to +-448 -- while the bf16 tensor next
        // to it carries it on: raise the sticky non-finite flag instead, so that the step reports NaN like the bf16 path.
        // (one test per 8 values: a sum is non-finite iff a term is, or the finite terms overflow -- 8 values near the bf16 maximum)
        const float chk = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        if (!(fabsf(chk) < 3.0e38f)) crd_tu_nonfinite = 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(r[j]));
        uint2 q8;
        q8.x = pack_fp8x4(r[0] * inv8, r[1] * inv8, r[2] * inv8, r[3] * inv8);
        q8.y = pack_fp8x4(r[4] * inv8, r[5] * inv8, r[6] * inv8, r[7] * inv8);
        *reinterpret_cast<uint2*>(dx8 + ((long long)b * P + pp) * dx8_ld + c0) = q8;
      }
      if (dx2) {          // second copy of the finished gradient: bf16(scale2[b] * dx)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= sc2;
        store8_bf16(dx2, ((long long)b * P + pp) * dx2_ld + c0, o);
      }
    }
  }
  }
  if (amax8) {            // one u32 max per workgroup, spread over the slots (same-address atomics serialise chip-wide: ~13 ns each)
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o_));
    if ((threadIdx.x & 63) == 0) s_amax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float a_ = s_amax[0];
      for (int w_ = 1; w_ < TPB / 64; ++w_) a_ = fmaxf(a_, s_amax[w_]);
      if (a_ > 0.f) atomicMax(amax8 + ((blockIdx.x + blockIdx.y * gridDim.x) % CRD_FP8_AMAX_SLOTS), __float_as_uint(a_));
    }
  }
}

inline void grid_for(long long P, int C, int B, dim3& grid, int& chunk, bool reduce = false) {
  int CG = C >> 3;
  int PL = TPB / CG; if (PL < 1) PL = 1;
  long long per_block = (long long)PL * 8;         // >= 8 pixels per pixel-lane (two batches of U loads)
  long long nblk = (P + per_block - 1) / per_block;
  static int minpix = -1, small = 256;
  if (minpix < 0) {
    minpix = crd_dev_int("CRD_GN_MINPIX", 2);
    small = crd_dev_int("CRD_GN_SMALL", 256);
  }
  if (nblk * B < small && minpix < 8) {              // small grids: fewer pixels per lane rather than idle CUs
    per_block = (long long)PL * minpix;
    nblk = (P + per_block - 1) / per_block;
  }
  static int cap_r = -1, cap_e = -1;
  if (cap_r < 0) { cap_r = crd_dev_int("CRD_GN_CAP_R", 768); cap_e = crd_dev_int("CRD_GN_CAP_E", 4096); }
  long long cap = (reduce ? cap_r : cap_e) / (B > 0 ? B : 1); if (cap < 1) cap = 1;
  if (nblk > cap) nblk = cap;
  if (nblk < 1) nblk = 1;
  chunk = (int)((P + nblk - 1) / nblk);
  nblk = (P + chunk - 1) / chunk;
  grid = dim3((unsigned)nblk, (unsigned)B);
}

int check_common(const char* name, int x_ld, int x_coff, int C, int x_f32) {
  if (C % 16 != 0 || C > 2048) { crd_set_error("%s: C must be a multiple of 16 and <= 2048 (got %d)", name, C); return CRD_E_UNSUPPORTED; }
  if (x_ld % 8 || x_coff % 8) { crd_set_error("%s: ld/coff must be multiples of 8", name); return CRD_E_INVALID; }
  (void)x_f32;
  return CRD_OK;
}

// LDS of the reducing kernels: one row of 2*C sums per pixel lane
inline size_t lds_rows(int C) {
  int PL = TPB / (C >> 3); if (PL < 1) PL = 1;
  return (size_t)PL * 2 * C * sizeof(float);
}

inline const void* off_ptr(const void* p, int f32, int coff) {
  return f32 ? (const void*)(reinterpret_cast<const float*>(p) + coff) : (const void*)(reinterpret_cast<const bf16_t*>(p) + coff);
}

}  // namespace

extern "C" int crd_gn_stats(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C,
                            crd_sum_t* stats, crd_sum_t* chan_sums, crd_stream_t stream) {
  CRD_CHECK_ARG(x && (stats || chan_sums), "crd_gn_stats: null pointer");
  int rc = check_common("crd_gn_stats", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk, true);
  if (x_f32) hipLaunchKernelGGL(k_gn_stats<1>, grid, dim3(TPB), lds_rows(C), as_stream(stream), off_ptr(x, x_f32, x_coff),
                                x_f32, x_ld, (long long)P, C, chunk, stats, chan_sums);
  else hipLaunchKernelGGL(k_gn_stats<0>, grid, dim3(TPB), lds_rows(C), as_stream(stream), off_ptr(x, x_f32, x_coff),
                          x_f32, x_ld, (long long)P, C, chunk, stats, chan_sums);
  CRD_LAUNCH_CHECK("crd_gn_stats");
  return CRD_OK;
}

extern "C" int crd_attn_out_residual_stats(const float* x, const float* u, const float* S, const float* bp, const float* dp,
                                           int32_t B, int32_t N, int32_t C, float* x1, crd_sum_t* stats, crd_stream_t stream) {
  CRD_CHECK_ARG(x && u && S && bp && x1 && stats, "crd_attn_out_residual_stats: null pointer");
  int rc = check_common("crd_attn_out_residual_stats", C, 0, C, 1);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(N, C, B, grid, chunk, true);
  hipLaunchKernelGGL(k_attn_out_residual_stats, grid, dim3(TPB), lds_rows(C), as_stream(stream), x, u, S, bp, dp, (long long)N, C,
                     chunk, x1, stats);
  CRD_LAUNCH_CHECK("crd_attn_out_residual_stats");
  return CRD_OK;
}

extern "C" int crd_gn_apply(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C,
                            const crd_sum_t* stats, int32_t gmul, const float* gamma, const float* beta, int32_t act,
                            const float* mask, void* y, int32_t y_f32, int32_t y_ld, int32_t y_coff, crd_stream_t stream) {
  CRD_CHECK_ARG(x && stats && gamma && beta && y, "crd_gn_apply: null pointer");
  CRD_CHECK_ARG(gmul >= 1 && (C / 16) % gmul == 0, "crd_gn_apply: bad gmul %d for C=%d", gmul, C);
  CRD_CHECK_ARG(y_ld % 8 == 0 && y_coff % 8 == 0, "crd_gn_apply: y_ld/y_coff must be multiples of 8");
  int rc = check_common("crd_gn_apply", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk);
  void* yp = y_f32 ? (void*)(reinterpret_cast<float*>(y) + y_coff) : (void*)(reinterpret_cast<bf16_t*>(y) + y_coff);
#define CRD_GN_APPLY(XF, YF, ACT)                                                                                        \
  hipLaunchKernelGGL((k_gn_apply<XF, YF, ACT>), grid, dim3(TPB), 0, as_stream(stream), off_ptr(x, x_f32, x_coff), x_f32, \
                     x_ld, (long long)P, C, chunk, stats, gmul, gamma, beta, act, mask, yp, y_ld, y_f32, 1.f, nullptr, 0)
  const int key = (x_f32 ? 4 : 0) | (y_f32 ? 2 : 0) | (act ? 1 : 0);
  switch (key) {
    case 0: CRD_GN_APPLY(0, 0, 0); break;  case 1: CRD_GN_APPLY(0, 0, 1); break;
    case 2: CRD_GN_APPLY(0, 1, 0); break;  case 3: CRD_GN_APPLY(0, 1, 1); break;
    case 4: CRD_GN_APPLY(1, 0, 0); break;  case 5: CRD_GN_APPLY(1, 0, 1); break;
    case 6: CRD_GN_APPLY(1, 1, 0); break;  default: CRD_GN_APPLY(1, 1, 1); break;
  }
#undef CRD_GN_APPLY
  CRD_LAUNCH_CHECK("crd_gn_apply");
  return CRD_OK;
}

extern "C" int crd_gn_apply_fp8(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C,
                                const crd_sum_t* stats, int32_t gmul, const float* gamma, const float* beta, int32_t act,
                                const float* mask, void* y_fp8, int32_t y_ld, int32_t y_coff, float y_scale, void* y_bf16,
                                int32_t yb_ld, int32_t yb_coff, crd_stream_t stream) {
  CRD_CHECK_ARG(x && stats && gamma && beta && y_fp8 && y_scale > 0.f, "crd_gn_apply_fp8: null pointer / bad scale");
  CRD_CHECK_ARG(gmul >= 1 && (C / 16) % gmul == 0, "crd_gn_apply_fp8: bad gmul %d for C=%d", gmul, C);
  CRD_CHECK_ARG(y_ld % 8 == 0 && y_coff % 8 == 0 && yb_ld % 8 == 0 && yb_coff % 8 == 0, "crd_gn_apply_fp8: ld/coff must be multiples of 8");
  int rc = check_common("crd_gn_apply_fp8", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk);
  void* yp = reinterpret_cast<unsigned char*>(y_fp8) + y_coff;
  void* y2 = y_bf16 ? (void*)(reinterpret_cast<bf16_t*>(y_bf16) + yb_coff) : nullptr;
#define CRD_GN_APPLY8(XF, YF, ACT)                                                                                       \
  hipLaunchKernelGGL((k_gn_apply<XF, YF, ACT>), grid, dim3(TPB), 0, as_stream(stream), off_ptr(x, x_f32, x_coff), x_f32, \
                     x_ld, (long long)P, C, chunk, stats, gmul, gamma, beta, act, mask, yp, y_ld, 0, 1.f / y_scale, y2, yb_ld)
  switch ((x_f32 ? 4 : 0) | (y2 ? 2 : 0) | (act ? 1 : 0)) {
    case 0: CRD_GN_APPLY8(0, 2, 0); break;  case 1: CRD_GN_APPLY8(0, 2, 1); break;
    case 2: CRD_GN_APPLY8(0, 3, 0); break;  case 3: CRD_GN_APPLY8(0, 3, 1); break;
    case 4: CRD_GN_APPLY8(1, 2, 0); break;  case 5: CRD_GN_APPLY8(1, 2, 1); break;
    case 6: CRD_GN_APPLY8(1, 3, 0); break;  default: CRD_GN_APPLY8(1, 3, 1); break;
  }
#undef CRD_GN_APPLY8
  CRD_LAUNCH_CHECK("crd_gn_apply_fp8");
  return CRD_OK;
}

extern "C" int crd_gn_bwd_reduce(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                                 int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                                 int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                                 crd_sum_t* r, float* scratch, int64_t scratch_capacity, crd_stream_t stream) {
  CRD_CHECK_ARG(x && dy && stats && gamma && beta && r, "crd_gn_bwd_reduce: null pointer");
  CRD_CHECK_ARG(gmul >= 1 && (C / 16) % gmul == 0, "crd_gn_bwd_reduce: bad gmul");
  CRD_CHECK_ARG(dy_ld % 8 == 0 && dy_coff % 8 == 0, "crd_gn_bwd_reduce: dy_ld/dy_coff must be multiples of 8");
  int rc = check_common("crd_gn_bwd_reduce", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  // channel slices (atomics path, C >= 256): the largest S in 8 / 4 / 2 whose slices are >= 64 channels and whole groups
  int S = 1;
  static int slicing = -1;
  if (slicing < 0) slicing = crd_dev_int("CRD_GN_RED_SLICES", 1);      // developer switch (A/B)
  if (slicing && !scratch && C >= 256) {
    const int cpg = 16 * gmul;
    for (int s_ = 8; s_ > 1; s_ >>= 1)
      if (C % s_ == 0 && (C / s_) % cpg == 0 && (C / s_) % 8 == 0 && C / s_ >= 64) { S = s_; break; }
  }
  const int Cs = C / S;
  grid_for(P, Cs, B * S, grid, chunk, true);            // the same number of workgroups: S times the pixels per workgroup
  grid = dim3(grid.x, (unsigned)B, (unsigned)S);
  float* part = (scratch && (long long)B * grid.x * 2 * C <= scratch_capacity && grid.x > 1) ? scratch : nullptr;
#define CRD_GN_RED(XF, DF, ACT)                                                                                              \
  hipLaunchKernelGGL((k_gn_bwd_reduce<XF, DF, ACT>), grid, dim3(TPB), lds_rows(Cs), as_stream(stream),                     \
                     off_ptr(x, x_f32, x_coff), x_f32, x_ld, off_ptr(dy, dy_f32, dy_coff), dy_f32, dy_ld, (long long)P, C, \
                     Cs, chunk, stats, gmul, gamma, beta, act, mask, r, part)
  switch ((x_f32 ? 4 : 0) | (dy_f32 ? 2 : 0) | (act ? 1 : 0)) {
    case 0: CRD_GN_RED(0, 0, 0); break;  case 1: CRD_GN_RED(0, 0, 1); break;
    case 2: CRD_GN_RED(0, 1, 0); break;  case 3: CRD_GN_RED(0, 1, 1); break;
    case 4: CRD_GN_RED(1, 0, 0); break;  case 5: CRD_GN_RED(1, 0, 1); break;
    case 6: CRD_GN_RED(1, 1, 0); break;  default: CRD_GN_RED(1, 1, 1); break;
  }
#undef CRD_GN_RED
  if (part)
    hipLaunchKernelGGL(k_gn_bwd_finalize, dim3(cdiv(C, 64), B), dim3(TPB), 0, as_stream(stream), scratch, (int)grid.x, C, gmul,
                       gamma, r, B);
  CRD_LAUNCH_CHECK("crd_gn_bwd_reduce");
  return CRD_OK;
}

static int gn_bwd_apply_impl(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                             int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                             int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                             const crd_sum_t* r, float* dgamma, float* dbeta, void* dx, int32_t dx_f32, int32_t dx_ld,
                             int32_t dx_coff, int32_t dx_accumulate, void* dx2, int32_t dx2_ld, const float* scale2,
                             void* dx8, int32_t dx8_ld, int32_t dx8_coff, const float* scale8, uint32_t* amax8, crd_stream_t stream) {
  unsigned char* dx8p = dx8 ? reinterpret_cast<unsigned char*>(dx8) + dx8_coff : nullptr;
  CRD_CHECK_ARG(x && dy && stats && gamma && beta && r && dx, "crd_gn_bwd_apply: null pointer");
  CRD_CHECK_ARG(!dx2 || dx2_ld % 8 == 0, "crd_gn_bwd_apply: dx2_ld must be a multiple of 8");
  CRD_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "crd_gn_bwd_apply: dgamma and dbeta go together");
  CRD_CHECK_ARG(gmul >= 1 && (C / 16) % gmul == 0, "crd_gn_bwd_apply: bad gmul");
  CRD_CHECK_ARG(dy_ld % 8 == 0 && dy_coff % 8 == 0 && dx_ld % 8 == 0 && dx_coff % 8 == 0,
                "crd_gn_bwd_apply: dy/dx ld/coff must be multiples of 8");
  int rc = check_common("crd_gn_bwd_apply", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk);
  void* dxp = dx_f32 ? (void*)(reinterpret_cast<float*>(dx) + dx_coff) : (void*)(reinterpret_cast<bf16_t*>(dx) + dx_coff);
#define CRD_GN_BAP(XF, DF, ACT)                                                                                              \
  hipLaunchKernelGGL((k_gn_bwd_apply<XF, DF, ACT>), grid, dim3(TPB), 0, as_stream(stream), off_ptr(x, x_f32, x_coff), x_f32, \
                     x_ld, off_ptr(dy, dy_f32, dy_coff), dy_f32, dy_ld, (long long)P, C, chunk, stats, gmul, gamma, beta,  \
                     act, mask, r, dgamma, dbeta, dxp, dx_f32, dx_ld, dx_accumulate, B, dx2, dx2_ld, scale2, dx8p, dx8_ld, scale8,  \
                     amax8)
  switch ((x_f32 ? 4 : 0) | (dy_f32 ? 2 : 0) | (act ? 1 : 0)) {
    case 0: CRD_GN_BAP(0, 0, 0); break;  case 1: CRD_GN_BAP(0, 0, 1); break;
    case 2: CRD_GN_BAP(0, 1, 0); break;  case 3: CRD_GN_BAP(0, 1, 1); break;
    case 4: CRD_GN_BAP(1, 0, 0); break;  case 5: CRD_GN_BAP(1, 0, 1); break;
    case 6: CRD_GN_BAP(1, 1, 0); break;  default: CRD_GN_BAP(1, 1, 1); break;
  }
#undef CRD_GN_BAP
  CRD_LAUNCH_CHECK("crd_gn_bwd_apply");
  return CRD_OK;
}

extern "C" int crd_gn_bwd_apply(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                                int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                                int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                                const crd_sum_t* r, float* dgamma, float* dbeta, void* dx, int32_t dx_f32, int32_t dx_ld,
                                int32_t dx_coff, int32_t dx_accumulate, void* dx2, int32_t dx2_ld, const float* scale2,
                                crd_stream_t stream) {
  return gn_bwd_apply_impl(x, x_f32, x_ld, x_coff, dy, dy_f32, dy_ld, dy_coff, B, P, C, stats, gmul, gamma, beta, act, mask, r, dgamma, dbeta,
                           dx, dx_f32, dx_ld, dx_coff, dx_accumulate, dx2, dx2_ld, scale2, nullptr, 0, 0, nullptr, nullptr, stream);
}

extern "C" int crd_gn_bwd_apply_fp8(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                                    int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                                    int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                                    const crd_sum_t* r, float* dgamma, float* dbeta, void* dx, int32_t dx_ld, int32_t dx_coff,
                                    void* dx_fp8, int32_t dx8_ld, int32_t dx8_coff, const float* scale_dev, uint32_t* amax_slots,
                                    crd_stream_t stream) {
  CRD_CHECK_ARG(dx_fp8 && scale_dev && amax_slots && dx8_ld % 8 == 0 && dx8_coff % 8 == 0,
                "crd_gn_bwd_apply_fp8: the e4m3 copy needs its buffer, a device scale and the amax slots (ld / coff multiples of 8)");
  return gn_bwd_apply_impl(x, x_f32, x_ld, x_coff, dy, dy_f32, dy_ld, dy_coff, B, P, C, stats, gmul, gamma, beta, act, mask, r, dgamma, dbeta,
                           dx, 0, dx_ld, dx_coff, 0, nullptr, 0, nullptr, dx_fp8, dx8_ld, dx8_coff, scale_dev, amax_slots, stream);
}
