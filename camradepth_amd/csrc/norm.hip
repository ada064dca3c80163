// GroupNorm family for gfx950: statistics, apply(+GELU,+dropout mask), backward.
// All kernels are HBM-bound streaming passes: one thread owns 8 consecutive channels (16-byte bf16
// / 32-byte fp32 accesses) of a pixel and keeps its channel group fixed while striding over pixels,
// so per-channel partial sums stay in registers; workgroup partials are merged in LDS and flushed
// with one global atomic per value.
#include "common.h"

namespace {

constexpr int TPB = 256;

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

__global__ __launch_bounds__(TPB) void k_gn_stats(const void* x, int x_f32, int x_ld, long long P, int C, int chunk,
                                                  float* stats, float* chan) {
  extern __shared__ float sm[];  // [C][2]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * C; i += TPB) sm[i] = 0.f;
  __syncthreads();
  Map m(C);
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
  if (m.active) {
    long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    for (long long p = p0 + m.pl; p < p1; p += m.PL) {
      float v[8];
      load8(x, ((long long)b * P + p) * x_ld + m.cg * 8, x_f32, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&sm[(m.cg * 8 + j) * 2], s[j]);
      atomicAdd(&sm[(m.cg * 8 + j) * 2 + 1], ss[j]);
    }
  }
  __syncthreads();
  if (chan)
    for (int i = threadIdx.x; i < 2 * C; i += TPB) atomicAdd(&chan[(long long)b * C * 2 + i], sm[i]);
  if (stats)
    for (int g = threadIdx.x; g < (C >> 4) * 2; g += TPB) {
      int slab = g >> 1, which = g & 1;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) a += sm[(slab * 16 + j) * 2 + which];
      atomicAdd(&stats[((long long)b * (C >> 4) + slab) * 2 + which], a);
    }
}

__global__ __launch_bounds__(TPB) void k_gn_apply(const void* x, int x_f32, int x_ld, long long P, int C, int chunk,
                                                  const float* stats, int gmul, const float* gamma, const float* beta,
                                                  int act, const float* mask, void* y, int y_ld, int y_f32) {
  const int b = blockIdx.y;
  Map m(C);
  if (!m.active) return;
  const int c0 = m.cg * 8;
  float mean, rstd;
  const int grp = (c0 >> 4) / gmul;
  gn_mean_rstd(stats + (long long)b * (C >> 4) * 2, grp * gmul, gmul, 1.f / ((float)P * 16.f * gmul), mean, rstd);
  float ga[8], be[8], mk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] = gamma[c0 + j] * rstd;
    be[j] = beta[c0 + j] - mean * ga[j];
    mk[j] = mask ? mask[(long long)b * C + c0 + j] : 1.f;
  }
  long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
  if (p1 > P) p1 = P;
  for (long long p = p0 + m.pl; p < p1; p += m.PL) {
    float v[8];
    load8(x, ((long long)b * P + p) * x_ld + c0, x_f32, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float u = v[j] * ga[j] + be[j];
      if (act == 1) u = gelu_exact(u);
      v[j] = u * mk[j];
    }
    if (y_f32) store8_f32(reinterpret_cast<float*>(y), ((long long)b * P + p) * y_ld + c0, v);
    else store8_bf16(y, ((long long)b * P + p) * y_ld + c0, v);
  }
}

// r[b][c] = (sum_p g, sum_p g*xhat), g = dy*mask*act'(u)
__global__ __launch_bounds__(TPB) void k_gn_bwd_reduce(const void* x, int x_f32, int x_ld, const void* dy, int dy_f32,
                                                       int dy_ld, long long P, int C, int chunk, const float* stats,
                                                       int gmul, const float* gamma, const float* beta, int act,
                                                       const float* mask, float* r) {
  extern __shared__ float sm[];  // [C][2]
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * C; i += TPB) sm[i] = 0.f;
  __syncthreads();
  Map m(C);
  if (m.active) {
    const int c0 = m.cg * 8;
    float mean, rstd;
    const int grp = (c0 >> 4) / gmul;
    gn_mean_rstd(stats + (long long)b * (C >> 4) * 2, grp * gmul, gmul, 1.f / ((float)P * 16.f * gmul), mean, rstd);
    float ga[8], be[8], mk[8], s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
      mk[j] = mask ? mask[(long long)b * C + c0 + j] : 1.f;
      s0[j] = s1[j] = 0.f;
    }
    long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    for (long long p = p0 + m.pl; p < p1; p += m.PL) {
      float v[8], d[8];
      load8(x, ((long long)b * P + p) * x_ld + c0, x_f32, v);
      load8(dy, ((long long)b * P + p) * dy_ld + c0, dy_f32, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float xh = (v[j] - mean) * rstd;
        float g = d[j] * mk[j];
        if (act == 1) g *= gelu_grad(xh * ga[j] + be[j]);
        s0[j] += g; s1[j] += g * xh;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&sm[(c0 + j) * 2], s0[j]);
      atomicAdd(&sm[(c0 + j) * 2 + 1], s1[j]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += TPB) atomicAdd(&r[(long long)b * C * 2 + i], sm[i]);
  // per-group sums S1 = sum_c gamma_c r0, S2 = sum_c gamma_c r1 (stored after the [B][C][2] block of r)
  const int cpg = 16 * gmul, G = C / cpg;
  float* rg = r + (long long)gridDim.y * C * 2 + (long long)b * G * 2;
  for (int gi = threadIdx.x; gi < 2 * G; gi += TPB) {
    const int grp = gi >> 1, which = gi & 1;
    float acc = 0.f;
    for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) acc += gamma[c] * sm[c * 2 + which];
    atomicAdd(&rg[grp * 2 + which], acc);
  }
}

__global__ __launch_bounds__(TPB) void k_gn_bwd_apply(const void* x, int x_f32, int x_ld, const void* dy, int dy_f32,
                                                      int dy_ld, long long P, int C, int chunk, const float* stats,
                                                      int gmul, const float* gamma, const float* beta, int act,
                                                      const float* mask, const float* r, float* dgamma, float* dbeta,
                                                      void* dx, int dx_f32, int dx_ld, int dx_acc, int B) {
  const int b = blockIdx.y;
  if (blockIdx.x == 0 && b == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += TPB) {
      float g0 = 0.f, g1 = 0.f;
      for (int bb = 0; bb < B; ++bb) { g0 += r[((long long)bb * C + c) * 2]; g1 += r[((long long)bb * C + c) * 2 + 1]; }
      dbeta[c] += g0;
      dgamma[c] += g1;
    }
  }
  Map m(C);
  if (!m.active) return;
  const int c0 = m.cg * 8;
  float mean, rstd;
  const int grp = (c0 >> 4) / gmul;
  const int cpg = 16 * gmul;
  const float inv_m = 1.f / ((float)P * cpg);
  gn_mean_rstd(stats + (long long)b * (C >> 4) * 2, grp * gmul, gmul, inv_m, mean, rstd);
  const float* rg = r + (long long)B * C * 2 + ((long long)b * (C / cpg) + grp) * 2;
  const float S1 = rg[0] * inv_m, S2 = rg[1] * inv_m;
  float ga[8], be[8], mk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
    mk[j] = mask ? mask[(long long)b * C + c0 + j] : 1.f;
  }
  long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
  if (p1 > P) p1 = P;
  for (long long p = p0 + m.pl; p < p1; p += m.PL) {
    float v[8], d[8], o[8];
    load8(x, ((long long)b * P + p) * x_ld + c0, x_f32, v);
    load8(dy, ((long long)b * P + p) * dy_ld + c0, dy_f32, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xh = (v[j] - mean) * rstd;
      float g = d[j] * mk[j];
      if (act == 1) g *= gelu_grad(xh * ga[j] + be[j]);
      o[j] = (ga[j] * g - S1 - xh * S2) * rstd;
    }
    const long long off = ((long long)b * P + p) * dx_ld + c0;
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
  }
}

// reduce=true: kernels that end in per-workgroup global atomics (stats / backward sums) get fewer, longer
// workgroups on large tensors so that the atomic traffic stays negligible
inline void grid_for(long long P, int C, int B, dim3& grid, int& chunk, bool reduce = false) {
  int CG = C >> 3;
  int PL = TPB / CG; if (PL < 1) PL = 1;
  long long per_block = (long long)PL * 4;         // >= 4 pixels per pixel-lane: short dependent-load chains
  long long nblk = (P + per_block - 1) / per_block;
  long long cap = (reduce ? 1024 : 4096) / (B > 0 ? B : 1); if (cap < 1) cap = 1;
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

inline const void* off_ptr(const void* p, int f32, int coff) {
  return f32 ? (const void*)(reinterpret_cast<const float*>(p) + coff) : (const void*)(reinterpret_cast<const bf16_t*>(p) + coff);
}

}  // namespace

extern "C" int crd_gn_stats(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C,
                            float* stats, float* chan_sums, crd_stream_t stream) {
  CRD_CHECK_ARG(x && (stats || chan_sums), "crd_gn_stats: null pointer");
  int rc = check_common("crd_gn_stats", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk, true);
  hipLaunchKernelGGL(k_gn_stats, grid, dim3(TPB), 2 * C * sizeof(float), as_stream(stream), off_ptr(x, x_f32, x_coff),
                     x_f32, x_ld, (long long)P, C, chunk, stats, chan_sums);
  CRD_LAUNCH_CHECK("crd_gn_stats");
  return CRD_OK;
}

extern "C" int crd_gn_apply(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C,
                            const float* stats, int32_t gmul, const float* gamma, const float* beta, int32_t act,
                            const float* mask, void* y, int32_t y_f32, int32_t y_ld, int32_t y_coff, crd_stream_t stream) {
  CRD_CHECK_ARG(x && stats && gamma && beta && y, "crd_gn_apply: null pointer");
  CRD_CHECK_ARG(gmul >= 1 && (C / 16) % gmul == 0, "crd_gn_apply: bad gmul %d for C=%d", gmul, C);
  CRD_CHECK_ARG(y_ld % 8 == 0 && y_coff % 8 == 0, "crd_gn_apply: y_ld/y_coff must be multiples of 8");
  int rc = check_common("crd_gn_apply", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk);
  hipLaunchKernelGGL(k_gn_apply, grid, dim3(TPB), 0, as_stream(stream), off_ptr(x, x_f32, x_coff), x_f32, x_ld,
                     (long long)P, C, chunk, stats, gmul, gamma, beta, act, mask,
                     y_f32 ? (void*)(reinterpret_cast<float*>(y) + y_coff) : (void*)(reinterpret_cast<bf16_t*>(y) + y_coff), y_ld, y_f32);
  CRD_LAUNCH_CHECK("crd_gn_apply");
  return CRD_OK;
}

extern "C" int crd_gn_bwd_reduce(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                                 int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const float* stats,
                                 int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                                 float* r, crd_stream_t stream) {
  CRD_CHECK_ARG(x && dy && stats && gamma && beta && r, "crd_gn_bwd_reduce: null pointer");
  CRD_CHECK_ARG(gmul >= 1 && (C / 16) % gmul == 0, "crd_gn_bwd_reduce: bad gmul");
  CRD_CHECK_ARG(dy_ld % 8 == 0 && dy_coff % 8 == 0, "crd_gn_bwd_reduce: dy_ld/dy_coff must be multiples of 8");
  int rc = check_common("crd_gn_bwd_reduce", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk, true);
  hipLaunchKernelGGL(k_gn_bwd_reduce, grid, dim3(TPB), 2 * C * sizeof(float), as_stream(stream),
                     off_ptr(x, x_f32, x_coff), x_f32, x_ld, off_ptr(dy, dy_f32, dy_coff), dy_f32, dy_ld, (long long)P, C,
                     chunk, stats, gmul, gamma, beta, act, mask, r);
  CRD_LAUNCH_CHECK("crd_gn_bwd_reduce");
  return CRD_OK;
}

extern "C" int crd_gn_bwd_apply(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                                int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const float* stats,
                                int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                                const float* r, float* dgamma, float* dbeta, void* dx, int32_t dx_f32, int32_t dx_ld,
                                int32_t dx_coff, int32_t dx_accumulate, crd_stream_t stream) {
  CRD_CHECK_ARG(x && dy && stats && gamma && beta && r && dx, "crd_gn_bwd_apply: null pointer");
  CRD_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "crd_gn_bwd_apply: dgamma and dbeta go together");
  CRD_CHECK_ARG(gmul >= 1 && (C / 16) % gmul == 0, "crd_gn_bwd_apply: bad gmul");
  CRD_CHECK_ARG(dy_ld % 8 == 0 && dy_coff % 8 == 0 && dx_ld % 8 == 0 && dx_coff % 8 == 0,
                "crd_gn_bwd_apply: dy/dx ld/coff must be multiples of 8");
  int rc = check_common("crd_gn_bwd_apply", x_ld, x_coff, C, x_f32);
  if (rc) return rc;
  dim3 grid; int chunk;
  grid_for(P, C, B, grid, chunk);
  void* dxp = dx_f32 ? (void*)(reinterpret_cast<float*>(dx) + dx_coff) : (void*)(reinterpret_cast<bf16_t*>(dx) + dx_coff);
  hipLaunchKernelGGL(k_gn_bwd_apply, grid, dim3(TPB), 0, as_stream(stream), off_ptr(x, x_f32, x_coff), x_f32, x_ld,
                     off_ptr(dy, dy_f32, dy_coff), dy_f32, dy_ld, (long long)P, C, chunk, stats, gmul, gamma, beta, act,
                     mask, r, dgamma, dbeta, dxp, dx_f32, dx_ld, dx_accumulate, B);
  CRD_LAUNCH_CHECK("crd_gn_bwd_apply");
  return CRD_OK;
}
