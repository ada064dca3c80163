// The NARROW pointwise layers of the encoder's Mlp at stages 1-2 (round 5): Mlp.fc2 (hidden 512 / 1024 -> 64 / 128 channels,
// reference: src/models/simplified_attention.py:20,41) and the data gradient of Mlp.fc1 (the same shape, :17,35 under autograd).
// They are all INPUT: 54.5 / 27 MB of hidden tensor read for 6.8 / 3.4 MB written (B = 8).  As 64 x 64 k_igemm tiles they took 16-18 us
// (the hidden tensor crosses the L2 twice -- once per column tile -- and every workgroup re-streams both operands slab by slab
// through LDS-DMA: 106 MB of L2 -> LDS traffic for 27 MB of data); the byte floor is 4.3 / 8.7 us.
//
// Structure (the mirror image of k_gn_pw_wide, gngemm.hip):
//   * a workgroup owns ALL N output columns, one wave per 16 of them: the wave's weights -- 16 columns x all of K -- sit in its
//     registers as A operands of v_mfma_f32_16x16x32_bf16 (K = 1024: 128 VGPRs), loaded once per workgroup;
//   * the rows stream: 32-row tiles, global -> registers (16 bytes per lane, whole rows: coalesced) one tile ahead -> LDS
//     (row stride K + 8: conflict-free ds_read_b128) -> B operands; the hidden tensor is read from HBM exactly once;
//   * XF = 1: Mlp.norm2 + the exact GELU are applied to the rows in registers on their way into LDS (a thread keeps the same 8
//     channels for every row it moves: its scale / shift live in registers), and the activated tensor H3 is written from the same
//     registers when fc2's weight gradient needs it -- crd_gn_apply's pass over the hidden tensor disappears;
//   * the 32 x N result goes through a small LDS strip so that threads store whole rows; the epilogues are conv_common.h's
//     arithmetic: EPI 1 = fp32 y = res + scale[b] * bf16(acc + bias) with the GroupNorm / per-channel sums of what was stored
//     (fc2 in front of the residual stream), EPI 0 = bf16 y (+ the reduce phase of the following GroupNorm's backward on an fp32
//     red_x: Block.norm2 behind fc1's data gradient).  Sums stay in registers across a workgroup's tiles (one sample per
//     workgroup) and leave as one fixed-point atomic per value at the end.
#include "conv_common.h"

using namespace crdk;

namespace {

constexpr int RT = 32;                       // rows per tile
typedef __attribute__((ext_vector_type(4))) unsigned u32x4n;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2n;
typedef __attribute__((ext_vector_type(4))) float f32x4n;

struct NarrowArgs {
  ConvK a;
  NarrowGn gn;
  int tiles, streams;                        // 32-row tiles per sample; workgroups per sample (workgroup s walks tiles s, s + streams, ..)
};

template <int K, int N, int EPI>
constexpr int stage_bytes() { return EPI == 1 ? RT * (N + 4) * 4 : RT * (N + 8) * 2; }
template <int K, int N, int EPI>
constexpr int lds_bytes() { return 2 * RT * (K + 8) * 2 + stage_bytes<K, N, EPI>(); }

template <int KS, int NWV, int XF, int EPI>
__global__ __launch_bounds__(NWV * 64) void k_pw_narrow(NarrowArgs A) {
  constexpr int K = KS * 32, N = NWV * 16, NT = NWV * 64;
  constexpr int LDX = K + 8;                 // bf16 per LDS row: + 16 bytes, so the 16 pixel rows of a fragment read cover the banks evenly
  constexpr int CPR = K / 8;                 // 16-byte chunks per row
  constexpr int RPP = NT / CPR;              // rows per load pass
  constexpr int NLD = RT / RPP;              // loads per thread and tile
  static_assert(NT % CPR == 0 && RT % RPP == 0 && NLD >= 1, "row tiling");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  bf16_t* sX = reinterpret_cast<bf16_t*>(lds);                          // [2][RT][LDX]
  char* sT = lds + 2 * RT * LDX * 2;                                    // staging strip
  const ConvK& a = A.a;
  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.x / A.streams, s0 = blockIdx.x - b * A.streams;
  const int nT = A.tiles, R = A.streams;

  // ---- this wave's weights: columns 16 wv .. + 15, all of K (A operand: lane = column l & 15, k = 8 (l >> 4) .. + 7 of every 32) ----
  bf16x8 wf[KS];
  {
    const bf16_t* wrow = a.w + (long long)(wv * 16 + (l & 15)) * K + 8 * (l >> 4);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
  }
  const int lr = t / CPR, lc = t - lr * CPR;           // row within a pass, 16-byte chunk = channels 8 lc .. + 7
  const bf16_t* xb = a.x + (long long)b * a.x_bstride + lc * 8;
  u32x4n xr[NLD];
  auto load_tile = [&](int tile) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      xr[i] = *reinterpret_cast<const u32x4n*>(__builtin_assume_aligned(xb + (long long)(tile * RT + i * RPP + lr) * a.x_ld, 16));
  };
  if (s0 < nT) load_tile(s0);
  // GroupNorm (+ GELU) of the rows on their way into LDS: x * sc + sh per channel, this thread's 8 channels
  float sc[XF ? 8 : 1], sh[XF ? 8 : 1];
  bf16_t* xnb = nullptr;
  if constexpr (XF) {
    const int c0 = lc * 8, grp = (c0 >> 4) / A.gn.gmul;
    float mean, rstd;
    gn_mean_rstd(A.gn.stats + (long long)b * (K >> 4) * 2, grp * A.gn.gmul, A.gn.gmul, A.gn.count, mean, rstd);
    // (parameters are views into the model's flat fp32 buffer: 4-byte aligned only -- scalar loads, once per thread)
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = A.gn.gamma[c0 + j] * rstd; sh[j] = A.gn.beta[c0 + j] - mean * sc[j]; }
    if (A.gn.xn) xnb = A.gn.xn + (long long)b * A.gn.xn_bstride + c0;
  }
  auto store_tile = [&](int tile, int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      u32x4n u = xr[i];
      const int row = i * RPP + lr;
      if constexpr (XF) {
        float f[8] = {bf_lo(u[0]), bf_hi(u[0]), bf_lo(u[1]), bf_hi(u[1]), bf_lo(u[2]), bf_hi(u[2]), bf_lo(u[3]), bf_hi(u[3])};
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = gelu_exact(f[j] * sc[j] + sh[j]);
        u[0] = pack_bf2(f[0], f[1]); u[1] = pack_bf2(f[2], f[3]); u[2] = pack_bf2(f[4], f[5]); u[3] = pack_bf2(f[6], f[7]);
        if (xnb) *reinterpret_cast<u32x4n*>(__builtin_assume_aligned(xnb + (long long)(tile * RT + row) * A.gn.xn_ld, 16)) = u;
      }
      *reinterpret_cast<u32x4n*>(__builtin_assume_aligned(sX + (buf * RT + row) * LDX + lc * 8, 16)) = u;
    }
  };
  if (s0 >= nT) return;
  store_tile(s0, 0);
  if (s0 + R < nT) load_tile(s0 + R);
  lds_barrier();

  // ---- per-thread epilogue state ----
  // the bias of this lane's 4 output channels (C layout of 16x16x32: lane = pixel l & 15, rows 4 (l >> 4) + r = channels)
  f32x4n bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
    const float* bp = a.bias + (long long)b * a.bias_bstride + wv * 16 + 4 * (l >> 4);
    bias4 = f32x4n{bp[0], bp[1], bp[2], bp[3]};
  }
  // EPI 1: thread -> 4 columns (float4) of rows t / (N / 4) + k NT / (N / 4); EPI 0: 8 columns (16 bytes of bf16) of row t / (N / 8)
  constexpr int GPR = EPI == 1 ? N / 4 : N / 8;
  constexpr int ERP = NT / GPR;                         // rows per epilogue pass
  static_assert(RT % ERP == 0, "epilogue passes");
  const int eg = t % GPR, er = t / GPR;
  float cs[EPI == 1 ? 4 : 1], css[EPI == 1 ? 4 : 1];
  float rs0[EPI == 0 ? 8 : 1], rs1[EPI == 0 ? 8 : 1];
  float rmean = 0.f, rrstd = 0.f;
  const bool redo = EPI == 0 && a.red_x != nullptr;
  if constexpr (EPI == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = css[j] = 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) rs0[j] = rs1[j] = 0.f;
    if (redo) {
      const int cpg = 16 * a.red_gmul;
      gn_mean_rstd(a.red_stats + (long long)b * (N >> 4) * 2, ((eg * 8) / cpg) * a.red_gmul, a.red_gmul, (float)a.OHW * cpg, rmean, rrstd);
    }
  }
  const float rscale = (EPI == 1 && a.res && a.res_scale) ? a.res_scale[b] : 1.f;

  int buf = 0;
  for (int tile = s0; tile < nT; tile += R, buf ^= 1) {
    // the epilogue's own inputs (residual rows / the GroupNorm input of the fused reduce) are requested NOW, under the MFMA phase:
    // asked for in the epilogue they cost a memory latency per tile with nothing to hide it (20.2 vs 16.5 us at stage 2)
    f32x4n pre[EPI == 1 ? RT / ERP : 2 * (RT / ERP)];
    if constexpr (EPI == 1) {
      if (a.res) {
        const float* resb = a.res + (long long)b * a.res_bstride;
#pragma unroll
        for (int ps = 0; ps < RT / ERP; ++ps)
          pre[ps] = *reinterpret_cast<const f32x4n*>(__builtin_assume_aligned(resb + ((long long)tile * RT + ps * ERP + er) * a.res_ld + eg * 4, 16));
      }
    } else {
      if (redo && a.red_x_f32) {
        const float* rxb = reinterpret_cast<const float*>(a.red_x) + (long long)b * a.red_x_bstride;
#pragma unroll
        for (int ps = 0; ps < RT / ERP; ++ps) {
          const float* rp = rxb + ((long long)tile * RT + ps * ERP + er) * a.red_x_ld + eg * 8;
          pre[2 * ps] = *reinterpret_cast<const f32x4n*>(__builtin_assume_aligned(rp, 16));
          pre[2 * ps + 1] = *reinterpret_cast<const f32x4n*>(__builtin_assume_aligned(rp + 4, 16));
        }
      }
    }
    // ---- 32 x 16 outputs of this wave: 2 pixel halves x KS MFMAs, the rows as B operands from LDS ----
    f32x4n acc[2] = {bias4, bias4};
    const bf16_t* sx = sX + (buf * RT + (l & 15)) * LDX + 8 * (l >> 4);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(__builtin_assume_aligned(sx + h * 16 * LDX + ks * 32, 16));
        acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], bfr, acc[h], 0, 0, 0);
      }
    }
    // ---- into the staging strip: pixel row h 16 + (l & 15), channels 16 wv + 4 (l >> 4) .. + 3 ----
    if constexpr (EPI == 1) {
      float* T = reinterpret_cast<float*>(sT);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<f32x4n*>(__builtin_assume_aligned(T + (h * 16 + (l & 15)) * (N + 4) + wv * 16 + 4 * (l >> 4), 16)) = acc[h];
    } else {
      bf16_t* T = reinterpret_cast<bf16_t*>(sT);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x2n d;
        d[0] = pack_bf2(acc[h][0], acc[h][1]);
        d[1] = pack_bf2(acc[h][2], acc[h][3]);
        *reinterpret_cast<u32x2n*>(__builtin_assume_aligned(T + (h * 16 + (l & 15)) * (N + 8) + wv * 16 + 4 * (l >> 4), 8)) = d;
      }
    }
    lds_barrier();          // the strip is complete; every wave is done reading tile `buf`
    // ---- whole rows out ----
    if constexpr (EPI == 1) {
      const float* T = reinterpret_cast<const float*>(sT);
      float* yb = reinterpret_cast<float*>(a.y) + (long long)b * a.y_bstride;
      const float* resb = a.res ? a.res + (long long)b * a.res_bstride : nullptr;
#pragma unroll
      for (int ps = 0; ps < RT / ERP; ++ps) {
        const int rl = ps * ERP + er;
        const long long p = (long long)tile * RT + rl;
        f32x4n v = *reinterpret_cast<const f32x4n*>(__builtin_assume_aligned(T + rl * (N + 4) + eg * 4, 16));
        if (resb) {
          const f32x4n q = pre[ps];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = q[j] + rscale * bf_round(v[j]);
        }
        *reinterpret_cast<f32x4n*>(__builtin_assume_aligned(yb + p * a.y_ld + eg * 4, 16)) = v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { cs[j] += v[j]; css[j] += v[j] * v[j]; }
      }
    } else {
      const bf16_t* T = reinterpret_cast<const bf16_t*>(sT);
      bf16_t* yb = reinterpret_cast<bf16_t*>(a.y) + (long long)b * a.y_bstride;
#pragma unroll
      for (int ps = 0; ps < RT / ERP; ++ps) {
        const int rl = ps * ERP + er;
        const long long p = (long long)tile * RT + rl;
        const u32x4n u = *reinterpret_cast<const u32x4n*>(__builtin_assume_aligned(T + rl * (N + 8) + eg * 8, 16));
        *reinterpret_cast<u32x4n*>(__builtin_assume_aligned(yb + p * a.y_ld + eg * 8, 16)) = u;
        if (redo) {
          float xq[8];
          if (a.red_x_f32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xq[j] = pre[2 * ps + (j >> 2)][j & 3];
          } else {
            load8(a.red_x, (long long)b * a.red_x_bstride + p * a.red_x_ld + eg * 8, 0, xq);
          }
          const float dq[8] = {bf_lo(u[0]), bf_hi(u[0]), bf_lo(u[1]), bf_hi(u[1]), bf_lo(u[2]), bf_hi(u[2]), bf_lo(u[3]), bf_hi(u[3])};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = (xq[j] - rmean) * rrstd;
            rs0[j] += dq[j];
            rs1[j] += dq[j] * xh;
          }
        }
      }
    }
    // ---- the next tile into the other buffer, the one after it into flight ----
    if (tile + R < nT) store_tile(tile + R, buf ^ 1);
    if (tile + 2 * R < nT) load_tile(tile + 2 * R);
    lds_barrier();          // tile buf ^ 1 is visible; the strip has been read
  }

  // ---- sums: fold the threads that share a column group through LDS (the row buffers are free), one atomic per value ----
  float* fr = reinterpret_cast<float*>(lds);
  if constexpr (EPI == 1) {
    if (!a.stats) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { fr[t * 8 + j * 2] = cs[j]; fr[t * 8 + j * 2 + 1] = css[j]; }
    __syncthreads();
    float* fc = fr + NT * 8;                                   // [N][2] column sums
    if (t < N * 2) {
      const int c = t >> 1, which = t & 1;
      float v = 0.f;
      for (int m = 0; m < NT / GPR; ++m) v += fr[((c >> 2) + m * GPR) * 8 + (c & 3) * 2 + which];
      fc[t] = v;
      if (a.chan) stat_add(&a.chan[((long long)b * N + c) * 2 + which], v);
    }
    __syncthreads();
    if (t < (N / 16) * 2) {
      const int slab = t >> 1, which = t & 1;
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) v += fc[(slab * 16 + j) * 2 + which];
      stat_add(a.stats + ((long long)b * a.G16 + slab) * 2 + which, v);
    }
  } else {
    if (!redo) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) { fr[t * 16 + j * 2] = rs0[j]; fr[t * 16 + j * 2 + 1] = rs1[j]; }
    __syncthreads();
    float* fw = fr + NT * 16;
    if (t < GPR * 16) {
      const int g = t >> 4, jk = t & 15;
      float v = 0.f;
      for (int m = 0; m < NT / GPR; ++m) v += fr[(g + m * GPR) * 16 + jk];
      const int c = g * 8 + (jk >> 1);
      grad_add(&a.red_r[((long long)b * N + c) * 2 + (jk & 1)], v);
      fw[t] = v * a.red_gamma[c];
    }
    __syncthreads();
    if (t < (N / 16) * 2) {
      const int slab = t >> 1, which = t & 1;
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) s2 += fw[(slab * 16 + j) * 2 + which];
      const int cpg = 16 * a.red_gmul, Bn = (int)gridDim.x / R;
      grad_add(&a.red_r[(long long)Bn * N * 2 + ((long long)b * (N / cpg) + (slab * 16) / cpg) * 2 + which], s2);
    }
  }
}

template <int KS, int NWV, int XF, int EPI>
int launch_narrow(const ConvK& k, const NarrowGn* gn, int B, hipStream_t st) {
  constexpr int K = KS * 32, N = NWV * 16;
  constexpr int lds = lds_bytes<K, N, EPI>();
  static_assert(lds <= 160 * 1024, "LDS");
  static_assert(NWV * 64 * 16 * 4 + N * 2 * 4 + 64 <= 2 * RT * (K + 8) * 2, "the sum folds reuse the row buffers");
  static bool attr_done = false;
  if (!attr_done) { crd_reserve_lds(reinterpret_cast<const void*>(&k_pw_narrow<KS, NWV, XF, EPI>), lds, "k_pw_narrow"); attr_done = true; }
  NarrowArgs A;
  A.a = k;
  if (gn) A.gn = *gn; else { A.gn.stats = nullptr; A.gn.gmul = 1; A.gn.gamma = A.gn.beta = nullptr; A.gn.count = 1.f; A.gn.xn = nullptr; A.gn.xn_ld = 0; A.gn.xn_bstride = 0; }
  A.tiles = k.OHW / RT;
  // workgroups: as many as stay resident (the LDS decides: one per CU at K = 1024, two at K = 512), balanced over a sample's tiles
  const int per_cu = (160 * 1024) / lds;
  int rmax = 256 * (per_cu < 1 ? 1 : per_cu) / B;
  if (rmax < 1) rmax = 1;
  const int tpw = cdiv(A.tiles, rmax);
  A.streams = cdiv(A.tiles, tpw);
  hipLaunchKernelGGL((k_pw_narrow<KS, NWV, XF, EPI>), dim3(B * A.streams), dim3(NWV * 64), lds, st, A);
  CRD_LAUNCH_CHECK("crd_conv_igemm(narrow pointwise)");
  return CRD_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Which launches take this kernel: 1x1 / stride 1, (K, N) = (512, 64) or (1024, 128) -- Mlp.fc2 and fc1's data gradient of encoder
// stages 1-2 --, whole 32-row tiles per sample, and one of the two epilogues it implements.
static int g_narrow_on = 1;
static int g_narrow_k1024_plain = 0;       // crd_tune_pw_narrow(2): also take the K = 1024 launches without a GroupNorm in front (benchmarks)
static long long g_narrow_launches = 0;

bool crd_pw_narrow_applicable(const ConvK& k, const NarrowGn* gn) {
  const int on = g_narrow_on;
  const bool shape = k.KW == 1 && k.stride == 1 && k.pad == 0 && k.Ktot == k.Cin && ((k.Cin == 512 && k.Cout == 64) || (k.Cin == 1024 && k.Cout == 128)) &&
                     k.IH * k.IW == k.OHW && k.OHW % RT == 0 && (k.x_ld & 7) == 0 && aligned16(k.x) && aligned16(k.w) && k.out_mode == 0 &&
                     k.gather_mode != 2 && !k.act && !k.accumulate && !k.stats_partial && k.col0 == 0;
  if (!on || !shape) return false;
  if (gn && !(gn->stats && gn->gamma && gn->beta && (k.Cin / 16) % gn->gmul == 0 &&
              (!gn->xn || (aligned16(gn->xn) && gn->xn_ld % 8 == 0))))
    return false;
  // measured (tools/bench_narrow.py, profiles/r05_narrow_pointwise_microbench.txt): behind GroupNorm + GELU this kernel beats the
  // generic tiles at both stages (36.7 vs 43.0 / 26.2 vs 37.2 us, with H3 stored 42.7 vs 52.7 / 30.6 vs 41.2); on rows that are
  // already activated it wins at K = 512 (14.8-23.9 vs 15.6-25.1 us) and LOSES at K = 1024 (16.5-20.2 vs 14.0-17.2: one workgroup
  // per CU, 256 KB of weights per workgroup for two 32-row tiles) -- those launches stay on k_igemm
  if (!gn && k.Cin == 1024 && !g_narrow_k1024_plain) return false;
  if (k.y_f32)          // fp32 residual-stream output (+ sums)
    return k.vecf_ok && !k.red_x && (k.y_ld & 3) == 0 && (!k.res || (k.res_ld & 3) == 0) && (!k.chan || k.stats) && (!k.stats || k.G16 == k.Cout / 16);
  // bf16 output (+ fused GroupNorm-backward reduce without activation)
  return !k.res && !k.stats && !k.chan && k.vec_ok && (k.y_ld & 7) == 0 &&
         (!k.red_x || (k.red_act == 0 && k.red_stats && k.red_gamma && k.red_r && (k.red_x_ld & 7) == 0 && aligned16(k.red_x) &&
                       (k.Cout / 16) % k.red_gmul == 0));
}

int crd_pw_narrow(const ConvK& k, const NarrowGn* gn, int B, hipStream_t st) {
  const bool big = k.Cin == 1024;
  ++g_narrow_launches;
  if (k.y_f32) {
    if (gn) return big ? launch_narrow<32, 8, 1, 1>(k, gn, B, st) : launch_narrow<16, 4, 1, 1>(k, gn, B, st);
    return big ? launch_narrow<32, 8, 0, 1>(k, gn, B, st) : launch_narrow<16, 4, 0, 1>(k, gn, B, st);
  }
  if (gn) return big ? launch_narrow<32, 8, 1, 0>(k, gn, B, st) : launch_narrow<16, 4, 1, 0>(k, gn, B, st);
  return big ? launch_narrow<32, 8, 0, 0>(k, gn, B, st) : launch_narrow<16, 4, 0, 0>(k, gn, B, st);
}

extern "C" int crd_tune_pw_narrow(int32_t on) {
  if (on < 0) return (int)(g_narrow_launches & 0x7fffffff);
  const int prev = g_narrow_on;
  g_narrow_on = on ? 1 : 0;
  g_narrow_k1024_plain = on == 2 ? 1 : 0;
  return prev;
}

extern "C" int crd_pw_narrow_supported(int32_t Cin, int32_t Cout, int32_t pixels) {
  const int on = g_narrow_on;
  return (on && ((Cin == 512 && Cout == 64) || (Cin == 1024 && Cout == 128)) && pixels > 0 && pixels % RT == 0) ? 1 : 0;
}
