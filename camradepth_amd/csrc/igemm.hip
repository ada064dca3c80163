// Implicit-GEMM convolution on MFMA for gfx950 (bf16 operands, fp32 accumulate).
//
// One kernel family covers every dense convolution / pointwise linear layer of the CamRaDepth hot
// path and their data gradients (see include/camradepth_hip.h: crd_conv_igemm).  GEMM view:
//   M = output pixels of one image (grid.z = image), N = Cout, K = (tap, cin) flattened, where a
//   16-byte granule (8 channels) never straddles a tap because Cin % 8 == 0.
// A-operand rows are gathered on the fly from the pixel-major activation tensor (im2col is never
// materialised); the 3x3 re-reads are absorbed by the XCD L2.  Tiles are staged through LDS with
// an XOR swizzle so the ds_read_b128 fragment reads of v_mfma_f32_32x32x16_bf16 are conflict-free.
#include <stdlib.h>
#include "conv_common.h"

using namespace crdk;
int crd_conv3x3_halo(const ConvK& k, int B, hipStream_t st, long long partial_cap);   // conv3x3.hip
bool crd_pw_wide_plain_applicable(const ConvK& k);                                      // gngemm.hip: wide pointwise layers
int crd_pw_wide_plain(const ConvK& k, int B, hipStream_t st);

namespace crdk {
__global__ __launch_bounds__(256) void k_stats_finalize(const float* partial, int n_tiles, int G16, crd_sum_t* stats) {
  // one workgroup per (group, sample); fixed summation order (thread-strided rows four at a time, the wave butterfly, then the
  // four waves in order): reproducible.  (64 threads walking the rows one dependent load at a time took 26 us for the 5200
  // rows of a 416 x 800 frame.)
  __shared__ float sw[4][2];
  const int g = blockIdx.x, b = blockIdx.y;
  const float* base = partial + ((long long)b * n_tiles * G16 + g) * 2;
  const long long rs = (long long)G16 * 2;
  float s = 0.f, ss = 0.f;
  int t = threadIdx.x;
  for (; t + 768 < n_tiles; t += 1024) {
    const float2 v0 = *reinterpret_cast<const float2*>(base + t * rs), v1 = *reinterpret_cast<const float2*>(base + (t + 256) * rs);
    const float2 v2 = *reinterpret_cast<const float2*>(base + (t + 512) * rs), v3 = *reinterpret_cast<const float2*>(base + (t + 768) * rs);
    s += (v0.x + v1.x) + (v2.x + v3.x); ss += (v0.y + v1.y) + (v2.y + v3.y);
  }
  for (; t < n_tiles; t += 256) { const float2 v = *reinterpret_cast<const float2*>(base + t * rs); s += v.x; ss += v.y; }
  s = wave_sum(s); ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) { sw[threadIdx.x >> 6][0] = s; sw[threadIdx.x >> 6][1] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = (sw[0][0] + sw[1][0]) + (sw[2][0] + sw[3][0]), q = (sw[0][1] + sw[1][1]) + (sw[2][1] + sw[3][1]);
    stats[((long long)b * G16 + g) * 2] += to_fx(a, STAT_ONE); stats[((long long)b * G16 + g) * 2 + 1] += to_fx(q, STAT_ONE);
  }
}
}  // namespace crdk

namespace {

constexpr int BK = 64;                 // K elements per LDS stage (8 granules of 8 bf16 per row)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// MODE 0: forward gather (any stride); 1: data-gradient gather, stride 1; 2: data-gradient gather, strided
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NST = LDS stages.  A K-slab of a 64x64 tile is only four MFMAs per wave, so with one slab in flight every K-step of
// the small encoder GEMMs costs a full memory latency (12 us for K = 640); NST - 1 slabs in flight hide it.
// KG > 1: intra-workgroup split-K.  KG groups of 4 waves each walk every KG-th K-slab with their own LDS stages and
// accumulators; the partial tiles are added through LDS and group 0 runs the epilogue.  For deep K on a grid too small
// to cover the CUs (the k = sr patch convs: 32 workgroups x 32..64 slabs, fc2 of the small stages) the K loop is bound
// by the issue cost of the LDS-DMA pieces (~0.45 us per slab whatever the stage count): more waves issue them in
// parallel.
// REGE: the register epilogue (conv_common.h: conv_epilogue_reg) -- MFMA operands swapped, stores straight from the accumulators, the
// epilogue's own inputs prefetched at kernel start.  64 x 64 tiles without split-K only; chosen per launch (launch_mode).
template <int WM, int WN, int TM, int TN, int MODE, int NST, int KG = 1, bool REGE = false>
__global__ __launch_bounds__(256 * KG) void k_igemm(ConvK a) {
  static_assert(WM * WN == 4, "4 waves per group");
  static_assert(!REGE || (TM == 1 && TN == 1 && KG == 1), "register epilogue: one 32 x 32 tile per wave, no split-K");
  static_assert(KG == 1 || (TM == 1 && TN == 1), "split-K groups exchange a single 32x32 tile per wave");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;   // each wave DMAs 8 rows per instruction, 4 waves -> 32 rows per pass
  constexpr int PER = A_IT + B_IT;                // DMA instructions per thread and stage
  static_assert((NST - 2) * PER < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];     // KG * NST * (BM + BN) * BK
  const int grp = KG > 1 ? (int)(threadIdx.x >> 8) : 0;
  bf16_t* sA = lds + grp * NST * (BM + BN) * BK;
  bf16_t* sB = sA + NST * BM * BK;

  const int t = threadIdx.x & 255, l = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int b = blockIdx.z, m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (a.dbg & 8) return;
  RegEpiState est;
  if constexpr (REGE) reg_epi_prefetch<WM, WN>(a, b, l, wm, wn, m0, n0, est);
  // hardware-bounds-checked buffer loads: an out-of-range offset returns zeros, which implements the conv padding,
  // the K tail and the partial M/N tiles without a single branch in the load path
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + (long long)b * a.x_bstride), 0, (int)(a.x_bstride * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const unsigned OOB = 0x80000000u;
  // LDS-DMA staging (buffer_load ... lds): one wave instruction fills 8 consecutive tile rows (8 x 128 B, lane-linear:
  // lane -> row l>>3, 16-byte slot l&7).  The XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is
  // applied on the SOURCE side: slot s of row r receives K-granule s ^ ((r>>1)&7).  A thread's rows are 32 apart, so
  // its swizzle -- and therefore its K-granule and (tap, channel) walk -- is the same for all of them.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int r0 = 8 * wv + (l >> 3);
  const int g = (l & 7) ^ ((r0 >> 1) & 7);

  int py[A_IT], px[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + r0 + 32 * i;
    const int oy = m / a.OW, ox = m - oy * a.OW;
    if (MODE == 0) { py[i] = oy * a.stride - a.pad; px[i] = ox * a.stride - a.pad; }
    else { py[i] = oy + a.pad; px[i] = ox + a.pad; }
    if (m >= a.OHW) py[i] = -(1 << 28);      // rows past the image: every tap lands out of range
  }
  unsigned woff[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int n = r0 + 32 * j, ng = n0 + n;
    woff[j] = (n < BN && ng < a.Cout) ? (unsigned)(ng * a.Ktot * 2) : OOB;
  }
  int kf = g * 8 + grp * BK, kc = kf, ky = 0, kx = 0;
  while (kc >= a.Cin) { kc -= a.Cin; if (++kx == a.KW) { kx = 0; ++ky; } }

  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto stage = [&](int buf) {
    const bool kok = kf < a.Ktot;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int iy, ix;
      bool ok = kok;
      if (MODE == 0) { iy = py[i] + ky; ix = px[i] + kx; }
      else if (MODE == 1) { iy = py[i] - ky; ix = px[i] - kx; }
      else {
        const int ty = py[i] - ky, tx = px[i] - kx;
        iy = ty / a.stride; ix = tx / a.stride;
        ok = ok && ty >= 0 && tx >= 0 && iy * a.stride == ty && ix * a.stride == tx;
      }
      ok = ok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const unsigned off = ok ? (unsigned)(((iy * a.IW + ix) * a.x_ld + kc) * 2) : OOB;
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of hipcc only needs the kernel's signature)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(sA + buf * BM * BK + (8 * wv + 32 * i) * BK), 16, off, 0, 0, 0);
#else
      (void)off;
#endif
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      if (8 * wv + 32 * j < BN) {          // wave-uniform
        const unsigned off = kok ? woff[j] + (unsigned)(kf * 2) : OOB;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(sB + buf * BN * BK + (8 * wv + 32 * j) * BK), 16,
                                                 off | (woff[j] & OOB), 0, 0, 0);
#else
        (void)off;
#endif
      }
    }
    kf += KG * BK; kc += KG * BK;
    while (kc >= a.Cin) { kc -= a.Cin; if (++kx == a.KW) { kx = 0; ++ky; } }
  };

  // accumulators start at the bias of their column (lane l holds column l&31 of every 32x32 tile): the load overlaps the
  // first DMA instead of adding a dependent memory latency to the epilogue
  f32x16 acc[TM][TN];
  if constexpr (REGE) {        // swapped layout: register r of a lane is channel (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the wave's 32 columns
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      acc[0][0][r] = (a.bias && col < a.Cout) ? a.bias[(long long)b * a.bias_bstride + col] : 0.f;
    }
  } else {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (l & 31);
    const float bias_v = (a.bias && col < a.Cout && grp == 0) ? a.bias[(long long)b * a.bias_bstride + col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bias_v;
  }
  }

  const int nK = ((a.dbg & 16) ? 0 : (a.Ktot + BK - 1) / BK + KG - 1) / KG;     // slabs per group (uniform: the tail is zero fill)
  // Slabs past the K range are still issued (their offsets are out of range: zero fill, no memory traffic), so the
  // number of DMAs in flight behind slab kt is always (NST - 2) * PER and the wait needs no tail cases.
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) stage(s);
  int cur = 0, nxt = NST - 1;
  for (int kt = 0; kt < nK; ++kt) {
    wait_vm<(NST - 2) * PER>();             // this thread's share of slab kt has landed ...
    lds_barrier();                          // ... everyone's has, and everyone is done reading slab kt-1 (NOT __syncthreads: common.h)
    stage(nxt);                             // refill slab kt-1's buffer under this slab's MFMAs
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[TM], bfr[TN];
      const int gi = ks * 2 + (l >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + (l & 31);
        af[i] = *reinterpret_cast<const bf16x8*>(&sA[cur * BM * BK + row * BK + ((gi ^ ((row >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + (l & 31);
        bfr[j] = *reinterpret_cast<const bf16x8*>(&sB[cur * BN * BK + row * BK + ((gi ^ ((row >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (REGE) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C[channel][pixel]
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    cur = cur + 1 == NST ? 0 : cur + 1;
    nxt = nxt + 1 == NST ? 0 : nxt + 1;
  }
  wait_vm<0>();                             // the zero-fill slabs still target the LDS the epilogue reuses
  __syncthreads();
  if (a.dbg & 4) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(a.y)[0] = 1.f; return; }
  if (KG > 1) {    // groups 1.. park their tile in their own (now idle) stage area, group 0 adds them up
    float* park = reinterpret_cast<float*>(sA);
    if (grp > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) park[r * 256 + t] = acc[0][0][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int g2 = 1; g2 < KG; ++g2) {
        const float* src = reinterpret_cast<const float*>(lds + g2 * NST * (BM + BN) * BK);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += src[r * 256 + t];
      }
    } else {
      conv_epilogue_idle<BM, BN, 256>(a);       // same barriers as the epilogue below
      return;
    }
  }

  // ---- epilogue (conv_common.h) ----
  if constexpr (REGE) {
    conv_epilogue_reg<WM, WN>(a, acc[0][0], b, l, wm, wn, n0, blockIdx.x, lds, est);
    return;
  }
  conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, blockIdx.x, lds,
                                [&](int i, int rr, bool& valid, int& row) {
    row = m0 + (wm * TM + i) * 32 + rr;
    valid = row < a.OHW;
  }, [&](int rl, bool& valid, int& row) {
    row = m0 + rl;
    valid = row < a.OHW;
  });
}

int g_rege_on = 1;
long long g_rege_launches = 0;
// bf16 output in the plain layout (what conv_epilogue's vector path covers, minus the patch scatter), no per-tile partial statistics,
// no per-channel sums; the fused reduce needs whole 16-channel slabs
bool crd_igemm_reg_epilogue_applies(const ConvK& k) {
  const bool ok = g_rege_on && k.out_mode == 0 && !k.y_f32 && !k.res && (k.Cout & 7) == 0 && (k.y_ld & 7) == 0 && k.vec_ok && !k.chan &&
                  (!k.red_x || ((k.Cout & 15) == 0 && (k.red_x_ld & 7) == 0 && (reinterpret_cast<uintptr_t>(k.red_x) & 15) == 0));
  if (ok) ++g_rege_launches;
  return ok;
}

template <int WM, int WN, int TM, int TN, int MODE, int NST, int KG = 1>
void launch_mode(const ConvK& k, dim3 grid, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr size_t lds = (size_t)KG * NST * (BM + BN) * BK * sizeof(bf16_t);
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    crd_reserve_lds(reinterpret_cast<const void*>(&k_igemm<WM, WN, TM, TN, MODE, NST, KG>), (int)lds, "k_igemm");
    attr_done = true;
  }
  ConvK kk = k;
  kk.lds_bytes = (int)lds;
  if constexpr (TM == 1 && TN == 1 && KG == 1 && WM == 2 && WN == 2) {
    if (crd_igemm_reg_epilogue_applies(k)) {
      hipLaunchKernelGGL((k_igemm<WM, WN, TM, TN, MODE, NST, KG, true>), grid, dim3(256), lds, st, kk);
      return;
    }
  }
  hipLaunchKernelGGL((k_igemm<WM, WN, TM, TN, MODE, NST, KG>), grid, dim3(256 * KG), lds, st, kk);
}

template <int WM, int WN, int TM, int TN, int NST, int KG = 1>
int launch(const ConvK& k0, int B, hipStream_t st, long long partial_cap) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  ConvK k = k0;
  k.n_tiles = cdiv(k.OHW, BM);
  if ((long long)B * k.n_tiles * k.G16 * 2 > partial_cap) k.stats_partial = nullptr;
  dim3 grid(k.n_tiles, cdiv(k.Cout, BN), B);
  if (k.gather_mode == 0) launch_mode<WM, WN, TM, TN, 0, NST, KG>(k, grid, st);
  else if (k.stride == 1) launch_mode<WM, WN, TM, TN, 1, NST, KG>(k, grid, st);
  else launch_mode<WM, WN, TM, TN, 2, NST, KG>(k, grid, st);
  if (k.stats && k.stats_partial)
    hipLaunchKernelGGL(k_stats_finalize, dim3(k.G16, B), dim3(256), 0, st, k.stats_partial, k.n_tiles, k.G16, k.stats);
  CRD_LAUNCH_CHECK("crd_conv_igemm");
  return CRD_OK;
}

}  // namespace

extern "C" int crd_conv_igemm(const crd_conv_desc* d, crd_stream_t stream) {
  CRD_CHECK_ARG(d && d->x && d->w && d->y, "crd_conv_igemm: null pointer");
  CRD_CHECK_ARG(d->Cin % 8 == 0 && d->x_ld % 8 == 0 && d->x_coff % 8 == 0,
                "crd_conv_igemm: Cin/x_ld/x_coff must be multiples of 8 (got %d/%d/%d)", d->Cin, d->x_ld, d->x_coff);
  CRD_CHECK_ARG(d->B > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0,
                "crd_conv_igemm: bad dims");
  CRD_CHECK_ARG(!(d->res && !d->y_f32), "crd_conv_igemm: residual epilogue needs fp32 output");
  CRD_CHECK_ARG(!d->stats || d->Cout % 16 == 0, "crd_conv_igemm: stats need Cout %% 16 == 0");
  CRD_CHECK_ARG(d->out_mode == 0 || (d->patch_k > 0 && d->patch_c > 0 && d->Cout == d->patch_k * d->patch_k * d->patch_c),
                "crd_conv_igemm: bad patch-scatter dims");
  CRD_UNSUPPORTED((long long)d->IH * d->IW * d->x_ld < (1ll << 30) && (long long)d->Cout * d->KH * d->KW * d->Cin < (1ll << 30),
                  "crd_conv_igemm: image or weight tensor too large for 32-bit byte offsets");
  ConvK k;
  k.x = reinterpret_cast<const bf16_t*>(d->x) + d->x_coff; k.x_ld = d->x_ld;
  k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin; k.x_bstride = (long long)d->IH * d->IW * d->x_ld;
  k.w = reinterpret_cast<const bf16_t*>(d->w);
  k.Cout = d->Cout; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.Ktot = d->KH * d->KW * d->Cin;
  k.OW = d->OW; k.OHW = d->OH * d->OW; k.gather_mode = d->gather_mode;
  k.y_ld = d->y_ld; k.y_f32 = d->y_f32;
  k.out_mode = d->out_mode; k.patch_k = d->patch_k; k.patch_c = d->patch_c;
  int YH = d->OH, YW = d->OW;
  if (d->out_mode == 1) { YH = d->OH * d->patch_k; YW = d->OW * d->patch_k; }
  k.YW = YW;
  k.y_bstride = (long long)YH * YW * d->y_ld;
  k.y = d->y_f32 ? (void*)(reinterpret_cast<float*>(d->y) + d->y_coff) : (void*)(reinterpret_cast<bf16_t*>(d->y) + d->y_coff);
  k.bias = d->bias; k.bias_bstride = d->bias_bstride; k.act = d->act;
  k.res = d->res; k.res_ld = d->res_ld; k.res_bstride = (long long)YH * YW * d->res_ld; k.res_scale = d->res_scale;
  k.accumulate = d->accumulate; k.stats = d->stats; k.G16 = d->Cout / 16;
  k.stats_partial = d->stats ? d->stats_partial : nullptr; k.n_tiles = 0; k.col0 = 0;
  k.chan = d->chan_sums;
  CRD_UNSUPPORTED(!d->chan_sums || (d->stats && (d->y_f32 || d->res || d->out_mode != 0)),
                  "crd_conv_igemm: chan_sums needs stats and an fp32 / residual output (the scalar epilogue)");
  k.vec_ok = (d->y_coff % 8 == 0) && ((reinterpret_cast<uintptr_t>(d->y) & 15) == 0);
  k.vecf_ok = d->y_f32 && d->y_coff % 4 == 0 && d->y_ld % 4 == 0 && d->Cout % 4 == 0 && (reinterpret_cast<uintptr_t>(d->y) & 15) == 0 &&
              (!d->res || (d->res_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(d->res) & 15) == 0));
  k.lds_bytes = 0;
  k.red_x = d->red_x; k.red_x_f32 = d->red_x_f32; k.red_x_ld = d->red_x_ld;
  k.red_x_bstride = (long long)YH * YW * d->red_x_ld;
  k.red_stats = d->red_stats; k.red_gamma = d->red_gamma; k.red_beta = d->red_beta; k.red_gmul = d->red_gmul;
  k.red_act = d->red_act; k.red_r = d->red_r;
  const long long pcap = d->stats_partial ? d->stats_partial_capacity : 0;
  { static int dbg = -1; if (dbg < 0) dbg = crd_dev_int("CRD_DBG", 0); k.dbg = dbg; }
  hipStream_t st = as_stream(stream);
  if (d->red_x) {
    const bool halo_shape = d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->IW >= 32 && d->IH >= 8;
    CRD_CHECK_ARG(d->red_stats && d->red_gamma && d->red_beta && d->red_r && d->red_gmul >= 1 && d->red_x_ld % 8 == 0 &&
                  (d->Cout / 16) % d->red_gmul == 0, "crd_conv_igemm: incomplete fused-reduce arguments");
    // the fused reduce lives in the vector epilogue and needs a tile whose threads keep their columns (256 % (BN/8) == 0):
    // the 64 / 32 / 128-column tiles, not the 96- and 160-column ones
    const bool small_path = d->Cout > 32 && (long long)cdiv(k.OHW, 128) * cdiv(d->Cout, 128) * d->B < 192;
    const bool tile_ok = small_path || d->Cout <= 64 || (d->Cout > 96 && d->Cout <= 128) || d->Cout > 160;
    CRD_UNSUPPORTED(tile_ok && d->Cout % 16 == 0 && !halo_shape && d->out_mode == 0 && !d->y_f32 && !d->res && d->y_ld % 8 == 0 &&
                    k.vec_ok, "crd_conv_igemm: the fused GroupNorm-backward reduce needs a bf16 vector-path output and a 32/64/128-column tile");
  }
  // 3x3 / stride 1 / pad 1 on grids at least one tile wide: halo-tile kernel (conv3x3.hip)
  if (d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->out_mode == 0 && d->IH == d->OH && d->IW == d->OW &&
      d->IW >= 32 && d->IH >= 8)
    return crd_conv3x3_halo(k, d->B, st, pcap);
  // pointwise layers that are all INPUT (Mlp.fc2 and fc1's data gradient at encoder stages 1-2): weights in registers, rows streamed once
  if (d->KH == 1 && d->KW == 1 && crd_pw_narrow_applicable(k, nullptr)) return crd_pw_narrow(k, nullptr, d->B, st);
  // pointwise layers that are all output (fc2's data gradient at encoder stages 1-2): the register-resident-weight kernel
  if (d->KH == 1 && d->KW == 1 && crd_pw_wide_plain_applicable(k)) return crd_pw_wide_plain(k, d->B, st);
  {   // developer override of the tile choice below (tools/bench_small_gemm.py sweeps it)
    static int force = -1;
    if (force < 0) force = crd_dev_int("CRD_IGEMM_FORCE", 0);
    switch (force) {
      case 1: return launch<2, 2, 1, 1, 4>(k, d->B, st, pcap);
      case 2: return launch<2, 2, 2, 1, 3>(k, d->B, st, pcap);
      case 3: return launch<2, 2, 2, 2, 2>(k, d->B, st, pcap);
      case 4: return launch<2, 2, 1, 1, 2, 4>(k, d->B, st, pcap);
      case 5: return launch<2, 2, 1, 1, 3>(k, d->B, st, pcap);
      case 6: return launch<2, 2, 1, 1, 2>(k, d->B, st, pcap);
      default: break;
    }
  }
  // small problems: 64x64 tiles so that the launch still covers the 256 CUs
  {
    const long long big_tiles = (long long)cdiv(k.OHW, 128) * cdiv(d->Cout, 128) * d->B;
    if (d->Cout > 32 && big_tiles < 192) {
      // stages follow the K depth: a stage is 16 KB of LDS, and with one or two K-slabs occupancy (workgroups whose
      // prologue / epilogue latencies overlap) is worth more than prefetch depth
      const int nK = cdiv(k.Ktot, BK);
      if (nK <= 2) return launch<2, 2, 1, 1, 2>(k, d->B, st, pcap);
      if (nK <= 4) return launch<2, 2, 1, 1, 3>(k, d->B, st, pcap);
      // deep K on a grid that cannot even cover the CUs (the spatial-reduction convs: 32 workgroups x 32..64 K-slabs)
      const long long small_tiles = (long long)cdiv(k.OHW, 64) * cdiv(d->Cout, 64) * d->B;
      if (nK >= 8 && small_tiles <= 256) return launch<2, 2, 1, 1, 2, 4>(k, d->B, st, pcap);   // 4-way split-K inside the workgroup (2 x 4 stages measured slower)
      return launch<2, 2, 1, 1, 4>(k, d->B, st, pcap);
    }
  }
  if (d->Cout <= 32) return launch<4, 1, 1, 1, 3>(k, d->B, st, pcap);
  if (d->Cout <= 64) return launch<2, 2, 2, 1, 3>(k, d->B, st, pcap);
  if (d->Cout <= 96) return launch<4, 1, 1, 3, 2>(k, d->B, st, pcap);
  if (d->Cout > 128 && d->Cout <= 160) return launch<4, 1, 1, 5, 2>(k, d->B, st, pcap);
  return launch<2, 2, 2, 2, 2>(k, d->B, st, pcap);
}

extern "C" int crd_tune_igemm_reg_epilogue(int32_t on) {
  if (on < 0) return (int)(g_rege_launches & 0x7fffffff);
  const int prev = g_rege_on;
  g_rege_on = on ? 1 : 0;
  return prev;
}
