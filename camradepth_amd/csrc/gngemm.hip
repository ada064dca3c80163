// GroupNorm-apply folded into the A-operand path of an MFMA GEMM (gfx950, bf16 operands, fp32 accumulate).
//
//   y = epilogue( W * act( GN(x) ) )        x: raw (un-normalised) fp32 or bf16 pixel-major tensor
//
// for the pointwise / non-overlapping-patch convolutions of the encoder blocks (reference:
// src/models/simplified_attention.py:34-43 fc1 / fc2 behind Mlp.norm2 + GELU, :96-100 attn.k behind attn.norm,
// :142-145 q behind Block.norm1 and fc1 behind Block.norm2).  Each of these was a crd_gn_apply launch followed by a
// crd_conv_igemm launch: an extra pass over the tensor and one more link in the encoder's latency-bound launch chain.
//
// The raw rows arrive in LDS by LDS-DMA exactly like k_igemm's operands (asynchronous, hardware bounds-checked, no
// registers in flight), and a short LDS -> LDS pass turns a landed slab into the MFMA operand:
//   x * scale[c] + shift[c]  (per sample and channel, from the GroupNorm sums the PRODUCER's epilogue left in `stats`)
//   -> optional exact GELU -> bf16 -> ds_write_b128 into the XOR-swizzled image the fragment reads expect.
// A first version took the rows through registers (global load -> transform -> ds_write): every K-slab then paid a
// global latency the compiler would not let span the loop, 2-4x slower than the launches it replaced on the deep-K
// layers.  The normalised tensor is still written once (by the workgroups of column tile 0) when a weight gradient
// needs it.  Two loop shapes:
//   * K <= 256 ("resident"): all of the tile's normalised A rows stay in LDS and the workgroup walks over the output
//     column chunks, streaming only weights -- fc1's N = 4..8 x K would otherwise normalise the same rows once per chunk;
//   * deeper K ("streaming", bf16 input): raw slabs and weight slabs share one ring, the transform runs one slab ahead
//     of the MFMAs.
// The epilogue (bias, residual + DropPath scale, GroupNorm sums of the output, fp32 / bf16 stores) is conv_common.h's.
#include <stdlib.h>
#include "conv_common.h"

using namespace crdk;

namespace {

constexpr int BK = 64;
constexpr int MAXRES = 4;                  // resident mode: at most this many K-slabs
typedef __attribute__((address_space(3))) void* lds_ptr;

struct GnIn {
  const void* x; int x_f32;               // raw input [B][IH*IW][x_ld] (+ channel offset applied), fp32 or bf16
  const float* stats; int gmul;           // [B][Cin/16][2] slab sums of x; a group = gmul slabs
  const float* gamma; const float* beta;  // [Cin]
  float inv_count;                        // 1 / (pixels per sample * channels per group)
  bf16_t* xn; int xn_ld; long long xn_bstride;    // optional store of act(GN(x)) (bf16), nullptr = none
  int n_chunks;                           // resident mode: column chunks of BN per workgroup (grid.y covers the rest)
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- pieces shared by the two kernels -----------------------------------------------------------------------------
template <int BM, int XF32>
struct ATile {
  static constexpr int ROW_BYTES = XF32 ? 256 : 128;                 // one K-slab of a raw row
  static constexpr int SLAB_BYTES = BM * ROW_BYTES;
  static constexpr int DMA_IT = XF32 ? BM / 16 : BM / 32;            // wave-instructions per wave and slab
  static constexpr int A_IT = BM / 32;                               // transform rows per thread and slab
};

// (scale, shift) of every input channel of sample b -> tab[Cin]
__device__ __forceinline__ void build_table(const ConvK& a, const GnIn& gi, int b, float2* tab) {
  const float* stb = gi.stats + (long long)b * (a.Cin >> 4) * 2;
  for (int c = threadIdx.x; c < a.Cin; c += 256) {
    float mean, rstd;
    gn_mean_rstd(stb, ((c >> 4) / gi.gmul) * gi.gmul, gi.gmul, gi.inv_count, mean, rstd);
    const float ga = gi.gamma[c] * rstd;
    tab[c] = make_float2(ga, gi.beta[c] - mean * ga);
  }
}

// LDS-DMA of raw slab kt of this workgroup's BM rows into `dst` (row-major, ROW_BYTES per row, lane-linear)
template <int BM, int XF32>
__device__ __forceinline__ void dma_raw(const ConvK& a, const __amdgpu_buffer_rsrc_t& rx, int m0, int kt, char* dst, int wv, int l) {
  using T = ATile<BM, XF32>;
  const unsigned OOB = 0x80000000u;
#pragma unroll
  for (int i = 0; i < T::DMA_IT; ++i) {
    int row, kf;
    if (XF32) { row = 4 * wv + 16 * i + (l >> 4); kf = kt * BK + (l & 15) * 4; }
    else { row = 8 * wv + 32 * i + (l >> 3); kf = kt * BK + (l & 7) * 8; }
    const int m = m0 + row;
    int kc = kf, ky = 0, kx = 0;
    if (a.KW > 1) { const int tap = kf / a.Cin; kc = kf - tap * a.Cin; ky = tap / a.KW; kx = tap - ky * a.KW; }
    const int oy = m / a.OW, ox = m - oy * a.OW;
    const bool ok = m < a.OHW && kf < a.Ktot;
    const unsigned off = ok ? (unsigned)((((oy * a.stride + ky) * a.IW + (ox * a.stride + kx)) * a.x_ld + kc) * (XF32 ? 4 : 2)) : OOB;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(dst + ((XF32 ? 4 : 8) * wv + (XF32 ? 16 : 32) * i) * T::ROW_BYTES), 16, off, 0, 0, 0);
#else
    (void)off;
#endif
  }
}

// raw slab (LDS) -> normalised bf16 operand slab (LDS, k_igemm's swizzle) [+ global store of the normalised rows]
template <int BM, int XF32, int ACT>
__device__ __forceinline__ void transform(const ConvK& a, const GnIn& gi, const float2* tab, int b, int m0, int kt, const char* raw,
                                          bf16_t* cooked, bool store_xn, int r0, int l) {
  using T = ATile<BM, XF32>;
  const int g = (l & 7) ^ ((r0 >> 1) & 7);          // K granule that belongs in slot l&7 of this thread's rows
  const int kf = kt * BK + g * 8;
  const bool kok = kf < a.Ktot;
  int kc = kf, ky = 0, kx = 0;
  if (a.KW > 1) { const int tap = kf / a.Cin; kc = kf - tap * a.Cin; ky = tap / a.KW; kx = tap - ky * a.KW; }
  float sc[8], sh[8];
  if (kok) {
    typedef __attribute__((ext_vector_type(4))) float f32x4t;
    const f32x4t* tp = reinterpret_cast<const f32x4t*>(tab + kc);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const f32x4t v = tp[j]; sc[2 * j] = v[0]; sh[2 * j] = v[1]; sc[2 * j + 1] = v[2]; sh[2 * j + 1] = v[3]; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = sh[j] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < T::A_IT; ++i) {
    const int row = r0 + 32 * i, m = m0 + row;
    float v[8];
    if (XF32) {
      // (ext_vector_type reads: behind a float4 / uint4 struct load the compiler waits vmcnt(0) -- common.h, lds_barrier)
      typedef __attribute__((ext_vector_type(4))) float f32x4v;
      const f32x4v* p = reinterpret_cast<const f32x4v*>(raw + row * T::ROW_BYTES + g * 32);
      const f32x4v lo = p[0], hi = p[1];
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    } else {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
      const u32x4v u = *reinterpret_cast<const u32x4v*>(raw + row * T::ROW_BYTES + g * 16);
      v[0] = bf_lo(u[0]); v[1] = bf_hi(u[0]); v[2] = bf_lo(u[1]); v[3] = bf_hi(u[1]);
      v[4] = bf_lo(u[2]); v[5] = bf_hi(u[2]); v[6] = bf_lo(u[3]); v[7] = bf_hi(u[3]);
    }
    const bool ok = kok && m < a.OHW;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = v[j] * sc[j] + sh[j];
      if (ACT == 1) y = gelu_exact(y);
      v[j] = ok ? y : 0.f;
    }
    uint4 q;
    q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]); q.z = pack_bf2(v[4], v[5]); q.w = pack_bf2(v[6], v[7]);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4c;
    *reinterpret_cast<u32x4c*>(cooked + row * BK + (l & 7) * 8) = u32x4c{q.x, q.y, q.z, q.w};
    if (store_xn && ok) {
      const int oy = m / a.OW, ox = m - oy * a.OW;
      const long long pix = (long long)(oy * a.stride + ky) * a.IW + (ox * a.stride + kx);
      *reinterpret_cast<uint4*>(gi.xn + (long long)b * gi.xn_bstride + pix * gi.xn_ld + kc) = q;
    }
  }
}

template <int BN>
__device__ __forceinline__ void dma_w(const ConvK& a, const __amdgpu_buffer_rsrc_t& rw, int n0, int kt, bf16_t* dst, int wv, int l, int r0) {
  const unsigned OOB = 0x80000000u;
  const int g = (l & 7) ^ ((r0 >> 1) & 7);
  const int kf = kt * BK + g * 8;
#pragma unroll
  for (int j = 0; j < BN / 32; ++j) {
    const int ng = n0 + r0 + 32 * j;
    const unsigned off = (ng < a.Cout && kf < a.Ktot) ? (unsigned)((ng * a.Ktot + kf) * 2) : OOB;
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + (8 * wv + 32 * j) * BK), 16, off, 0, 0, 0);
#else
    (void)off;
#endif
  }
}

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void mfma_slab(const bf16_t* sa, const bf16_t* sb, f32x16 (&acc)[TM][TN], int wm, int wn, int l) {
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    bf16x8 af[TM], bfr[TN];
    const int gi2 = ks * 2 + (l >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 32 + (l & 31);
      af[i] = *reinterpret_cast<const bf16x8*>(&sa[row * BK + ((gi2 ^ ((row >> 1) & 7)) << 3)]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = (wn * TN + j) * 32 + (l & 31);
      bfr[j] = *reinterpret_cast<const bf16x8*>(&sb[row * BK + ((gi2 ^ ((row >> 1) & 7)) << 3)]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  }
}

template <int TM, int TN, int WN>
__device__ __forceinline__ void init_acc(const ConvK& a, f32x16 (&acc)[TM][TN], int b, int n0, int wn, int l) {
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (l & 31);
    const float bias_v = (a.bias && col < a.Cout) ? a.bias[(long long)b * a.bias_bstride + col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bias_v;
  }
}

// ---- resident mode: K <= MAXRES slabs; the workgroup keeps its normalised rows and walks over column chunks ----------
// LDS: [epilogue staging | raw slabs (aliased)] [cooked nK x BM x 64] [weight ring NSB x BN x 64] [table]
template <int WM, int WN, int TM, int TN, int NSB, int XF32, int ACT>
__global__ __launch_bounds__(256) void k_gngemm_res(ConvK a, GnIn gi, int stage_bytes) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  using T = ATile<BM, XF32>;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  const int nK = (a.Ktot + BK - 1) / BK;
  char* raw = reinterpret_cast<char*>(lds);                                        // nK slabs, dead after the transform
  bf16_t* cooked = reinterpret_cast<bf16_t*>(raw + stage_bytes);                   // [nK][BM][BK]
  bf16_t* sB = cooked + nK * BM * BK;                                              // [NSB][BN][BK]
  float2* tab = reinterpret_cast<float2*>(sB + NSB * BN * BK);

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int b = blockIdx.z, m0 = blockIdx.x * BM;
  const int chunk0 = blockIdx.y * gi.n_chunks;
  const int r0 = 8 * wv + (l >> 3);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(gi.x) + (long long)b * a.x_bstride * (XF32 ? 4 : 2)), 0, (int)(a.x_bstride * (XF32 ? 4 : 2)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const int total = gi.n_chunks * nK;                // weight steps: (chunk, slab) flattened

  for (int kt = 0; kt < nK; ++kt) dma_raw<BM, XF32>(a, rx, m0, kt, raw + kt * T::SLAB_BYTES, wv, l);
#pragma unroll
  for (int s = 0; s < NSB - 1; ++s)                   // past-the-end steps are still issued (zero fill): uniform waits
    dma_w<BN>(a, rw, (chunk0 + s / nK) * BN, s % nK, sB + s * BN * BK, wv, l, r0);
  if (!(a.dbg & 1)) build_table(a, gi, b, tab);       // its dependent loads end in a full wait: the raw slabs have landed too
  wait_vm<0>();
  __syncthreads();
  if (a.dbg & 8) return;
  if (!(a.dbg & 2)) for (int kt = 0; kt < nK; ++kt)
    transform<BM, XF32, ACT>(a, gi, tab, b, m0, kt, raw + kt * T::SLAB_BYTES, cooked + kt * BM * BK, gi.xn != nullptr && blockIdx.y == 0, r0, l);

  f32x16 acc[TM][TN];
  int s = 0;
  for (int c = 0; c < gi.n_chunks; ++c) {
    const int n0 = (chunk0 + c) * BN;
    if (n0 >= a.Cout) break;                          // workgroup-uniform
    init_acc<TM, TN, WN>(a, acc, b, n0, wn, l);
    for (int kt = 0; kt < nK; ++kt, ++s) {
      wait_vm<(NSB - 2) * (BN / 32)>();               // weight step s has landed (this thread's share) ...
      lds_barrier();                                  // ... everyone's; cooked rows / the previous step's slot are settled
      const int sn = s + NSB - 1;
      dma_w<BN>(a, rw, (chunk0 + sn / nK) * BN, sn % nK, sB + (sn % NSB) * BN * BK, wv, l, r0);
      mfma_slab<TM, TN, WM, WN>(cooked + kt * BM * BK, sB + (s % NSB) * BN * BK, acc, wm, wn, l);
    }
    // epilogue of this chunk in the staging area (the raw slabs' space); in-flight weight DMAs target the ring, not it
    lds_barrier();
    if (a.dbg & 4) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(a.y)[0] = 1.f; continue; }
    conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, blockIdx.x, lds,
                                  [&](int i, int rr, bool& valid, int& row) { row = m0 + (wm * TM + i) * 32 + rr; valid = row < a.OHW; },
                                  [&](int rl, bool& valid, int& row) { row = m0 + rl; valid = row < a.OHW; });
    __syncthreads();
  }
  (void)total;
  wait_vm<0>();
}

// ---- streaming mode (bf16 input, deep K): raw and weight slabs share a ring of NS stages --------------------------
// LDS: [raw ring NS x BM x 64 (bf16) | weight ring NS x BN x 64 | cooked 2 x BM x 64] [table]; epilogue staging at 0
template <int WM, int WN, int TM, int TN, int NS, int ACT>
__global__ __launch_bounds__(256) void k_gngemm_str(ConvK a, GnIn gi) {
  static_assert(WM * WN == 4 && NS >= 3, "4 waves, >= 3 stages");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  using T = ATile<BM, 0>;
  constexpr int PER = T::DMA_IT + BN / 32;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  char* raw = reinterpret_cast<char*>(lds);                                        // [NS][BM][128 B]
  bf16_t* sB = reinterpret_cast<bf16_t*>(raw + NS * T::SLAB_BYTES);                // [NS][BN][BK]
  bf16_t* cooked = sB + NS * BN * BK;                                              // [2][BM][BK]
  float2* tab = reinterpret_cast<float2*>(cooked + 2 * BM * BK);

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int b = blockIdx.z, m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int r0 = 8 * wv + (l >> 3);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(gi.x) + (long long)b * a.x_bstride * 2), 0, (int)(a.x_bstride * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const int nK = (a.Ktot + BK - 1) / BK;
  const bool store_xn = gi.xn != nullptr && blockIdx.y == 0;
  auto stage = [&](int kt) {                         // slabs past the K range: zero fill, no traffic (uniform wait counts)
    dma_raw<BM, 0>(a, rx, m0, kt, raw + (kt % NS) * T::SLAB_BYTES, wv, l);
    dma_w<BN>(a, rw, n0, kt, sB + (kt % NS) * BN * BK, wv, l, r0);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) stage(s);
  build_table(a, gi, b, tab);
  f32x16 acc[TM][TN];
  init_acc<TM, TN, WN>(a, acc, b, n0, wn, l);
  wait_vm<(NS - 2) * PER>();                          // slab 0 has landed
  lds_barrier();                                      // ... everyone's share, and the table
  transform<BM, 0, ACT>(a, gi, tab, b, m0, 0, raw, cooked, store_xn, r0, l);
  for (int kt = 0; kt < nK; ++kt) {
    // slabs issued: 0 .. kt+NS-2.  Slab kt+1 (transformed below) must have landed: NS-3 later slabs may be in flight.
    wait_vm<(NS - 3) * PER>();
    lds_barrier();         // cooked[kt&1] complete; MFMAs of kt-1 done with cooked[(kt-1)&1] and ring slot (kt-1)%NS
    stage(kt + NS - 1);
    if (kt + 1 < nK)
      transform<BM, 0, ACT>(a, gi, tab, b, m0, kt + 1, raw + ((kt + 1) % NS) * T::SLAB_BYTES, cooked + ((kt + 1) & 1) * BM * BK, store_xn, r0, l);
    mfma_slab<TM, TN, WM, WN>(cooked + (kt & 1) * BM * BK, sB + (kt % NS) * BN * BK, acc, wm, wn, l);
  }
  wait_vm<0>();
  __syncthreads();
  conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, blockIdx.x, lds,
                                [&](int i, int rr, bool& valid, int& row) { row = m0 + (wm * TM + i) * 32 + rr; valid = row < a.OHW; },
                                [&](int rl, bool& valid, int& row) { row = m0 + rl; valid = row < a.OHW; });
}

template <int BM, int BN>
constexpr size_t epilogue_bytes() { return (size_t)BM * (BN + 8) * 4 + 256 * 16 * 4 + 2048; }   // fp32 staging tile + folds behind it

template <int WM, int WN, int TM, int TN, int NSB, int XF32, int ACT>
int launch_res(const ConvK& k0, GnIn gi, int B, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  using T = ATile<BM, XF32>;
  ConvK k = k0;
  k.n_tiles = cdiv(k.OHW, BM);
  const int nK = cdiv(k.Ktot, BK), chunks = cdiv(k.Cout, BN);
  // column chunks per workgroup: all of them when the row tiles alone cover the chip, else spread over grid.y
  int gy = 1;
  while (gy < chunks && (long long)k.n_tiles * B * gy < 256) ++gy;
  gi.n_chunks = cdiv(chunks, gy);
  gy = cdiv(chunks, gi.n_chunks);
  size_t stage_bytes = (size_t)nK * T::SLAB_BYTES;
  if (stage_bytes < epilogue_bytes<BM, BN>()) stage_bytes = epilogue_bytes<BM, BN>();
  stage_bytes = (stage_bytes + 1023) / 1024 * 1024;
  const size_t lds = stage_bytes + (size_t)(nK * BM + NSB * BN) * BK * 2 + (size_t)k.Cin * sizeof(float2);
  CRD_UNSUPPORTED(lds <= 160 * 1024, "crd_gn_conv: resident tile does not fit in LDS");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gngemm_res<WM, WN, TM, TN, NSB, XF32, ACT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  k.lds_bytes = (int)stage_bytes;
  hipLaunchKernelGGL((k_gngemm_res<WM, WN, TM, TN, NSB, XF32, ACT>), dim3(k.n_tiles, gy, B), dim3(256), lds, st, k, gi, (int)stage_bytes);
  CRD_LAUNCH_CHECK("crd_gn_conv");
  return CRD_OK;
}

template <int WM, int WN, int TM, int TN, int NS, int ACT>
int launch_str(const ConvK& k0, GnIn gi, int B, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  ConvK k = k0;
  k.n_tiles = cdiv(k.OHW, BM);
  size_t lds = (size_t)NS * (BM + BN) * BK * 2 + (size_t)2 * BM * BK * 2 + (size_t)k.Cin * sizeof(float2);
  if (lds < epilogue_bytes<BM, BN>()) lds = epilogue_bytes<BM, BN>();
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gngemm_str<WM, WN, TM, TN, NS, ACT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  k.lds_bytes = (int)lds;
  gi.n_chunks = 1;
  hipLaunchKernelGGL((k_gngemm_str<WM, WN, TM, TN, NS, ACT>), dim3(k.n_tiles, cdiv(k.Cout, BN), B), dim3(256), lds, st, k, gi);
  CRD_LAUNCH_CHECK("crd_gn_conv");
  return CRD_OK;
}

template <int XF32, int ACT>
int dispatch(const ConvK& k, const GnIn& gi, int B, hipStream_t st) {
  const int nK = cdiv(k.Ktot, BK);
  if (nK <= MAXRES) {
    if (k.Cout <= 64) return launch_res<2, 2, 1, 1, 3, XF32, ACT>(k, gi, B, st);          // 64 x 64 chunks
    return launch_res<2, 2, 1, 2, 3, XF32, ACT>(k, gi, B, st);                             // 64 x 128 chunks
  }
  if (XF32) { crd_set_error("crd_gn_conv: fp32 input is supported up to K = %d", MAXRES * BK); return CRD_E_UNSUPPORTED; }
  if (k.Cout <= 64) return launch_str<2, 2, 1, 1, 5, ACT>(k, gi, B, st);                   // 64 x 64
  return launch_str<2, 2, 1, 2, 5, ACT>(k, gi, B, st);                                     // 64 x 128
}

}  // namespace

extern "C" int crd_gn_conv(const crd_conv_desc* d, const crd_gn_input* n, crd_stream_t stream) {
  CRD_CHECK_ARG(d && n && d->x && d->w && d->y && n->stats && n->gamma && n->beta, "crd_gn_conv: null pointer");
  CRD_CHECK_ARG(d->Cin % 16 == 0 && d->x_ld % 8 == 0 && d->x_coff % 8 == 0, "crd_gn_conv: Cin must be a multiple of 16, x_ld/x_coff of 8");
  CRD_CHECK_ARG(d->B > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0 && d->KH == d->KW && d->KH >= 1, "crd_gn_conv: bad dims");
  CRD_UNSUPPORTED(d->stride == d->KH && d->pad == 0 && d->gather_mode == 0 && d->out_mode == 0 && d->IH == d->OH * d->stride &&
                  d->IW == d->OW * d->stride, "crd_gn_conv: pointwise or non-overlapping patch convolutions only");
  CRD_CHECK_ARG(n->gmul >= 1 && (d->Cin / 16) % n->gmul == 0 && (n->act == 0 || n->act == 1), "crd_gn_conv: bad GroupNorm arguments");
  CRD_CHECK_ARG(!(d->res && !d->y_f32), "crd_gn_conv: residual epilogue needs fp32 output");
  CRD_CHECK_ARG(!d->stats || d->Cout % 16 == 0, "crd_gn_conv: stats need Cout %% 16 == 0");
  CRD_UNSUPPORTED(d->red_x == nullptr && d->stats_partial == nullptr, "crd_gn_conv: no fused backward reduce / partial statistics here");
  CRD_UNSUPPORTED((long long)d->Cout * d->KH * d->KW * d->Cin < (1ll << 30) && d->Cin <= 4096 &&
                  (long long)d->IH * d->IW * d->x_ld * (n->x_f32 ? 4 : 2) < (1ll << 31), "crd_gn_conv: tensor too large for 32-bit byte offsets");
  CRD_CHECK_ARG(!n->xn || (n->xn_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(n->xn) & 15) == 0), "crd_gn_conv: xn rows must be 16-byte aligned");
  ConvK k;
  k.x = nullptr; k.x_ld = d->x_ld;
  k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin; k.x_bstride = (long long)d->IH * d->IW * d->x_ld;
  k.w = reinterpret_cast<const bf16_t*>(d->w);
  k.Cout = d->Cout; k.KW = d->KW; k.stride = d->stride; k.pad = 0; k.Ktot = d->KH * d->KW * d->Cin;
  k.OW = d->OW; k.OHW = d->OH * d->OW; k.gather_mode = 0;
  k.y_ld = d->y_ld; k.y_f32 = d->y_f32;
  k.out_mode = 0; k.patch_k = 0; k.patch_c = 0; k.YW = d->OW;
  k.y_bstride = (long long)d->OH * d->OW * d->y_ld;
  k.y = d->y_f32 ? (void*)(reinterpret_cast<float*>(d->y) + d->y_coff) : (void*)(reinterpret_cast<bf16_t*>(d->y) + d->y_coff);
  k.bias = d->bias; k.bias_bstride = d->bias_bstride; k.act = d->act;
  k.res = d->res; k.res_ld = d->res_ld; k.res_bstride = (long long)d->OH * d->OW * d->res_ld; k.res_scale = d->res_scale;
  k.accumulate = d->accumulate; k.stats = d->stats; k.G16 = d->Cout / 16;
  k.stats_partial = nullptr; k.n_tiles = 0; k.col0 = 0;
  k.chan = d->chan_sums;
  CRD_UNSUPPORTED(!d->chan_sums || (d->stats && (d->y_f32 || d->res)), "crd_gn_conv: chan_sums needs stats and an fp32 / residual output");
  k.vec_ok = (d->y_coff % 8 == 0) && ((reinterpret_cast<uintptr_t>(d->y) & 15) == 0);
  k.vecf_ok = d->y_f32 && d->y_coff % 4 == 0 && d->y_ld % 4 == 0 && d->Cout % 4 == 0 && (reinterpret_cast<uintptr_t>(d->y) & 15) == 0 &&
              (!d->res || (d->res_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(d->res) & 15) == 0));
  k.lds_bytes = 0;
  k.red_x = nullptr; k.red_x_ld = 0; k.red_x_bstride = 0; k.red_stats = nullptr; k.red_gamma = nullptr; k.red_beta = nullptr;
  k.red_gmul = 1; k.red_act = 0; k.red_r = nullptr;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("CRD_DBG"); dbg = e ? atoi(e) : 0; } k.dbg = dbg; }
  GnIn gi;
  gi.x_f32 = n->x_f32;
  gi.x = n->x_f32 ? (const void*)(reinterpret_cast<const float*>(d->x) + d->x_coff) : (const void*)(reinterpret_cast<const bf16_t*>(d->x) + d->x_coff);
  CRD_CHECK_ARG((reinterpret_cast<uintptr_t>(gi.x) & 15) == 0 && (!n->x_f32 || d->x_ld % 4 == 0), "crd_gn_conv: x rows must be 16-byte aligned");
  gi.stats = n->stats; gi.gmul = n->gmul; gi.gamma = n->gamma; gi.beta = n->beta;
  gi.inv_count = 1.f / ((float)d->IH * (float)d->IW * 16.f * (float)n->gmul);
  gi.xn = reinterpret_cast<bf16_t*>(n->xn); gi.xn_ld = n->xn_ld; gi.xn_bstride = (long long)d->IH * d->IW * n->xn_ld;
  gi.n_chunks = 1;
  hipStream_t st = as_stream(stream);
  if (n->x_f32) return n->act ? dispatch<1, 1>(k, gi, d->B, st) : dispatch<1, 0>(k, gi, d->B, st);
  return n->act ? dispatch<0, 1>(k, gi, d->B, st) : dispatch<0, 0>(k, gi, d->B, st);
}
