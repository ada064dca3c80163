// Fused Mlp of an encoder Block on gfx950 (reference: Block.forward / Mlp.forward, src/models/simplified_attention.py:34-43,
// 141-145):   x2 = x1 + drop_path( fc2( GELU( norm2( dwconv3x3( norm1( fc1( Block.norm2(x1) ) ) ) ) ) ) )
//
// In stages 3 and 4 of the encoder (16 x 26 and 8 x 13 pixels at 256 x 416) this chain was four launches -- fc1 with the
// GroupNorm folded into its operand load, the depthwise conv, GroupNorm + GELU, fc2 -- of 7-13 us each for a few MFLOP: each
// launch is a dependent chain (arguments -> statistics -> operands -> MFMA -> store) through HBM and there are 34 blocks of
// them (the encoder is 10 % of the FLOPs and half of the step).  The structure of the block makes the whole
// chain independent per (sample, 64-channel slab of the hidden tensor):
//   * Mlp.norm1 groups are 16 hidden channels, Mlp.norm2 groups are hidden / (dim / 16) = 64 hidden channels in these stages
//     (ff_expansion 4), both GroupNorms are per sample;
//   * the depthwise conv is per channel;
//   * fc2 is a sum over hidden channels.
// So ONE workgroup owns (sample b, hidden channels 64 s .. 64 s + 63) for ALL pixels of the sample: fc1 on MFMA (A = the
// slab's 64 weight rows, B = normalised x1 streamed through LDS in 32-channel chunks), its GroupNorm statistics over the
// whole sample inside the workgroup, norm1 applied in LDS, the 3 x 3 stencil straight from LDS, norm2 + exact GELU in LDS,
// and the slab's contribution to fc2 on MFMA.  The hidden tensor never waits for another workgroup; h1 / h2 / h3 are
// still written (the backward pass and the weight gradients read them) but nothing on the chain reads them back.
// The fc2 contributions of the hidden / 64 slabs are fp32 partial tiles; crd_mlp_reduce adds them in slab order (fixed:
// reproducible), rounds to bf16 where the reference's autocast does, applies bias, DropPath scale and the residual, and
// produces the GroupNorm sums of x2 for the next block: 2 launches per Mlp instead of 4.
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MW = 16, MT = 64 * MW;            // waves / threads of a k_mlp_fwd workgroup
constexpr int SLAB = 64;                        // hidden channels per workgroup = one Mlp.norm2 group
constexpr int XC = 32;                          // x channels per staged chunk (64-byte LDS rows)

struct MlpK {
  const float* x1; const crd_sum_t* st2; const float* g2; const float* b2;     // Block.norm2 input / sums / affine
  const bf16_t* w1; const float* bias1;                                        // fc1 packed [hid][Cs], bias [hid]
  const float* n1g; const float* n1b;                                          // Mlp.norm1 [hid]
  const float* w9; const float* dwb;                                           // depthwise [9][hid] fp32, bias [hid]
  const float* n2g; const float* n2b;                                          // Mlp.norm2 [hid]
  const bf16_t* w2;                                                            // fc2 packed [Cs][hid]
  bf16_t* xn2; bf16_t* h1; bf16_t* h2; bf16_t* h3;                              // optional outputs (nullptr: not stored)
  crd_sum_t* sth1; crd_sum_t* sth2;                                            // g16 sums of h1 / h2 [B][hid/16][2]
  float* part;                                                                 // [hid/64][B][N][Cs] fp32
  int B, H, W, N, NP, Cs, hid;                                                 // NP = N rounded up to 32
};

#ifdef CRD_MLP_PROF
__device__ unsigned long long g_mlp_prof[16];
#define MLP_STAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0) g_mlp_prof[i] = __builtin_readcyclecounter(); } while (0)
#else
#define MLP_STAMP(i) do {} while (0)
#endif
__device__ __forceinline__ int swz128(int p, int g) { return g ^ ((p >> 1) & 7); }     // 128-byte rows, 8 granules
__device__ __forceinline__ int swz64(int p, int g) { return g ^ ((p >> 2) & 3); }      // 64-byte rows, 4 granules

// 16 waves per workgroup (one workgroup per CU: ~150 KB of LDS): every phase below is a chain of LDS round trips, and with one
// wave per SIMD (the first version: 4 waves, 61 us per launch -- fc1 19, stencil 14, fc2 11, GELU 7 us; tools/prof_mlp.py)
// nothing hides them; four waves per SIMD do.  NCH = C / 32 chunks of x, XIT = ceil(pixels / 128) rows per thread and chunk.
template <int NCH, int XIT>
__global__ __launch_bounds__(MT) void k_mlp_fwd(MlpK a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Cs = a.Cs, N = a.N, NP = a.NP, W1LD = Cs + 8;                       // fc1 rows padded by one granule: conflict-free reads
  float2* tab2 = reinterpret_cast<float2*>(smem);                              // [Cs] (scale, shift) of Block.norm2
  float* red = reinterpret_cast<float*>(tab2 + Cs);                            // [16 waves][8] + [8] statistics folds
  float* sW9 = red + 192;                                                      // [10][64]: depthwise taps + bias of the slab
  bf16_t* sW1 = reinterpret_cast<bf16_t*>(sW9 + 640);                          // [64][Cs + 8]
  bf16_t* sW2 = sW1 + SLAB * W1LD;                                             // [Cs][64], swz128
  bf16_t* bufA = sW2 + Cs * SLAB;                                              // [NP][64]: h1 -> norm1(h1)
  bf16_t* bufB = bufA + NP * SLAB;                                             // 2 x [NP][32] x chunks, then [NP][64]: h2 -> h3
  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int slab = blockIdx.x, b = blockIdx.y, slabs = gridDim.x;
  const int hc0 = slab * SLAB;
  const float* xb = a.x1 + (long long)b * N * Cs;
  const int NT = NP >> 5;                                                      // 32-pixel row tiles
  MLP_STAMP(0);

  // ---- phase 0: the first three x chunks, the Block.norm2 sums and both weight slabs are requested together
  const int xg = t & 7, prow = t >> 3;                                         // 4-channel group of a chunk / pixel row (+ 128 i)
  float4 xr[3][XIT];                                                           // chunk c lives in xr[c % 3]: requested two chunks ahead
  auto load_chunk = [&](int c, float4 (&dst)[XIT]) {
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int p = prow + 128 * i;
      dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < N) dst[i] = *reinterpret_cast<const float4*>(xb + (long long)p * Cs + c * XC + xg * 4);
    }
  };
  load_chunk(0, xr[0]);
  if (NCH > 1) load_chunk(1, xr[1]);
  for (int c = t; c < Cs; c += MT) {
    float mean, rstd;
    gn_mean_rstd(a.st2 + (long long)b * (Cs >> 4) * 2, c >> 4, 1, (float)N * 16.f, mean, rstd);
    const float ga = a.g2[c] * rstd;
    tab2[c] = make_float2(ga, a.b2[c] - mean * ga);
  }
  {
    const int G1 = Cs >> 3;                                                    // 16-byte granules per fc1 row
    for (int i = t; i < SLAB * G1; i += MT) {
      const int r = i / G1, g = i - r * G1;
      *reinterpret_cast<uint4*>(sW1 + r * W1LD + g * 8) = *reinterpret_cast<const uint4*>(a.w1 + (long long)(hc0 + r) * Cs + g * 8);
    }
    for (int i = t; i < Cs * 8; i += MT) {
      const int co = i >> 3, g = i & 7;
      *reinterpret_cast<uint4*>(sW2 + co * SLAB + swz128(co, g) * 8) =
          *reinterpret_cast<const uint4*>(a.w2 + (long long)co * a.hid + hc0 + g * 8);
    }
  }
  for (int i = t; i < 640; i += MT) sW9[i] = i < 576 ? a.w9[(long long)(i >> 6) * a.hid + hc0 + (i & 63)] : a.dwb[hc0 + (i & 63)];
  __syncthreads();                                                             // tab2, the weight slabs and the stencil taps are in LDS
  MLP_STAMP(1);

  // ---- phase 1: h1 = bf16(fc1(xn) + bias), swapped operands: a lane holds ONE pixel (column) and 16 hidden channels.
  // MFMA units (row tile, column tile j): wave w owns units w and w + 16 (both have j = w & 1).
  const int uj = wv & 1;
  f32x16 acc[2];
  {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = a.bias1[hc0 + uj * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][r] = bv[r];
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    bf16_t* xs = bufB + (c & 1) * NP * XC;
    float2 sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sc[j] = tab2[c * XC + xg * 4 + j];
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int p = prow + 128 * i;
      const float4 v = xr[c % 3][i];
      if (p < NP) {
        const uint2 q = make_uint2(pack_bf2(v.x * sc[0].x + sc[0].y, v.y * sc[1].x + sc[1].y),
                                   pack_bf2(v.z * sc[2].x + sc[2].y, v.w * sc[3].x + sc[3].y));
        *reinterpret_cast<uint2*>(xs + p * XC + swz64(p, xg >> 1) * 8 + (xg & 1) * 4) = p < N ? q : make_uint2(0, 0);
        // Block.norm2(x1) for the fc1 weight gradient: chunk c is written by slab c mod slabs (every slab has the values)
        if (a.xn2 && (c % slabs) == slab && p < N) *reinterpret_cast<uint2*>(a.xn2 + ((long long)b * N + p) * Cs + c * XC + xg * 4) = q;
      }
    }
    __syncthreads();
    if (c + 2 < NCH) load_chunk(c + 2, xr[(c + 2) % 3]);                      // two chunks ahead
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int gi = ks * 2 + (l >> 5);
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW1 + (uj * 32 + (l & 31)) * W1LD + c * XC + gi * 8);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int tile = (wv >> 1) + 8 * k;
        if (tile < NT) {                                                       // wave-uniform
          const int p = tile * 32 + (l & 31);
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + p * XC + swz64(p, gi) * 8);
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[k], 0, 0, 0);
        }
      }
    }
  }
  MLP_STAMP(2);
  // rounded h1 -> LDS (a lane's channels of its column tile: 8 q + 4 half + {0..3}, q = 0..3: 8-byte runs) + Mlp.norm1 sums
  {
    float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};                               // 16-channel groups 2 uj + (q >> 1)
    const int half = l >> 5;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tile = (wv >> 1) + 8 * k;
      if (tile >= NT) continue;
      const int p = tile * 32 + (l & 31);
      const bool pok = p < N;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 d = make_uint2(pack_bf2(acc[k][4 * q], acc[k][4 * q + 1]), pack_bf2(acc[k][4 * q + 2], acc[k][4 * q + 3]));
        *reinterpret_cast<uint2*>(bufA + p * SLAB + swz128(p, uj * 4 + q) * 8 + half * 4) = d;
        if (pok) {
          const float v0 = bf_lo(d.x), v1 = bf_hi(d.x), v2 = bf_lo(d.y), v3 = bf_hi(d.y);
          s[q >> 1] += (v0 + v1) + (v2 + v3);
          ss[q >> 1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) { s[g] = wave_sum(s[g]); ss[g] = wave_sum(ss[g]); }
    if (l == 0) { red[wv * 4] = s[0]; red[wv * 4 + 1] = ss[0]; red[wv * 4 + 2] = s[1]; red[wv * 4 + 3] = ss[1]; }
  }
  __syncthreads();
  if (t < 8) {      // (16-channel group g4 = t >> 1, moment t & 1): the eight waves with column tile g4 >> 1, in order (a fixed tree)
    const int g4 = t >> 1, j = g4 >> 1;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[(2 * w + j) * 4 + (g4 & 1) * 2 + (t & 1)];
    red[128 + t] = v;
    a.sth1[((long long)b * (a.hid >> 4) + (hc0 >> 4) + g4) * 2 + (t & 1)] = to_fx(v, STAT_ONE);
  }
  __syncthreads();
  MLP_STAMP(3);

  // ---- phase 2: h1 -> global, norm1(h1) in place.  Thread = (pixel prow + 128 i, granule hg): 128 contiguous bytes per pixel
  const int hg = t & 7, hch = hc0 + hg * 8;
  {
    float ga[8], be[8];
    load8t<1>(a.n1g, hch, ga);
    load8t<1>(a.n1b, hch, be);
    // (mean / rstd from the fixed-point values a reader of sth1 sees -- gn_mean_rstd's arithmetic -- so that this kernel and
    // the backward pass normalise with the same numbers)
    float mean, rstd;
    gn_moments(to_fx(red[128 + (hg >> 1) * 2], STAT_ONE), to_fx(red[128 + (hg >> 1) * 2 + 1], STAT_ONE), (float)N * 16.f, mean, rstd);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] *= rstd; be[j] -= mean * ga[j]; }
#pragma unroll 1
    for (int i = 0; i < XIT; ++i) {
      const int p = prow + 128 * i;
      if (p >= N) continue;
      uint4* cell = reinterpret_cast<uint4*>(bufA + p * SLAB + swz128(p, hg) * 8);
      const uint4 u = *cell;
      if (a.h1) *reinterpret_cast<uint4*>(a.h1 + ((long long)b * N + p) * a.hid + hch) = u;
      uint4 o;
      o.x = pack_bf2(bf_lo(u.x) * ga[0] + be[0], bf_hi(u.x) * ga[1] + be[1]);
      o.y = pack_bf2(bf_lo(u.y) * ga[2] + be[2], bf_hi(u.y) * ga[3] + be[3]);
      o.z = pack_bf2(bf_lo(u.z) * ga[4] + be[4], bf_hi(u.z) * ga[5] + be[5]);
      o.w = pack_bf2(bf_lo(u.w) * ga[6] + be[6], bf_hi(u.w) * ga[7] + be[7]);
      *cell = o;
    }
  }
  __syncthreads();
  MLP_STAMP(4);

  // ---- phase 3: h2 = bf16(dwconv3x3(norm1(h1)) + bias) from LDS (zero padding AFTER the normalisation), Mlp.norm2 sums.
  // No branch around the nine reads (they issue back to back): a tap outside the image reads the centre pixel and is masked.
  {
    float s = 0.f, ss = 0.f;
#pragma unroll 1
    for (int i = 0; i < XIT; ++i) {
      const int p = prow + 128 * i;
      if (p >= N) continue;
      const int y = p / a.W, x = p - y * a.W;
      float o[8];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sW9 + 576 + hg * 8), b1 = *reinterpret_cast<const f32x4*>(sW9 + 576 + hg * 8 + 4);
        o[0] = b0[0]; o[1] = b0[1]; o[2] = b0[2]; o[3] = b0[3]; o[4] = b1[0]; o[5] = b1[1]; o[6] = b1[2]; o[7] = b1[3];
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {               // three pixel reads in flight at a time; the taps come from LDS as well
        uint4 nb[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = y + ky - 1, xx = x + kx - 1;
          const bool ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
          const int q = ok ? yy * a.W + xx : p;
          uint4 u = *reinterpret_cast<const uint4*>(bufA + q * SLAB + swz128(q, hg) * 8);
          if (!ok) u = make_uint4(0, 0, 0, 0);
          nb[kx] = u;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const uint4 u = nb[kx];
          const float* wt = sW9 + (ky * 3 + kx) * 64 + hg * 8;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt), w1 = *reinterpret_cast<const f32x4*>(wt + 4);
          o[0] += bf_lo(u.x) * w0[0]; o[1] += bf_hi(u.x) * w0[1]; o[2] += bf_lo(u.y) * w0[2]; o[3] += bf_hi(u.y) * w0[3];
          o[4] += bf_lo(u.z) * w1[0]; o[5] += bf_hi(u.z) * w1[1]; o[6] += bf_lo(u.w) * w1[2]; o[7] += bf_hi(u.w) * w1[3];
        }
      }
      uint4 u;
      u.x = pack_bf2(o[0], o[1]); u.y = pack_bf2(o[2], o[3]); u.z = pack_bf2(o[4], o[5]); u.w = pack_bf2(o[6], o[7]);
      *reinterpret_cast<uint4*>(bufB + p * SLAB + swz128(p, hg) * 8) = u;
      if (a.h2) *reinterpret_cast<uint4*>(a.h2 + ((long long)b * N + p) * a.hid + hch) = u;
      s += ((bf_lo(u.x) + bf_hi(u.x)) + (bf_lo(u.y) + bf_hi(u.y))) + ((bf_lo(u.z) + bf_hi(u.z)) + (bf_lo(u.w) + bf_hi(u.w)));
      ss += ((bf_lo(u.x) * bf_lo(u.x) + bf_hi(u.x) * bf_hi(u.x)) + (bf_lo(u.y) * bf_lo(u.y) + bf_hi(u.y) * bf_hi(u.y))) +
            ((bf_lo(u.z) * bf_lo(u.z) + bf_hi(u.z) * bf_hi(u.z)) + (bf_lo(u.w) * bf_lo(u.w) + bf_hi(u.w) * bf_hi(u.w)));
    }
    // lane = (pixel lane << 3) | granule: fold the two granules of a 16-channel slab (bit 0) and the 8 pixel lanes (bits 3-5)
    s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
#pragma unroll
    for (int o2 = 8; o2 < 64; o2 <<= 1) { s += __shfl_xor(s, o2); ss += __shfl_xor(ss, o2); }
    if (l < 8 && (l & 1) == 0) { red[wv * 8 + l] = s; red[wv * 8 + l + 1] = ss; }
  }
  __syncthreads();
  if (t < 8) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < MW; ++w) v += red[w * 8 + t];                          // the sixteen waves in order
    red[128 + t] = v;
    a.sth2[((long long)b * (a.hid >> 4) + (hc0 >> 4) + (t >> 1)) * 2 + (t & 1)] = to_fx(v, STAT_ONE);
  }
  __syncthreads();
  MLP_STAMP(5);

  // ---- phase 4: h3 = bf16(GELU(norm2(h2))) in place (+ global); the group is the whole slab: its four 16-channel sums
  {
    float ga[8], be[8];
    load8t<1>(a.n2g, hch, ga);
    load8t<1>(a.n2b, hch, be);
    // (the sums are added as the fixed-point values a reader of sth2 would see, so that this kernel and the backward agree)
    long long qs = 0, qss = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) { qs += to_fx(red[128 + g * 2], STAT_ONE); qss += to_fx(red[128 + g * 2 + 1], STAT_ONE); }
    float mean, rstd;
    gn_moments(qs, qss, (float)N * 64.f, mean, rstd);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] *= rstd; be[j] -= mean * ga[j]; }
#pragma unroll 1
    for (int i = 0; i < XIT; ++i) {
      const int p = prow + 128 * i;
      if (p >= NP) continue;
      uint4* cell = reinterpret_cast<uint4*>(bufB + p * SLAB + swz128(p, hg) * 8);
      if (p >= N) { *cell = make_uint4(0, 0, 0, 0); continue; }                // rows of the last partial MFMA tile
      const uint4 u = *cell;
      uint4 o;
      o.x = pack_bf2(gelu_exact(bf_lo(u.x) * ga[0] + be[0]), gelu_exact(bf_hi(u.x) * ga[1] + be[1]));
      o.y = pack_bf2(gelu_exact(bf_lo(u.y) * ga[2] + be[2]), gelu_exact(bf_hi(u.y) * ga[3] + be[3]));
      o.z = pack_bf2(gelu_exact(bf_lo(u.z) * ga[4] + be[4]), gelu_exact(bf_hi(u.z) * ga[5] + be[5]));
      o.w = pack_bf2(gelu_exact(bf_lo(u.w) * ga[6] + be[6]), gelu_exact(bf_hi(u.w) * ga[7] + be[7]));
      *cell = o;
      if (a.h3) *reinterpret_cast<uint4*>(a.h3 + ((long long)b * N + p) * a.hid + hch) = o;
    }
  }
  __syncthreads();
  MLP_STAMP(6);

  // ---- phase 5: this slab's contribution to fc2: part[slab][b][p][co] = sum_k h3[p][k] * w2[co][64 slab + k]   (fp32).
  // Units (row tile, 32-column tile) dealt to the 16 waves; a lane holds one column and 16 rows: 128-byte runs per store.
  {
    float* pb = a.part + ((long long)slab * a.B + b) * N * Cs;
    const int CT = Cs >> 5;
    for (int u = wv; u < NT * CT; u += MW) {
      const int tile = u / CT, ct = u - tile * CT;
      const int p = tile * 32 + (l & 31), co = ct * 32 + (l & 31);
      f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(bufB + p * SLAB + swz128(p, ks * 2 + (l >> 5)) * 8);
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sW2 + co * SLAB + swz128(co, ks * 2 + (l >> 5)) * 8);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, wf, o, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (row < N) pb[(long long)row * Cs + co] = o[r];
      }
    }
  }
  MLP_STAMP(7);
}

// x2 = x1 + dp[b] * bf16( sum_s part[s] + bias2 ), with the GroupNorm sums of x2 (g16 and per channel) for the next block's
// norm1.  Thread mapping as the GroupNorm kernels: 8 consecutive channels per thread, pixel lanes folded through LDS.
__global__ __launch_bounds__(TPB) void k_mlp_reduce(const float* part, int slabs, long long slab_stride, const float* x1,
                                                    const float* bias2, const float* dp, long long P, int C, int chunk, float* x2,
                                                    crd_sum_t* stats, crd_sum_t* chan) {
  extern __shared__ float sm[];  // [PL][C][2]
  const int b = blockIdx.y;
  const int CG = C >> 3;
  int PL = TPB / CG;
  if (PL < 1) PL = 1;
  const bool active = (int)threadIdx.x < PL * CG;
  const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
  if (active) {
    float bb[8];
    load8t<1>(bias2, cg * 8, bb);
    const float dps = dp ? dp[b] : 1.f;
    long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk;
    if (p1 > P) p1 = P;
    for (long long p = p0 + pl; p < p1; p += PL) {
      const long long off = ((long long)b * P + p) * C + cg * 8;
      float v[8], x[8];
      load8t<1>(x1, off, x);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      for (int s0 = 0; s0 < slabs; s0 += 4) {           // four slabs per pass in flight (clamped index, select past the end)
        float w[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load8t<1>(part, (long long)(s0 + k < slabs ? s0 + k : slabs - 1) * slab_stride + off, w[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = s0 + k < slabs;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += ok ? w[k][j] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x[j] += dps * bf_round(v[j] + bb[j]);
        s[j] += x[j]; ss[j] += x[j] * x[j];
      }
      store8_f32(x2, off, x);
    }
    if (stats) {
      float4* row = reinterpret_cast<float4*>(sm + (long long)pl * 2 * C + cg * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) row[j] = make_float4(s[2 * j], ss[2 * j], s[2 * j + 1], ss[2 * j + 1]);
    }
  }
  if (!stats) return;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += TPB) {
    float v = sm[i];
    for (int r = 1; r < PL; ++r) v += sm[(long long)r * 2 * C + i];
    sm[i] = v;
    if (chan) stat_add(&chan[(long long)b * C * 2 + i], v);
  }
  __syncthreads();
  for (int g = threadIdx.x; g < (C >> 4) * 2; g += TPB) {
    const int slab = g >> 1, which = g & 1;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += sm[(slab * 16 + j) * 2 + which];
    stat_add(&stats[((long long)b * (C >> 4) + slab) * 2 + which], v);
  }
}

size_t mlp_lds_bytes(int N, int Cs) {
  const int NP = (N + 31) / 32 * 32;
  return (size_t)Cs * 8 + (192 + 640) * 4 + (size_t)SLAB * (Cs + 8) * 2 + (size_t)Cs * SLAB * 2 + 2 * (size_t)NP * SLAB * 2;
}

// instantiations: NCH = C / 32 in {1, 2, 4, 5, 8}; XIT = 1 (<= 128 pixels) or 4 (<= 416; with NCH = 8 the x ring would not
// fit in 128 VGPRs next to the accumulators, and no stage of the model needs it)
template <int NCH, int XIT>
int mlp_launch(const MlpK& k, int slabs, size_t lds, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    crd_reserve_lds(reinterpret_cast<const void*>(&k_mlp_fwd<NCH, XIT>), 160 * 1024, "k_mlp_fwd");
    attr_done = true;
  }
  hipLaunchKernelGGL((k_mlp_fwd<NCH, XIT>), dim3(slabs, k.B), dim3(MT), lds, st, k);
  return 0;
}

}  // namespace

#ifdef CRD_MLP_PROF
extern "C" int crd_dbg_mlp_prof(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlp_prof), sizeof(g_mlp_prof)); }
#endif

extern "C" int crd_mlp_fused_supported(int32_t H, int32_t W, int32_t Cs, int32_t hid) {
  const int N = H * W, nch = Cs / 32;
  if (Cs % 32 != 0 || !(nch == 1 || nch == 2 || nch == 4 || nch == 5 || nch == 8) || hid % SLAB != 0 || hid / (Cs / 16) != SLAB) return 0;
  if (N > 416 || (N > 128 && nch == 8) || mlp_lds_bytes(N, Cs) > 160 * 1024) return 0;
  return hid / SLAB;
}

extern "C" int crd_mlp_fwd(const crd_mlp_desc* d, crd_stream_t stream) {
  CRD_CHECK_ARG(d && d->x1 && d->x1_stats && d->norm_gamma && d->norm_beta && d->w_fc1 && d->b_fc1 && d->norm1_gamma && d->norm1_beta &&
                    d->w9 && d->b_dw && d->norm2_gamma && d->norm2_beta && d->w_fc2 && d->h1_stats && d->h2_stats && d->fc2_partials,
                "crd_mlp_fwd: null pointer");
  const int slabs = crd_mlp_fused_supported(d->H, d->W, d->C, d->hidden);
  CRD_UNSUPPORTED(slabs > 0, "crd_mlp_fwd: needs hidden / (C / 16) == 64, C %% 32 == 0, C <= 256 and H * W <= 416 (got %dx%d, C %d, hidden %d)",
                  d->H, d->W, d->C, d->hidden);
  MlpK k;
  k.x1 = d->x1; k.st2 = d->x1_stats; k.g2 = d->norm_gamma; k.b2 = d->norm_beta;
  k.w1 = reinterpret_cast<const bf16_t*>(d->w_fc1); k.bias1 = d->b_fc1; k.n1g = d->norm1_gamma; k.n1b = d->norm1_beta;
  k.w9 = d->w9; k.dwb = d->b_dw; k.n2g = d->norm2_gamma; k.n2b = d->norm2_beta;
  k.w2 = reinterpret_cast<const bf16_t*>(d->w_fc2);
  k.xn2 = reinterpret_cast<bf16_t*>(d->xn); k.h1 = reinterpret_cast<bf16_t*>(d->h1); k.h2 = reinterpret_cast<bf16_t*>(d->h2);
  k.h3 = reinterpret_cast<bf16_t*>(d->h3);
  k.sth1 = d->h1_stats; k.sth2 = d->h2_stats; k.part = d->fc2_partials;
  k.B = d->B; k.H = d->H; k.W = d->W; k.N = d->H * d->W; k.NP = (k.N + 31) / 32 * 32; k.Cs = d->C; k.hid = d->hidden;
  const size_t lds = mlp_lds_bytes(k.N, k.Cs);
  hipStream_t st = as_stream(stream);
  const int nch = k.Cs / 32;
  if (k.N <= 128) {
    switch (nch) {
      case 1: mlp_launch<1, 1>(k, slabs, lds, st); break;
      case 2: mlp_launch<2, 1>(k, slabs, lds, st); break;
      case 4: mlp_launch<4, 1>(k, slabs, lds, st); break;
      case 5: mlp_launch<5, 1>(k, slabs, lds, st); break;
      default: mlp_launch<8, 1>(k, slabs, lds, st); break;
    }
  } else {
    switch (nch) {
      case 1: mlp_launch<1, 4>(k, slabs, lds, st); break;
      case 2: mlp_launch<2, 4>(k, slabs, lds, st); break;
      case 4: mlp_launch<4, 4>(k, slabs, lds, st); break;
      default: mlp_launch<5, 4>(k, slabs, lds, st); break;
    }
  }
  CRD_LAUNCH_CHECK("crd_mlp_fwd");
  return CRD_OK;
}

extern "C" int crd_mlp_reduce(const float* fc2_partials, int32_t slabs, const float* x1, const float* b_fc2, const float* dp,
                              int32_t B, int32_t N, int32_t C, float* x2, crd_sum_t* stats, crd_sum_t* chan_sums, crd_stream_t stream) {
  CRD_CHECK_ARG(fc2_partials && x1 && b_fc2 && x2 && slabs >= 1, "crd_mlp_reduce: null pointer");
  CRD_CHECK_ARG(C % 16 == 0 && C <= 2048 && (stats || !chan_sums), "crd_mlp_reduce: C must be a multiple of 16; chan_sums goes with stats");
  const int CG = C >> 3;
  int PL = TPB / CG;
  if (PL < 1) PL = 1;
  // ~2 pixels per pixel lane on these small grids (as the GroupNorm kernels: fewer pixels per lane rather than idle CUs)
  long long per_block = (long long)PL * 2;
  long long nblk = (N + per_block - 1) / per_block;
  long long cap = 1024 / (B > 0 ? B : 1);
  if (cap < 1) cap = 1;
  if (nblk > cap) nblk = cap;
  const int chunk = (int)((N + nblk - 1) / nblk);
  nblk = (N + chunk - 1) / chunk;
  hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)nblk, B), dim3(TPB), stats ? (size_t)PL * 2 * C * sizeof(float) : 0, as_stream(stream),
                     fc2_partials, slabs, (long long)B * N * C, x1, b_fc2, dp, (long long)N, C, chunk, x2, stats, chan_sums);
  CRD_LAUNCH_CHECK("crd_mlp_reduce");
  return CRD_OK;
}
