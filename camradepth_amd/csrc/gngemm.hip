// GroupNorm-apply folded into the A-operand path of an MFMA GEMM (gfx950, bf16 operands, fp32 accumulate).
//
//   y = epilogue( W * act( GN(x) ) )        x: raw (un-normalised) fp32 or bf16 pixel-major tensor
//
// for the pointwise / non-overlapping-patch convolutions of the encoder blocks (reference:
// src/models/simplified_attention.py:34-43 fc1 / fc2 behind Mlp.norm2 + GELU, :96-100 attn.k behind attn.norm,
// :142-145 q behind Block.norm1 and fc1 behind Block.norm2).  Each of these was a crd_gn_apply launch followed by a
// crd_conv_igemm launch: an extra pass over the tensor and one more link in the encoder's latency-bound launch chain.
//
// Both operands take the classic route -- buffer_load to registers two K-steps ahead, ds_write into a two-stage LDS tile -- and
// the raw rows are normalised in registers on their way:
//   x * scale[c] + shift[c]  (per sample and channel, from the GroupNorm sums the PRODUCER's epilogue left in `stats`; the
//   (scale, shift) table of the sample sits in LDS) -> optional exact GELU -> bf16 -> ds_write_b128 into the XOR-swizzled
//   image the fragment reads expect.
// The normalised tensor is still written once (by the workgroups of column tile 0) when a weight gradient needs it.
// History: the first register-path version was 2-4x slower than the two launches it replaced --
// its __syncthreads() waited for every register load in flight (vmcnt(0)), so each K-slab paid a full memory latency.  Two
// LDS-DMA versions followed (raw tile resident in LDS / streamed through a ring, LDS -> LDS normalise pass): correct, but
// 50-120 KB of LDS per workgroup, a serial walk over column chunks and their drained rings left them 1.1-2.3x slower than the
// pair as well.  This one uses lds_barrier() (common.h) and matches or beats the pair on every shape without GELU.
// The epilogue (bias, residual + DropPath scale, GroupNorm sums of the output, fp32 / bf16 stores) is conv_common.h's.
#include <stdlib.h>
#include "conv_common.h"

using namespace crdk;

namespace {

constexpr int BK = 64;

struct GnIn {
  const void* x; int x_f32;               // raw input [B][IH*IW][x_ld] (+ channel offset applied), fp32 or bf16
  const crd_sum_t* stats; int gmul;       // [B][Cin/16][2] slab sums of x; a group = gmul slabs
  const float* gamma; const float* beta;  // [Cin]
  float count;                            // pixels per sample * channels per group
  bf16_t* xn; int xn_ld; long long xn_bstride;    // optional store of act(GN(x)) (bf16), nullptr = none
};


// (scale, shift) of every input channel of sample b -> tab[Cin]
__device__ __forceinline__ void build_table(const ConvK& a, const GnIn& gi, int b, float2* tab) {
  const crd_sum_t* stb = gi.stats + (long long)b * (a.Cin >> 4) * 2;
  for (int c = threadIdx.x; c < a.Cin; c += 256) {
    float mean, rstd;
    gn_mean_rstd(stb, ((c >> 4) / gi.gmul) * gi.gmul, gi.gmul, gi.count, mean, rstd);
    const float ga = gi.gamma[c] * rstd;
    tab[c] = make_float2(ga, gi.beta[c] - mean * ga);
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

template <int BM, int BN>
constexpr size_t epilogue_bytes() { return (size_t)BM * (BN + 8) * 4 + 256 * 16 * 4 + 2048; }   // fp32 staging tile + folds behind it

// One workgroup per (row tile, column tile) like k_igemm; 32-48 KB of LDS, so 3-4 workgroups per CU hide each other's
// latencies; no counted waits (the compiler tracks register loads exactly, lds_barrier() keeps them in flight across barriers).
template <int WM, int WN, int TM, int TN, int XF32, int ACT>
__device__ __forceinline__ void gngemm_body(const ConvK& a, const GnIn& gi, const int bx, const int by, const int bz) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4r;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sA = lds;                                   // [2][BM][BK]   (the epilogue's staging area aliases the tiles)
  bf16_t* sB = sA + 2 * BM * BK;                      // [2][BN][BK]
  float2* tab = reinterpret_cast<float2*>(reinterpret_cast<char*>(lds) + a.lds_bytes);   // [Cin], behind tiles / staging

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int b = bz, m0 = bx * BM, n0 = by * BN;
  const int r0 = 8 * wv + (l >> 3);                   // this thread's rows: r0 + 32 i; LDS slot l & 7 <- K granule g (k_igemm's swizzle)
  const int g = (l & 7) ^ ((r0 >> 1) & 7);
  const unsigned OOB = 0x80000000u;
  const int esz = XF32 ? 4 : 2;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(reinterpret_cast<const char*>(gi.x) + (long long)b * a.x_bstride * esz), 0, (int)(a.x_bstride * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const int nK = (a.Ktot + BK - 1) / BK;
  const bool store_xn = gi.xn != nullptr && by == 0;
  // the stored copy of the normalised operand goes through a buffer descriptor: rows / granules that are not stored carry an
  // out-of-range offset and the hardware drops them, so the store is UNCONDITIONAL (round 6: under `if (store_xn && ok)` the compiler
  // could no longer count the requests in flight and drained the two-slab prefetch at every use -- see the K loop below)
  const __amdgpu_buffer_rsrc_t rxn = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(store_xn ? gi.xn + (long long)b * gi.xn_bstride : reinterpret_cast<bf16_t*>(const_cast<bf16_t*>(a.w))), 0,
      store_xn ? (int)(gi.xn_bstride * 2) : 0, 0x00020000);

  // per-row constants: pixel index of the row's patch origin, validity; and the loop-invariant BYTE offsets of that origin in x / xn
  // with the validity folded in (an invalid row starts out of range and stays there whatever the K loop adds): every load and store of
  // the K loop is unconditional.  As `ok ? computed : OOB` inside the loop the compiler sank the multiply into an exec-mask branch
  // around the load (two loads into the same registers on two paths, each behind a vmcnt(0)) in the bf16-input variants.
  int rowpix[A_IT];
  bool rowok[A_IT];
  unsigned rowoff[A_IT], xnoff[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + r0 + 32 * i;
    const int oy = m / a.OW, ox = m - oy * a.OW;
    rowok[i] = m < a.OHW;
    rowpix[i] = oy * a.stride * a.IW + ox * a.stride;
    rowoff[i] = rowok[i] ? (unsigned)(rowpix[i] * a.x_ld * esz) : OOB;
    xnoff[i] = (rowok[i] && store_xn) ? (unsigned)(rowpix[i] * gi.xn_ld * 2) : OOB;
  }
  unsigned woff[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int ng = n0 + r0 + 32 * j;
    woff[j] = ng < a.Cout ? (unsigned)(ng * a.Ktot * 2) : OOB;
  }
  struct Regs { u32x4r a[A_IT][XF32 ? 2 : 1]; u32x4r w[B_IT]; };
  auto kpos = [&](int kt, int& kc, int& tappix, bool& kok) {     // this thread's K granule of slab kt: channel, pixel offset of its tap
    const int kf = kt * BK + g * 8;
    kok = kf < a.Ktot;
    kc = kf; tappix = 0;
    if (a.KW > 1) { const int tap = kf / a.Cin; kc = kf - tap * a.Cin; const int ky = tap / a.KW, kx = tap - ky * a.KW; tappix = ky * a.IW + kx; }
  };
  auto load_slab = [&](int kt, Regs& r) {
    int kc, tappix; bool kok;
    kpos(kt, kc, tappix, kok);
    const unsigned km = kok ? 0u : OOB;                  // K tail / slabs past the end: out of range as well
    const unsigned add = (unsigned)((tappix * a.x_ld + kc) * esz);
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const unsigned off = (rowoff[i] + add) | km;
      r.a[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      if (XF32) r.a[i][XF32 ? 1 : 0] = __builtin_amdgcn_raw_buffer_load_b128(rx, off + 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j)
      r.w[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (woff[j] + (unsigned)((kt * BK + g * 8) * 2)) | km, 0, 0);
  };
  auto store_slab = [&](int kt, int stage, const Regs& r) {
    int kc, tappix; bool kok;
    kpos(kt, kc, tappix, kok);
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
    for (int i = 0; i < A_IT; ++i) {
      float v[8];
      if (XF32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = __uint_as_float(r.a[i][0][j]); v[4 + j] = __uint_as_float(r.a[i][XF32 ? 1 : 0][j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = bf_lo(r.a[i][0][j]); v[2 * j + 1] = bf_hi(r.a[i][0][j]); }
      }
      const bool ok = kok && rowok[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float y = v[j] * sc[j] + sh[j];
        if (ACT == 1) y = gelu_exact(y);
        v[j] = ok ? y : 0.f;
      }
      const u32x4r q = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      *reinterpret_cast<u32x4r*>(sA + stage * BM * BK + (r0 + 32 * i) * BK + (l & 7) * 8) = q;
      __builtin_amdgcn_raw_buffer_store_b128(q, rxn, (xnoff[i] + (unsigned)((tappix * gi.xn_ld + kc) * 2)) | (kok ? 0u : OOB), 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) *reinterpret_cast<u32x4r*>(sB + stage * BN * BK + (r0 + 32 * j) * BK + (l & 7) * 8) = r.w[j];
  };

  // everything the first two K-steps need is requested before anything is waited for: the table's inputs first (they return
  // first), then slabs 0 and 1
  if (!(a.dbg & 1)) build_table(a, gi, b, tab);
  Regs r0s, r1s;
  load_slab(0, r0s);
  load_slab(1, r1s);                                  // (slabs past the end: every offset out of range -- zeros, no traffic)
  f32x16 acc[TM][TN];
  init_acc<TM, TN, WN>(a, acc, b, n0, wn, l);
  lds_barrier();                                      // the table
  store_slab(0, 0, r0s);
  load_slab(2, r0s);
  lds_barrier();
  // Steady state, two slabs per trip and NO conditional memory operation in it (round 6).  With `if (kt + 3 < nK) load_slab(...)` --
  // a wave-uniform branch around six loads -- the number of requests in flight depends on the path, and the compiler's wait insertion
  // assumes the path that issued fewer: every use of a slab's registers then also drained the YOUNGER slab's requests
  // (s_waitcnt vmcnt(5..0) in the ISA where vmcnt(12..9) was meant), i.e. the two-slab prefetch was one slab deep.  Slabs past the end
  // carry out-of-range offsets instead (zeros in, nothing out); an odd slab count ends in a single-slab tail.
  int kt = 0;
  for (; kt + 2 <= nK; kt += 2) {
    store_slab(kt + 1, 1, r1s);
    load_slab(kt + 3, r1s);
    mfma_slab<TM, TN, WM, WN>(sA, sB, acc, wm, wn, l);
    lds_barrier();
    store_slab(kt + 2, 0, r0s);
    load_slab(kt + 4, r0s);
    mfma_slab<TM, TN, WM, WN>(sA + BM * BK, sB + BN * BK, acc, wm, wn, l);
    lds_barrier();
  }
  if (kt < nK) {
    mfma_slab<TM, TN, WM, WN>(sA, sB, acc, wm, wn, l);
    lds_barrier();
  }
  if (a.dbg & 4) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(a.y)[0] = 1.f; return; }
  conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, bx, lds,
                                [&](int i, int rr, bool& valid, int& row) { row = m0 + (wm * TM + i) * 32 + rr; valid = row < a.OHW; },
                                [&](int rl, bool& valid, int& row) { row = m0 + rl; valid = row < a.OHW; });
}

template <int WM, int WN, int TM, int TN, int XF32, int ACT>
__global__ __launch_bounds__(256) void k_gngemm_reg(ConvK a, GnIn gi) {
  gngemm_body<WM, WN, TM, TN, XF32, ACT>(a, gi, blockIdx.x, blockIdx.y, blockIdx.z);
}

// TWO independent problems behind the same GroupNorm in ONE launch (round 6: attn.q and the attn.sr patch convolution of a Block both read
// norm1(x) -- simplified_attention.py:96-100 -- and were two dependent-latency-bound launches in a row, the second waiting for the first's
// stored XN; here sr normalises its own rows, and the shorter q problem finishes under it).  Workgroups [0, nb0) take problem 0
// (row tiles fastest), the rest problem 1.
struct Gn2 { ConvK a0, a1; GnIn g0, g1; int nb0, nt0, nt1; };
template <int WM, int WN, int TM, int TN, int XF32, int ACT>
__global__ __launch_bounds__(256) void k_gngemm_reg2(Gn2 p) {
  const int x = blockIdx.x;
  if (x < p.nb0) gngemm_body<WM, WN, TM, TN, XF32, ACT>(p.a0, p.g0, x % p.nt0, x / p.nt0, blockIdx.y);
  else { const int y = x - p.nb0; gngemm_body<WM, WN, TM, TN, XF32, ACT>(p.a1, p.g1, y % p.nt1, y / p.nt1, blockIdx.y); }
}

template <int WM, int WN, int TM, int TN, int XF32, int ACT>
int launch_reg2(const ConvK& k0, const GnIn& g0, const ConvK& k1, const GnIn& g1, int B, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  Gn2 p;
  p.a0 = k0; p.a1 = k1; p.g0 = g0; p.g1 = g1;
  p.nt0 = p.a0.n_tiles = cdiv(k0.OHW, BM); p.nt1 = p.a1.n_tiles = cdiv(k1.OHW, BM);
  p.nb0 = p.nt0 * cdiv(k0.Cout, BN);
  const int nb1 = p.nt1 * cdiv(k1.Cout, BN);
  size_t tiles = (size_t)2 * (BM + BN) * BK * 2;
  if (tiles < epilogue_bytes<BM, BN>()) tiles = epilogue_bytes<BM, BN>();
  tiles = (tiles + 255) / 256 * 256;
  const int cin = k0.Cin > k1.Cin ? k0.Cin : k1.Cin;
  const size_t lds = tiles + (size_t)cin * sizeof(float2);
  CRD_UNSUPPORTED(lds <= 160 * 1024, "crd_gn_conv2: table does not fit in LDS");
  static bool attr_done = false;
  if (!attr_done) {
    crd_reserve_lds(reinterpret_cast<const void*>(&k_gngemm_reg2<WM, WN, TM, TN, XF32, ACT>), 160 * 1024, "k_gngemm_reg2");
    attr_done = true;
  }
  p.a0.lds_bytes = p.a1.lds_bytes = (int)tiles;
  hipLaunchKernelGGL((k_gngemm_reg2<WM, WN, TM, TN, XF32, ACT>), dim3(p.nb0 + nb1, B), dim3(256), lds, st, p);
  CRD_LAUNCH_CHECK("crd_gn_conv2");
  return CRD_OK;
}

template <int WM, int WN, int TM, int TN, int XF32, int ACT>
int launch_reg(const ConvK& k0, GnIn gi, int B, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  ConvK k = k0;
  k.n_tiles = cdiv(k.OHW, BM);
  size_t tiles = (size_t)2 * (BM + BN) * BK * 2;
  if (tiles < epilogue_bytes<BM, BN>()) tiles = epilogue_bytes<BM, BN>();
  tiles = (tiles + 255) / 256 * 256;
  const size_t lds = tiles + (size_t)k.Cin * sizeof(float2);
  CRD_UNSUPPORTED(lds <= 160 * 1024, "crd_gn_conv: table does not fit in LDS");
  static bool attr_done = false;
  if (!attr_done) {
    crd_reserve_lds(reinterpret_cast<const void*>(&k_gngemm_reg<WM, WN, TM, TN, XF32, ACT>), 160 * 1024, "k_gngemm_reg");
    attr_done = true;
  }
  k.lds_bytes = (int)tiles;
  hipLaunchKernelGGL((k_gngemm_reg<WM, WN, TM, TN, XF32, ACT>), dim3(k.n_tiles, cdiv(k.Cout, BN), B), dim3(256), lds, st, k, gi);
  CRD_LAUNCH_CHECK("crd_gn_conv");
  return CRD_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Wide pointwise layers (Mlp.fc1 behind Block.norm2: K = C = 64 / 128 / 160 channels in, N = hidden = 512 / 1024 / 640 out;
// src/models/simplified_attention.py:34-37).  These launches are ALL output: 54.5 / 27 MB of bf16 for 6.8 / 3.4 MB in, and the
// tile-per-workgroup kernel above wrote them at 1.2-1.8 TB/s -- 37 / 25 us where a fill of the same bytes takes 10 / 6 us
// (tools/bench_write.py, tools/sweep_igemm.sh: with the K loop compiled out 33 / 21 us remain: per-workgroup prologue, LDS-staged
// epilogue with two barriers, 16 statistics atomics per 64 x 128 tile, 3-5 workgroups per CU).  Here instead:
//   * a workgroup owns NB = 128 or 256 output columns of ONE sample and walks over 64-row tiles of it; its weights (NB x K) are
//     loaded ONCE, straight into MFMA A-operand registers (rows = output channels) -- no LDS, no re-reads;
//   * the raw fp32 rows of the NEXT tile are in flight (registers) while the current one is multiplied and stored; they are
//     normalised on their way into a double-buffered bf16 LDS tile (row stride 4 x odd dwords: conflict-free 16-byte B reads);
//   * a wave turns its 32-pixel x 32 WCT-column strip through a private piece of LDS (no workgroup barrier) and stores whole
//     128-byte rows: 8 lanes per row, eight full lines per wave instruction;
//   * GroupNorm sums of the output stay in registers across the workgroup's tiles and leave as 8-16 atomics per wave at the end.
// XF = 1: the fp32 rows behind a GroupNorm (fc1); XF = 0: plain bf16 rows (fc2's data gradient at stages 1-2: d(h3) = W2^T d(x2), the
// same shape transposed -- crd_conv_igemm sends it here).
// RED (XF = 0, one column tile per wave): the reduce phase of the backward of the GroupNorm (+ GELU) whose dy this launch produces
// (Mlp.norm2 behind fc2's data gradient; crd_conv_desc.red_*): every lane keeps (sum g, sum g xhat) of its 16 channels in registers
// across the workgroup's tiles -- g = dy x GELU'(xhat gamma + beta) from the ROUNDED output, as crd_gn_bwd_reduce reads it back -- and
// they leave once per workgroup: a half-wave fold, one atomic pair per channel, one per group for the gamma-weighted sums.
template <int KS, int WCT, int XF, bool RED = false>
__global__ __launch_bounds__(256, 2) void k_gn_pw_wide(ConvK a, GnIn gi, int ncb, int R) {
  static_assert(!RED || (XF == 0 && WCT == 1), "fused reduce: plain rows, 128-column workgroups");
  constexpr int K = KS * 16, LDA = K + 8, G4 = K / 4, NB = 4 * WCT * 32, BM = 64;
  constexpr int G8 = K / 8, NX = XF ? KS : (KS + 1) / 2;          // 16-byte granules per row / per thread
  constexpr int LDT = WCT * 32 + 8;                                // per-wave transposition strip [32 pixels][LDT] (rows of 16 x odd bytes)
  constexpr int GPRW = WCT * 4, RPP = 64 / GPRW;                   // 16-byte granules per strip row; strip rows per store instruction
  typedef __attribute__((ext_vector_type(4))) float f32x4t;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2t;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4r;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sA = lds;                                              // [2][BM][LDA]
  float2* tab = reinterpret_cast<float2*>(lds + 2 * BM * LDA);    // [K]
  bf16_t* sT = reinterpret_cast<bf16_t*>(tab + K);                 // [4 waves][32][LDT]
  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cb = blockIdx.x % ncb, rest = blockIdx.x / ncb;
  const int b = rest / R, s0 = rest - b * R;
  const int P = a.OHW, nT = (P + BM - 1) / BM;
  const int n0 = cb * NB + wv * WCT * 32;
  const bool store_xn = gi.xn != nullptr && cb == 0;

  // ---- everything the first tile needs is requested before anything is waited for: weights, the table's inputs, tile 0
  bf16x8 wf[WCT][KS];
#pragma unroll
  for (int j = 0; j < WCT; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[j][ks] = *reinterpret_cast<const bf16x8*>(a.w + (long long)(n0 + j * 32 + (l & 31)) * K + ks * 16 + (l >> 5) * 8);
  const float* xb = reinterpret_cast<const float*>(gi.x) + (long long)b * a.x_bstride;
  const bf16_t* xh = reinterpret_cast<const bf16_t*>(gi.x) + (long long)b * a.x_bstride;
  constexpr bool D2 = KS <= 8;                         // two tiles ahead (K = 160: 80 more registers would spill; its workgroups have 1-2 tiles)
  f32x4t xr0[NX], xr1[D2 ? NX : 1];                    // rows of the tiles one and two steps ahead (in flight)
  auto load_tile = [&](int tile, f32x4t (&xr)[NX]) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (XF) {
        const int id = t + 256 * i, row = id / G4, g4 = id - row * G4;
        int p = tile * BM + row;
        p = p < P ? p : P - 1;
        xr[i] = *reinterpret_cast<const f32x4t*>(xb + (long long)p * a.x_ld + g4 * 4);
      } else {
        int id = t + 256 * i;
        id = id < BM * G8 ? id : BM * G8 - 1;           // (K = 160: 1280 granules on 5 x 256 threads exactly; K = 64 / 128 too)
        const int row = id / G8, g8 = id - row * G8;
        int p = tile * BM + row;
        p = p < P ? p : P - 1;
        xr[i] = *reinterpret_cast<const f32x4t*>(xh + (long long)p * a.x_ld + g8 * 8);
      }
    }
  };
  auto store_tile = [&](int tile, int buf, const f32x4t (&xr)[NX]) {          // normalise -> bf16 -> LDS (and the stored copy the weight gradient reads)
    if (!XF) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int id = t + 256 * i;
        if (id < BM * G8) { const int row = id / G8, g8 = id - row * G8; *reinterpret_cast<f32x4t*>(sA + (buf * BM + row) * LDA + g8 * 8) = xr[i]; }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int id = t + 256 * i, row = id / G4, g4 = id - row * G4;
      const f32x4t t0 = *reinterpret_cast<const f32x4t*>(tab + g4 * 4), t1 = *reinterpret_cast<const f32x4t*>(tab + g4 * 4 + 2);
      const f32x4t v = xr[i];
      u32x2t q;
      q[0] = pack_bf2(v[0] * t0[0] + t0[1], v[1] * t0[2] + t0[3]);
      q[1] = pack_bf2(v[2] * t1[0] + t1[1], v[3] * t1[2] + t1[3]);
      *reinterpret_cast<u32x2t*>(sA + (buf * BM + row) * LDA + g4 * 4) = q;
      const int p = tile * BM + row;
      if (store_xn && p < P) *reinterpret_cast<u32x2t*>(gi.xn + (long long)b * gi.xn_bstride + (long long)p * gi.xn_ld + g4 * 4) = q;
    }
  };
  // invariant at the head of step(tile, P): sA[P] holds `tile`, xr(P^1) (in flight) tile + R, xr(P) tile + 2R.  One tile ahead was
  // not enough: a tile's MFMAs and stores take ~1 us, a first-touch row ~3 (24.7 us per launch at 64 -> 512, 64 x 104 x 8)
  if (s0 < nT) load_tile(s0, xr0);
  if constexpr (D2) { if (s0 + R < nT) load_tile(s0 + R, xr1); }
  if (XF) { build_table(a, gi, b, tab); lds_barrier(); }          // (the table only: the weights and rows in flight are not waited for here)
  if (s0 >= nT) return;
  store_tile(s0, 0, xr0);
  if (D2 && s0 + 2 * R < nT) load_tile(s0 + 2 * R, xr0);
  lds_barrier();

  float st_s[WCT][2], st_q[WCT][2];
#pragma unroll
  for (int j = 0; j < WCT; ++j) st_s[j][0] = st_s[j][1] = st_q[j][0] = st_q[j][1] = 0.f;
  int le = l;
  asm volatile("" : "+v"(le));
  const int half = le >> 5, px = le & 31;
  // the bias of this lane's 16 channels per column tile, once (as loads inside the epilogue they were waited for one by one with
  // vmcnt(0), which also drained the rows in flight and the tile's stores): the accumulators start from it
  f32x4t bvr[WCT][4];
#pragma unroll
  for (int j = 0; j < WCT; ++j)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      bvr[j][g4] = f32x4t{0.f, 0.f, 0.f, 0.f};
      if (a.bias) bvr[j][g4] = *reinterpret_cast<const f32x4t*>(a.bias + (long long)b * a.bias_bstride + n0 + j * 32 + 8 * g4 + 4 * half);
    }
  // fused reduce: gamma / beta of the lane's 16 channels, the moments of their group, the running sums
  f32x4t rga[RED ? 4 : 1], rbe[RED ? 4 : 1], rs0[RED ? 4 : 1], rs1[RED ? 4 : 1];
  float rmean = 0.f, rrstd = 0.f;
  const bf16_t* rxb = nullptr;
  if constexpr (RED) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      rga[g4] = *reinterpret_cast<const f32x4t*>(a.red_gamma + n0 + 8 * g4 + 4 * half);
      rbe[g4] = *reinterpret_cast<const f32x4t*>(a.red_beta + n0 + 8 * g4 + 4 * half);
      rs0[g4] = f32x4t{0.f, 0.f, 0.f, 0.f};
      rs1[g4] = f32x4t{0.f, 0.f, 0.f, 0.f};
    }
    const int cpg = 16 * a.red_gmul;
    gn_mean_rstd(a.red_stats + (long long)b * (a.Cout >> 4) * 2, (n0 / cpg) * a.red_gmul, a.red_gmul, (float)P * cpg, rmean, rrstd);
    rxb = reinterpret_cast<const bf16_t*>(a.red_x) + (long long)b * a.red_x_bstride + n0 + 4 * half;
  }
  bf16_t* yb = reinterpret_cast<bf16_t*>(a.y) + (long long)b * a.y_bstride;

  auto step = [&](int tile, int buf, f32x4t (&xnext)[NX]) __attribute__((always_inline)) {
    const int nxt = tile + R;
    if (!D2 && nxt < nT) load_tile(nxt, xnext);
    u32x2t rxv[RED ? 2 : 1][RED ? 4 : 1];              // the GroupNorm's raw input at this lane's pixels and channels (requested now)
    if constexpr (RED) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        int p = tile * BM + rt * 32 + px;
        p = p < P ? p : P - 1;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) rxv[rt][g4] = *reinterpret_cast<const u32x2t*>(rxb + (long long)p * a.red_x_ld + 8 * g4);
      }
    }
    f32x16 acc[2][WCT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < WCT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][j][r] = bvr[j][r >> 2][r & 3];
    const bf16_t* sa = sA + buf * BM * LDA;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(sa + (rt * 32 + (l & 31)) * LDA + ks * 16 + (l >> 5) * 8);
#pragma unroll
        for (int j = 0; j < WCT; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][ks], bfr, acc[rt][j], 0, 0, 0);
      }
    // ---- epilogue.  A lane holds pixel px of row tile rt and the channels (r&3) + 8 (r>>2) + 4 half of each 32-column tile: stored
    // from there, a wave instruction wrote 32-byte pieces of 32 different rows, four instructions per 128-byte line, and the
    // launch stayed at 2.2 TB/s (64 -> 512 at 64 x 104 x 8: 27.8 us).  So the wave turns its 32 x (32 WCT) strip through a PRIVATE
    // piece of LDS (no workgroup barrier: a wave's LDS operations execute in order) and stores whole rows: 8 lanes = 128 bytes.
    bf16_t* T = sT + wv * 32 * LDT;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const bool pok = tile * BM + rt * 32 + px < P;
#pragma unroll
      for (int j = 0; j < WCT; ++j) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          u32x2t d;
          d[0] = pack_bf2(acc[rt][j][4 * g4], acc[rt][j][4 * g4 + 1]);
          d[1] = pack_bf2(acc[rt][j][4 * g4 + 2], acc[rt][j][4 * g4 + 3]);
          if (pok) {
            const float v0 = bf_lo(d[0]), v1 = bf_hi(d[0]), v2 = bf_lo(d[1]), v3 = bf_hi(d[1]);
            st_s[j][g4 >> 1] += (v0 + v1) + (v2 + v3);
            st_q[j][g4 >> 1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            if constexpr (RED) {
              const float dq[4] = {v0, v1, v2, v3};
              const float xq[4] = {bf_lo(rxv[rt][g4][0]), bf_hi(rxv[rt][g4][0]), bf_lo(rxv[rt][g4][1]), bf_hi(rxv[rt][g4][1])};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float xh = (xq[e] - rmean) * rrstd;
                float gg = dq[e];
                if (a.red_act == 1) gg *= gelu_grad(xh * rga[g4][e] + rbe[g4][e]);
                rs0[g4][e] += gg;
                rs1[g4][e] += gg * xh;
              }
            }
          }
          *reinterpret_cast<u32x2t*>(T + px * LDT + j * 32 + 8 * g4 + 4 * half) = d;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 32 / RPP; ++it) {
        const int r = it * RPP + le / GPRW, gq = le % GPRW;
        const int p = tile * BM + rt * 32 + r;
        // (ext_vector_type accesses: through HIP's uint4 struct the compiler split the LDS read into dwords and put vmcnt(0) in front)
        const u32x4r u = *reinterpret_cast<const u32x4r*>(__builtin_assume_aligned(T + r * LDT + gq * 8, 16));
        if (p < P && (!(a.dbg & 32) || u[0] == 0x12345u))
          *reinterpret_cast<u32x4r*>(__builtin_assume_aligned(yb + (long long)p * a.y_ld + n0 + gq * 8, 16)) = u;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (nxt < nT) store_tile(nxt, buf ^ 1, xnext);
    if (D2 && nxt + 2 * R < nT) load_tile(nxt + 2 * R, xnext);
    lds_barrier();          // (NOT __syncthreads(): its vmcnt(0) waits for the rows just requested and for this tile's stores -- 2 us per tile)
  };
  for (int tile = s0; tile < nT; tile += 2 * R) {
    if constexpr (D2) step(tile, 0, xr1); else step(tile, 0, xr0);
    if (tile + R < nT) step(tile + R, 1, xr0);
  }
  if constexpr (RED) {
    // lanes 0-31 / 32-63 hold the same channels for 32 different pixels: fold each half-wave, then lane 0 / 32 adds its 16 channels
    float w[2][2] = {{0.f, 0.f}, {0.f, 0.f}};        // gamma-weighted sums of the wave's two 16-channel slabs: [slab][moment]
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v0 = rs0[g4][e], v1 = rs1[g4][e];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
        if ((l & 31) == 0) {
          const int c = n0 + 8 * g4 + 4 * half + e;
          grad_add(&a.red_r[((long long)b * a.Cout + c) * 2], v0);
          grad_add(&a.red_r[((long long)b * a.Cout + c) * 2 + 1], v1);
        }
        w[g4 >> 1][0] += v0 * rga[g4][e];
        w[g4 >> 1][1] += v1 * rga[g4][e];
      }
    const int Bn = (int)gridDim.x / (ncb * R), cpg = 16 * a.red_gmul;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float tot = w[sl][m] + __shfl_xor(w[sl][m], 32);          // lane 0 (channels + 0) and lane 32 (channels + 4)
        if (l == 0) grad_add(&a.red_r[(long long)Bn * a.Cout * 2 + ((long long)b * (a.Cout / cpg) + (n0 + 16 * sl) / cpg) * 2 + m], tot);
      }
  }
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < WCT; ++j)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const float sv = wave_sum(st_s[j][sl]), sq = wave_sum(st_q[j][sl]);
        if (l == 0) {
          crd_sum_t* dst = a.stats + ((long long)b * a.G16 + ((n0 + j * 32) >> 4) + sl) * 2;
          stat_add(dst, sv);
          stat_add(dst + 1, sq);
        }
      }
  }
}

// Does the wide kernel take this launch?  (Mlp.fc1 of every encoder stage at the benchmark sizes.)
bool pw_wide_applies(const ConvK& k, const GnIn& gi, int act_in) {
  static int on = -1;
  if (on < 0) on = crd_dev_int("CRD_PW_WIDE", 1);
  const bool shape = k.KW == 1 && k.stride == 1 && (k.Cin == 64 || k.Cin == 128 || k.Cin == 160) && k.Ktot == k.Cin && k.Cout >= 256 &&
                     k.Cout % 128 == 0 && k.x_ld % 4 == 0;
  return on && shape && gi.x_f32 && act_in == 0 && !k.y_f32 && !k.res && !k.act && !k.accumulate && !k.chan && k.vec_ok &&
         (k.y_ld & 7) == 0 && (!k.bias || ((reinterpret_cast<uintptr_t>(k.bias) & 15) == 0 && k.bias_bstride % 4 == 0)) &&
         (!gi.xn || gi.xn_ld % 4 == 0) && (reinterpret_cast<uintptr_t>(k.w) & 15) == 0;
}

template <int KS, int WCT, int XF, bool RED = false>
int launch_pw_wide(const ConvK& k, const GnIn& gi, int B, hipStream_t st) {
  constexpr int K = KS * 16, NB = 4 * WCT * 32;
  const int ncb = k.Cout / NB, nT = cdiv(k.OHW, 64);
  const int wg_per_cu = crd_dev_int("CRD_PW_OCC", 2);
  int rmax = 256 * wg_per_cu / (ncb * B);
  if (rmax < 1) rmax = 1;
  const int tpw = cdiv(nT, rmax);                   // tiles per workgroup, then as few streams as that needs (balanced)
  const int R = cdiv(nT, tpw);
  const size_t lds = (size_t)2 * 64 * (K + 8) * 2 + (size_t)K * sizeof(float2) + (size_t)4 * 32 * (WCT * 32 + 8) * 2;
  hipLaunchKernelGGL((k_gn_pw_wide<KS, WCT, XF, RED>), dim3(ncb * R * B), dim3(256), lds, st, k, gi, ncb, R);
  CRD_LAUNCH_CHECK("crd_gn_conv(wide pointwise)");
  return CRD_OK;
}

template <int XF>
int dispatch_pw_wide(const ConvK& k, const GnIn& gi, int B, hipStream_t st) {
  // 128 columns per workgroup (256 measured 10 % slower: 27.2 vs 24.7 us at 64 -> 512 -- fewer, fatter workgroups hide less latency)
  const bool w2 = k.Cout % 256 == 0 && crd_dev_int("CRD_PW_WCT2", 0);
  if (k.Cin == 64) return w2 ? launch_pw_wide<4, 2, XF>(k, gi, B, st) : launch_pw_wide<4, 1, XF>(k, gi, B, st);
  if (k.Cin == 128) return w2 ? launch_pw_wide<8, 2, XF>(k, gi, B, st) : launch_pw_wide<8, 1, XF>(k, gi, B, st);
  return launch_pw_wide<10, 1, XF>(k, gi, B, st);        // (K = 160 with 256 columns would not fit the register file: 80 weight + 64 accumulator + 40 prefetch registers)
}

template <int XF32, int ACT>
int dispatch(const ConvK& k, const GnIn& gi, int B, hipStream_t st) {
  // 64 x 64 tiles when 64 x 128 ones would not cover the chip (as crd_conv_igemm chooses)
  const long long big_tiles = (long long)cdiv(k.OHW, 64) * cdiv(k.Cout, 128) * B;
  if (k.Cout <= 64 || big_tiles < 256) return launch_reg<2, 2, 1, 1, XF32, ACT>(k, gi, B, st);
  return launch_reg<2, 2, 1, 2, XF32, ACT>(k, gi, B, st);
}

}  // namespace

// crd_conv_igemm's wide pointwise launches without a GroupNorm in front (igemm.hip): plain bf16 rows in, bf16 out, optional sums
bool crd_pw_wide_plain_applicable(const ConvK& k) {
  static int on = -1;
  if (on < 0) on = crd_dev_int("CRD_PW_WIDE", 1);
  return on && k.KW == 1 && k.stride == 1 && k.pad == 0 && (k.Cin == 64 || k.Cin == 128 || k.Cin == 160) && k.Ktot == k.Cin && k.Cout >= 256 &&
         k.Cout % 128 == 0 && (k.x_ld & 7) == 0 && (reinterpret_cast<uintptr_t>(k.x) & 15) == 0 && k.IH * k.IW == k.OHW && !k.y_f32 && !k.res &&
         !k.act && !k.accumulate && !k.chan && !k.stats_partial && k.out_mode == 0 && k.vec_ok && (k.y_ld & 7) == 0 &&
         (!k.red_x || (crd_dev_int("CRD_PW_WIDE_RED", 0) && !k.red_x_f32 && (k.red_x_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(k.red_x) & 7) == 0 && k.red_gmul >= 2 &&
                       (reinterpret_cast<uintptr_t>(k.red_gamma) & 15) == 0 && (reinterpret_cast<uintptr_t>(k.red_beta) & 15) == 0)) &&
         (!k.bias || ((reinterpret_cast<uintptr_t>(k.bias) & 15) == 0 && k.bias_bstride % 4 == 0)) && (reinterpret_cast<uintptr_t>(k.w) & 15) == 0;
}
int crd_pw_wide_plain(const ConvK& k, int B, hipStream_t st) {
  GnIn gi;
  gi.x = k.x; gi.x_f32 = 0; gi.stats = nullptr; gi.gmul = 1; gi.gamma = nullptr; gi.beta = nullptr; gi.count = 1.f;
  gi.xn = nullptr; gi.xn_ld = 0; gi.xn_bstride = 0;
  if (k.red_x) {          // with the reduce phase of the following GroupNorm's backward in the epilogue (128-column workgroups)
                          // (correct -- tests/test_gpu_igemm.py runs it in a developer build -- but SLOWER in the step: 18.01 vs 17.91 ms;
                          //  the GELU' of 27-54 M elements costs more in this epilogue than the streaming reduce kernel it replaces)
    if (k.Cin == 64) return launch_pw_wide<4, 1, 0, true>(k, gi, B, st);
    if (k.Cin == 128) return launch_pw_wide<8, 1, 0, true>(k, gi, B, st);
    return launch_pw_wide<10, 1, 0, true>(k, gi, B, st);
  }
  return dispatch_pw_wide<0>(k, gi, B, st);
}

static int gn_conv_args(const crd_conv_desc* d, const crd_gn_input* n, ConvK& k, GnIn& gi) {
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
  k.red_x = nullptr; k.red_x_f32 = 0; k.red_x_ld = 0; k.red_x_bstride = 0; k.red_stats = nullptr; k.red_gamma = nullptr; k.red_beta = nullptr;
  k.red_gmul = 1; k.red_act = 0; k.red_r = nullptr;
  { static int dbg = -1; if (dbg < 0) dbg = crd_dev_int("CRD_DBG", 0); k.dbg = dbg; }
  gi.x_f32 = n->x_f32;
  gi.x = n->x_f32 ? (const void*)(reinterpret_cast<const float*>(d->x) + d->x_coff) : (const void*)(reinterpret_cast<const bf16_t*>(d->x) + d->x_coff);
  CRD_CHECK_ARG((reinterpret_cast<uintptr_t>(gi.x) & 15) == 0 && (!n->x_f32 || d->x_ld % 4 == 0), "crd_gn_conv: x rows must be 16-byte aligned");
  gi.stats = n->stats; gi.gmul = n->gmul; gi.gamma = n->gamma; gi.beta = n->beta;
  gi.count = (float)d->IH * (float)d->IW * 16.f * (float)n->gmul;
  gi.xn = reinterpret_cast<bf16_t*>(n->xn); gi.xn_ld = n->xn_ld; gi.xn_bstride = (long long)d->IH * d->IW * n->xn_ld;
  return CRD_OK;
}

extern "C" int crd_gn_conv(const crd_conv_desc* d, const crd_gn_input* n, crd_stream_t stream) {
  ConvK k;
  GnIn gi;
  { const int rc = gn_conv_args(d, n, k, gi); if (rc != CRD_OK) return rc; }
  hipStream_t st = as_stream(stream);
  if (n->act == 1 && !n->x_f32 && d->KH == 1) {        // Mlp.norm2 + GELU in front of fc2 at stages 1-2: the narrow streaming kernel
    ConvK kn = k;
    kn.x = reinterpret_cast<const bf16_t*>(gi.x);
    NarrowGn ng;
    ng.stats = gi.stats; ng.gmul = gi.gmul; ng.gamma = gi.gamma; ng.beta = gi.beta; ng.count = gi.count;
    ng.xn = gi.xn; ng.xn_ld = gi.xn_ld; ng.xn_bstride = gi.xn_bstride;
    if (crd_pw_narrow_applicable(kn, &ng)) return crd_pw_narrow(kn, &ng, d->B, st);
  }
  if (pw_wide_applies(k, gi, n->act)) return dispatch_pw_wide<1>(k, gi, d->B, st);
  if (n->x_f32) return n->act ? dispatch<1, 1>(k, gi, d->B, st) : dispatch<1, 0>(k, gi, d->B, st);
  return n->act ? dispatch<0, 1>(k, gi, d->B, st) : dispatch<0, 0>(k, gi, d->B, st);
}

extern "C" int crd_gn_conv2(const crd_conv_desc* d0, const crd_gn_input* n0, const crd_conv_desc* d1, const crd_gn_input* n1, crd_stream_t stream) {
  ConvK k0, k1;
  GnIn g0, g1;
  { const int rc = gn_conv_args(d0, n0, k0, g0); if (rc != CRD_OK) return rc; }
  { const int rc = gn_conv_args(d1, n1, k1, g1); if (rc != CRD_OK) return rc; }
  CRD_UNSUPPORTED(d0->B == d1->B && n0->x_f32 == 1 && n1->x_f32 == 1 && n0->act == 0 && n1->act == 0,
                  "crd_gn_conv2: two problems of one batch behind a GroupNorm of the fp32 residual stream, no activation");
  // both on the 64 x 64 tiles (what crd_gn_conv picks for each of them alone at the encoder's sizes; a wide problem takes its own launch)
  auto small = [&](const ConvK& k) { return k.Cout <= 64 || (long long)cdiv(k.OHW, 64) * cdiv(k.Cout, 128) * d0->B < 256; };
  CRD_UNSUPPORTED(small(k0) && small(k1), "crd_gn_conv2: both problems must take the 64 x 64 tiles");
  return launch_reg2<2, 2, 1, 1, 1, 0>(k0, g0, k1, g1, d0->B, as_stream(stream));
}
