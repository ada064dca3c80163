// fp8 (OCP e4m3) forward of the decoder's 3x3 ConvLayers on the block-scaled MFMA of gfx950
// (v_mfma_scale_f32_32x32x64_f8f6f4, all block scales 2^0: plain fp8 at twice the bf16 matrix rate) -- BASELINE.json
// config 5.  Same persistent one-wave-per-SIMD structure as k_conv3x3p (conv3x3p.hip): 16 x 32-pixel tiles, halo of a
// channel chunk in LDS once, weight slab of every (chunk, tap) through an LDS-DMA ring, epilogue straight from the
// accumulators with v_permlane32_swap.  What changes:
//   * activations and weights are 1 byte per channel, so a 64-byte LDS row holds 64 channels: ONE MFMA (K = 64) per
//     32 x 32 tile and step where the bf16 kernel needs four (K = 16) -- the DMA, LDS and barrier cost of a step is the
//     same, the arithmetic behind it doubles;
//   * a lane's MFMA operand is 32 contiguous bytes of its row (k = 32 (l>>5) .. + 31): two ds_read_b128; the whole
//     fragment set of step n+1 is read while the MFMAs of step n issue;
//   * software scales: y = acc * x_scale * w_scale[cout] (per-tensor activation scale from calibration, per-output-channel
//     weight scale from the weights), then the usual bf16 store + GroupNorm sums.
// Round 5: MODE 1 is the DATA GRADIENT of the same convolution (mirrored taps, e4m3 dy with a device-resident scale, weights
// quantised per input channel, accumulate epilogue) -- crd_conv3x3_fp8_dgrad.
#include "conv_common.h"
#include <cstdlib>

using namespace crdk;

namespace {

constexpr int TH = 16, TW = 32, HW_ = TW + 2, HROWS = (TH + 2) * HW_;   // as conv3x3p.hip
constexpr int QKC = 64;                    // channels per chunk = bytes per LDS row
constexpr int NW = 4, TM = 4;
constexpr int HG = (HROWS + 15) / 16, HPAD = HG * 16, HT = (HG + NW - 1) / NW;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// -DCRD_CONV3_ABLATE=bits: parts compiled out for timing experiments (results are wrong): 256 no DMA, 16 no vmcnt wait,
// 32 no barrier, 64 no fragment reads, 128 no MFMA, 512 no epilogue stores
#ifdef CRD_CONV3_ABLATE
#define ABL(bit) ((CRD_CONV3_ABLATE) & (bit))
#else
#define ABL(bit) 0
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
  if (ABL(16)) return; asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct F8K {
  const unsigned char* x; int x_ld; long long x_bstride;     // fp8 activations [B][H*W][x_ld]
  const unsigned char* w; int Cout, Cin, Ktot;               // fp8 weights [Cout][9][Cin]
  const float* w_scale; float x_scale;
  int H, W;
  bf16_t* y; int y_ld; long long y_bstride;
  float* stats_partial; int G16;                              // [B][tiles][NW][G16][2] or nullptr
  const float* x_scale_dev;                                   // non-null: the activation scale lives in device memory (fp8 gradients)
  int accumulate;                                             // y += v (bf16 read-modify-write, as k_conv3x3p)
};

// MODE 0: forward; 1: data gradient (taps mirrored: dx[iy,ix] += w[ky,kx] dy[iy+1-ky, ix+1-kx], weights [Cin][tap][Cout])
template <int TN, int WS, int MODE>
__global__ __launch_bounds__(256, 1) void k_conv3x3_fp8(F8K a, int tiles_x, int tiles_y, int tiles_total) {
  constexpr int D = WS - 1;
  static_assert(D >= 3 && D <= 8, "slab prefetch distance");
  constexpr int BN = TN * 32;
  constexpr int WGROUPS = BN / 16, WJ = (WGROUPS + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
  unsigned char* sH = lds8;                          // [2][HPAD][64]
  unsigned char* sW = sH + 2 * HPAD * QKC;           // [WS][BN][64]
  unsigned char* sD = sW + WS * BN * QKC;            // [16][64] dummy landing area
  float* sS = reinterpret_cast<float*>(sD + 16 * QKC);   // [BN] output scales x_scale * w_scale[n0 + c]

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = blockIdx.y * BN;
  const int H = a.H, W = a.W, Cin = a.Cin;
  const int nChunks = (Cin + QKC - 1) / QKC;
  const unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot, 0x00020000);
  const float xs = a.x_scale_dev ? *a.x_scale_dev : a.x_scale;
  for (int c = t; c < BN; c += 256) sS[c] = n0 + c < a.Cout ? xs * a.w_scale[n0 + c] : 0.f;

  const int wch = ((l & 3) ^ ((l >> 4) & 3)) * 16;            // channel of this lane's 16-byte granule (k_conv3x3p's swizzle)
  unsigned wvo[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int g = NW * j + wv;
    const int n = 16 * g + (l >> 2), ng = n0 + n;
    wvo[j] = (g < WGROUPS && ng < a.Cout) ? (unsigned)(ng * a.Ktot + wch) : OOB;
  }
  auto stage_weights = [&](int chunk, int tap, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool lane_ok = wch < Cin - chunk * QKC;
    if (ABL(256)) return;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int g = NW * j + wv;
      unsigned char* dst = g < WGROUPS ? sW + slot * BN * QKC + 16 * g * QKC : sD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, lane_ok ? wvo[j] : OOB, tap * Cin + chunk * QKC, 0, 0);
    }
#else
    (void)chunk; (void)tap; (void)slot;
#endif
  };
  unsigned hvo[HT];
  auto halo_offsets = [&](int tile) {
    const int bb = tile / (tiles_x * tiles_y), rem = tile - bb * (tiles_x * tiles_y);
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    int lh = l;                                         // opaque: otherwise the per-lane (hy, hx) of every s are hoisted out of
    asm volatile("" : "+v"(lh));                        // the tile loop, spilled, and each reload waits vmcnt(0) behind the ring
#pragma unroll
    for (int s = 0; s < HT; ++s) {
      const int G = NW * s + wv;
      const int hr = 16 * G + (lh >> 2);
      const int hy = hr / HW_, hx = hr - hy * HW_;
      const int iy = tyi * TH - 1 + hy, ix = txi * TW - 1 + hx;
      const bool ok = tile < tiles_total && G < HG && hr < HROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int ch = ((lh & 3) ^ ((hr >> 2) & 3)) * 16;
      hvo[s] = ok ? (unsigned)((iy * W + ix) * a.x_ld + ch) : OOB;
    }
    return bb;
  };
  auto stage_halo = [&](const __amdgpu_buffer_rsrc_t& rx, int chunk, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tail = Cin - chunk * QKC;
    if (ABL(256)) return;
#pragma unroll
    for (int s = 0; s < HT; ++s) {
      const int G = NW * s + wv;
      const int ch = ((l & 3) ^ (((16 * G + (l >> 2)) >> 2) & 3)) * 16;
      const bool real = G < HG;
      unsigned char* dst = real ? sH + buf * HPAD * QKC + G * 16 * QKC : sD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)dst, 16, (real && ch < tail) ? hvo[s] : OOB, chunk * QKC, 0, 0);
    }
#else
    (void)rx; (void)chunk; (void)buf;
#endif
  };
  auto make_rx = [&](int bb) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)bb * a.x_bstride), 0, (int)a.x_bstride, 0x00020000);
  };
  // a lane's operand: 32 bytes of its row.  WHICH 32 is free as long as activations and weights agree (a permutation of k):
  // granules h and 2 + h (h = l >> 5), i.e. per read the two half-waves take adjacent granules exactly as the bf16
  // kernel's two k-steps do -- its conflict-free (row >> 2) & 3 swizzle carries over.
  // (ext_vector_type loads, not HIP's uint4 struct: behind a struct load the compiler cannot rule out aliasing with the
  // LDS-DMA writes in flight and puts s_waitcnt vmcnt(0) in front of every step's reads -- measured 0.56 -> ms)
  auto read_row = [&](const unsigned char* base, int row) {
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const int h = l >> 5, sw = (row >> 2) & 3;
    const i32x4 lo = *reinterpret_cast<const i32x4*>(base + row * QKC + ((h ^ sw) << 4));
    const i32x4 hi = *reinterpret_cast<const i32x4*>(base + row * QKC + (((2 + h) ^ sw) << 4));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // fragment reads of (halo buffer hb, tap, ring slot wb): activation rows i0 .. i1-1 and, with B, the weight rows
  auto read_a = [&](int hb, int tap, int i0, int i1, i32x8 (&af)[TM]) {
    if (ABL(64)) return;
    const int ky_ = tap / 3, kx_ = tap - ky_ * 3;
    const int ky = MODE == 0 ? ky_ : 2 - ky_, kx = MODE == 0 ? kx_ : 2 - kx_;
    const unsigned char* hbase = sH + hb * HPAD * QKC;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      if (i >= i0 && i < i1) af[i] = read_row(hbase, (wv * TM + i + ky) * HW_ + (l & 31) + kx);
  };
  auto read_b = [&](int wb, i32x8 (&bfr)[TN]) {
    if (ABL(64)) return;
    const unsigned char* wbase = sW + wb * BN * QKC;
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[j] = read_row(wbase, j * 32 + (l & 31));
  };

  f32x16 acc[TM][TN];
  int tile = blockIdx.x;
  if (tile >= tiles_total) return;
  int b = __builtin_amdgcn_readfirstlane(halo_offsets(tile));
  __amdgpu_buffer_rsrc_t rx = make_rx(b);
  stage_halo(rx, 0, 0);
  int pc = 0, pt = 0;
#pragma unroll
  for (int s = 0; s < D; ++s) {
    stage_weights(pc, pt, s);
    if (++pt == 9) { pt = 0; if (++pc == nChunks) pc = 0; }
  }
  wait_vm<(D - 2) * WJ>();                            // halo + slabs 0 and 1 (step 0 reads the fragments of step 1)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (also publishes sS)
  int gchunk = 0, wb = 0, wnext = D % WS;
  i32x8 fa[TM], fb[TN], nb[TN];
  if (ABL(64)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = i32x8{l, l, l, l, l, l, l, l};
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = nb[j] = i32x8{l, l, l, l, l, l, l, l};
  }
  read_a(0, 0, 0, TM, fa);
  read_b(0, fb);
  const int SC = 0x7f7f7f7f;                          // E8M0 block scales: 2^0 for every 32-element block

  for (; tile < tiles_total; tile += gridDim.x) {
    const int rem = tile - b * (tiles_x * tiles_y);
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int ty0 = tyi * TH, tx0 = txi * TW;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // one (chunk, tap) step.  Registers: 256 accumulators + 3 x 32 fragment registers -- the activation fragments of the
    // next step are read IN PLACE (rows 0, 1 once the first eight MFMAs, which are the ones that use them, have issued; rows
    // 2, 3 after the last eight), the weight fragments into the other of two sets (tap loop unrolled by two: copying next
    // -> current costs 32 v_mov, an eighth of a step's MFMA issue time).
    for (int chunk = 0; chunk < nChunks; ++chunk, ++gchunk) {
      const int hb = gchunk & 1;
      // the next halo: next chunk, or (the ring keeps running) chunk 0 of the workgroup's next tile
      if (chunk + 1 == nChunks) {
        const int nbi = __builtin_amdgcn_readfirstlane(halo_offsets(tile + gridDim.x));
        if (tile + gridDim.x < tiles_total) { b = nbi; rx = make_rx(b); }
        stage_halo(rx, 0, hb ^ 1);
      } else {
        stage_halo(rx, chunk + 1, hb ^ 1);
      }
      auto step = [&](int tap, i32x8 (&cb)[TN], i32x8 (&pb)[TN]) __attribute__((always_inline)) {
        stage_weights(pc, pt, wnext);
        if (++pt == 9) { pt = 0; if (++pc == nChunks) pc = 0; }
        // the slab and halo of step + 1 are visible (previous step's wait + barrier)
        const int wrap = tap == 8;
        const int wb1 = wb + 1 == WS ? 0 : wb + 1;
        const int hb1 = hb ^ wrap, tap1 = wrap ? 0 : tap + 1;
        auto mfmas = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              if (i >= i0 && i < i1) {
                if (!ABL(128)) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cb[j], fa[i], acc[i][j], 0, 0, 0, SC, 0, SC);
                else asm volatile("" : "+v"(acc[i][j]) : "v"(fa[i]), "v"(cb[j]));
              }
        };
        read_b(wb1, pb);
        mfmas(0, TM / 2);
        read_a(hb1, tap1, 0, TM / 2, fa);
        mfmas(TM / 2, TM);
        read_a(hb1, tap1, TM / 2, TM, fa);
        // the slab of step + 2 must have landed before the next step reads it: all but the requests of the last D-2 steps,
        // plus this chunk's halo burst (issued before tap 0's slab request) while tap <= D-3.
        // (no lgkmcnt wait here: LDS reads return in order and the next step waits for these fragments before its barrier,
        // so every read is done at least one barrier before its slot or halo buffer is written again)
        if (tap <= D - 3) wait_vm<(D - 2) * WJ + HT>();
        else wait_vm<(D - 2) * WJ>();
        if (!ABL(32)) __builtin_amdgcn_s_barrier();
        wb = wb1;
        wnext = wnext + 1 == WS ? 0 : wnext + 1;
      };
#pragma unroll 1
      for (int tap = 0; tap < 8; tap += 2) {
        step(tap, fb, nb);
        step(tap + 1, nb, fb);
      }
      step(8, fb, nb);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = nb[j];        // nine taps: one copy per chunk puts the sets back in phase
    }
    // ---- epilogue (k_conv3x3p's, plus the scales) ----
    {
      int le = l;
      asm volatile("" : "+v"(le));
      const int half = le >> 5, px = le & 31;
      const int bt = (tile - rem) / (tiles_x * tiles_y);
      bf16_t* yb = a.y + (long long)bt * a.y_bstride;
      float s[TN][2], ss[TN][2];
#pragma unroll
      for (int j = 0; j < TN; ++j) s[j][0] = s[j][1] = ss[j][0] = ss[j][1] = 0.f;
      const int x = tx0 + px;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int y = ty0 + wv * TM + i;
        const bool pok = y < H && x < W;
        bf16_t* row = yb + ((long long)y * W + x) * a.y_ld + n0 + half * 8;
        uint4 u[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          __builtin_amdgcn_sched_barrier(0);
          // (the tile stays in its accumulation registers until here: without this the compiler reads a few of the LAST tiles'
          // values at the top of the epilogue, spills them, and every reload waits vmcnt(0) -- behind the stores before it)
          asm volatile("" : "+a"(acc[i][j]));
          uint32_t d[4][2];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            // accumulator r of this lane is output channel j * 32 + (r&3) + 8 (r>>2) + 4 half
            // (ext_vector_type read: behind a float4 struct load the compiler waits vmcnt(0), i.e. for the whole ring -- common.h)
            typedef __attribute__((ext_vector_type(4))) float f32x4s;
            const f32x4s sc = *reinterpret_cast<const f32x4s*>(sS + j * 32 + 8 * g4 + 4 * half);
            d[g4][0] = pack_bf2(acc[i][j][4 * g4] * sc[0], acc[i][j][4 * g4 + 1] * sc[1]);
            d[g4][1] = pack_bf2(acc[i][j][4 * g4 + 2] * sc[2], acc[i][j][4 * g4 + 3] * sc[3]);
            if (pok) {
              const float v0 = bf_lo(d[g4][0]), v1 = bf_hi(d[g4][0]), v2 = bf_lo(d[g4][1]), v3 = bf_hi(d[g4][1]);
              s[j][g4 >> 1] += (v0 + v1) + (v2 + v3);
              ss[j][g4 >> 1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            }
          }
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            auto r0 = __builtin_amdgcn_permlane32_swap(d[2 * pr][0], d[2 * pr + 1][0], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(d[2 * pr][1], d[2 * pr + 1][1], false, false);
            u[j][pr] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pok && (!ABL(512) || a.x_scale == 12345.f)) {
          if (MODE == 1 && a.accumulate) {
            // gradient accumulation as in k_conv3x3p: all of the row's old values first (one 16-channel half of every 32-column
            // tile at a time), one wait, then new = bf16(bf16(v) + old)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              uint4 o[TN];
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                o[j] = make_uint4(0, 0, 0, 0);
                if (n0 + j * 32 + pr * 16 + half * 8 < a.Cout) o[j] = *reinterpret_cast<const uint4*>(row + j * 32 + pr * 16);
              }
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                uint4& v = u[j][pr];
                const uint4 q = o[j];
                v.x = pack_bf2(bf_lo(v.x) + bf_lo(q.x), bf_hi(v.x) + bf_hi(q.x));
                v.y = pack_bf2(bf_lo(v.y) + bf_lo(q.y), bf_hi(v.y) + bf_hi(q.y));
                v.z = pack_bf2(bf_lo(v.z) + bf_lo(q.z), bf_hi(v.z) + bf_hi(q.z));
                v.w = pack_bf2(bf_lo(v.w) + bf_lo(q.w), bf_hi(v.w) + bf_hi(q.w));
              }
            }
          }
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
              if (n0 + j * 32 + pr * 16 + half * 8 < a.Cout) *reinterpret_cast<uint4*>(row + j * 32 + pr * 16) = u[j][pr];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (a.stats_partial) {
        float* prow = a.stats_partial + ((((long long)bt * (tiles_x * tiles_y) + rem) * NW + wv) * a.G16) * 2;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            // (ds_bpermute on the opaque lane id: __shfl_xor keeps the kernel-entry lane id live across the main loop, where it
            // is spilled -- and its reload waits vmcnt(0) behind the tile's stores and the ring)
            float sv = s[j][sl], sq = ss[j][sl];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              sv += __int_as_float(__builtin_amdgcn_ds_bpermute((le ^ o) << 2, __float_as_int(sv)));
              sq += __int_as_float(__builtin_amdgcn_ds_bpermute((le ^ o) << 2, __float_as_int(sq)));
            }
            const int gidx = ((n0 + j * 32) >> 4) + sl;
            if (le == 0 && gidx < a.G16) *reinterpret_cast<float2*>(prow + gidx * 2) = make_float2(sv, sq);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      read_a(gchunk & 1, 0, 0, TM, fa);               // (again: so that the fragments need not survive the epilogue)
      read_b(wb, fb);
    }
  }
  wait_vm<0>();
}

template <int TN, int MODE = 0>
int launch8(const F8K& k, int B, hipStream_t st) {
  constexpr int BN = TN * 32, WS = 6;
  const int tiles_x = cdiv(k.W, TW), tiles_y = cdiv(k.H, TH);
  const int tiles_total = tiles_x * tiles_y * B;
  const int gy = cdiv(k.Cout, BN);
  int gx = 256 / gy;
  if (gx < 1) gx = 1;
  if (gx > tiles_total) gx = tiles_total;
  const size_t lds = (size_t)(2 * HPAD * QKC + WS * BN * QKC + 16 * QKC) + BN * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) { crd_reserve_lds(reinterpret_cast<const void*>(&k_conv3x3_fp8<TN, WS, MODE>), (int)lds, "k_conv3x3_fp8"); attr_done = true; }
  hipLaunchKernelGGL((k_conv3x3_fp8<TN, WS, MODE>), dim3(gx, gy), dim3(256), lds, st, k, tiles_x, tiles_y, tiles_total);
  CRD_LAUNCH_CHECK("crd_conv3x3_fp8");
  return CRD_OK;
}

// ---- quantisation helpers -------------------------------------------------------------------------------------------
// amax over a channel slice of a pixel-major bf16 tensor -> atomicMax on the (non-negative) float's bit pattern
__global__ __launch_bounds__(256) void k_amax_bf16(const bf16_t* x, long long rows, int ld, int C, unsigned* out) {
  const int CG = C >> 3;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * CG; i += (long long)gridDim.x * 256) {
    const long long r = i / CG;
    const int g = (int)(i - r * CG);
    float v[8];
    load8(x, r * ld + g * 8, 0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// y = e4m3(x * inv_scale), 8 channels per thread
__global__ __launch_bounds__(256) void k_quant_fp8(const bf16_t* x, long long rows, int ld, int C, unsigned char* y, int y_ld, float inv_scale) {
  const int CG = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * CG; i += (long long)gridDim.x * 256) {
    const long long r = i / CG;
    const int g = (int)(i - r * CG);
    float v[8];
    load8(x, r * ld + g * 8, 0, v);
    uint2 q;
    q.x = pack_fp8x4(v[0] * inv_scale, v[1] * inv_scale, v[2] * inv_scale, v[3] * inv_scale);
    q.y = pack_fp8x4(v[4] * inv_scale, v[5] * inv_scale, v[6] * inv_scale, v[7] * inv_scale);
    *reinterpret_cast<uint2*>(y + r * y_ld + g * 8) = q;
  }
}

// the same with the scale in device memory (this step's amax / 448: just-in-time scaling of a gradient tensor)
__global__ __launch_bounds__(256) void k_quant_fp8_dev(const bf16_t* x, long long rows, int ld, int C, unsigned char* y, int y_ld, const float* scale) {
  const int CG = C >> 3;
  const float inv_scale = 1.f / fmaxf(*scale, 1e-30f);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * CG; i += (long long)gridDim.x * 256) {
    const long long r = i / CG;
    const int g = (int)(i - r * CG);
    float v[8];
    load8(x, r * ld + g * 8, 0, v);
    uint2 q;
    q.x = pack_fp8x4(v[0] * inv_scale, v[1] * inv_scale, v[2] * inv_scale, v[3] * inv_scale);
    q.y = pack_fp8x4(v[4] * inv_scale, v[5] * inv_scale, v[6] * inv_scale, v[7] * inv_scale);
    *reinterpret_cast<uint2*>(y + r * y_ld + g * 8) = q;
  }
}

// one wave per tensor: the max over its amax slots -> scale = margin * amax / 448 (kept when nothing was recorded); slots zeroed
// unless keep (round 6: the three gradient slices of a decoder stage share one scale; just-in-time scaling takes the RUNNING max as
// the stage's layers produce their slices, and only the last update of the stage clears the slots)
__global__ __launch_bounds__(64) void k_fp8_scale_update(unsigned* slots, float* scales, float margin, int keep) {
  static_assert(CRD_FP8_AMAX_SLOTS == 64, "one slot per lane");
  unsigned* s_ = slots + (long long)blockIdx.x * CRD_FP8_AMAX_SLOTS;
  float m = __uint_as_float(s_[threadIdx.x]);
  if (!keep) s_[threadIdx.x] = 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (threadIdx.x == 0 && m > 0.f) scales[blockIdx.x] = margin * m / E4M3_MAX;
}

// one workgroup per output channel: scale[co] = amax / 448 (1 if the row is all zero), w8[co][tap][0..Cin) = e4m3(w / scale),
// channels Cin .. Cin8-1 of every tap zero (the fp8 kernel walks K in 16-channel granules)
__global__ __launch_bounds__(256) void k_weight_quant_fp8(const bf16_t* w, int taps, int Cin, int Cin8, unsigned char* w8, float* scale) {
  __shared__ float red[4];
  const int co = blockIdx.x, K = taps * Cin;
  const bf16_t* row = w + (long long)co * K;
  float m = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) m = fmaxf(m, fabsf(bf2f(row[i])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = m > 0.f ? m / E4M3_MAX : 1.f;
  if (threadIdx.x == 0) scale[co] = sc;
  const float inv = 1.f / sc;
  for (int i = threadIdx.x * 4; i < taps * Cin8; i += 1024) {          // Cin, Cin8 multiples of 4
    const int tap = i / Cin8, c = i - tap * Cin8;
    unsigned q = 0;
    if (c < Cin) {
      const bf16_t* p = row + tap * Cin + c;
      q = pack_fp8x4(bf2f(p[0]) * inv, bf2f(p[1]) * inv, bf2f(p[2]) * inv, bf2f(p[3]) * inv);
    }
    *reinterpret_cast<unsigned*>(w8 + (long long)co * taps * Cin8 + i) = q;
  }
}

}  // namespace

extern "C" int crd_amax_bf16(const void* x, int64_t rows, int32_t ld, int32_t coff, int32_t C, float* amax, crd_stream_t stream) {
  CRD_CHECK_ARG(x && amax && rows > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && coff % 8 == 0, "crd_amax_bf16: bad argument");
  long long n = (rows * (C / 8) + 255) / 256;
  if (n > 2048) n = 2048;
  hipLaunchKernelGGL(k_amax_bf16, dim3((unsigned)n), dim3(256), 0, as_stream(stream), reinterpret_cast<const bf16_t*>(x) + coff, (long long)rows, ld, C,
                     reinterpret_cast<unsigned*>(amax));
  CRD_LAUNCH_CHECK("crd_amax_bf16");
  return CRD_OK;
}

extern "C" int crd_quant_fp8(const void* x, int64_t rows, int32_t ld, int32_t coff, int32_t C, void* y, int32_t y_ld, int32_t y_coff,
                             float scale, crd_stream_t stream) {
  CRD_CHECK_ARG(x && y && rows > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && coff % 8 == 0 && y_ld % 8 == 0 && y_coff % 8 == 0 && scale > 0.f,
                "crd_quant_fp8: bad argument (channels in multiples of 8)");
  long long n = (rows * (C / 8) + 255) / 256;
  if (n > 4096) n = 4096;
  hipLaunchKernelGGL(k_quant_fp8, dim3((unsigned)n), dim3(256), 0, as_stream(stream), reinterpret_cast<const bf16_t*>(x) + coff, (long long)rows, ld, C,
                     reinterpret_cast<unsigned char*>(y) + y_coff, y_ld, 1.f / scale);
  CRD_LAUNCH_CHECK("crd_quant_fp8");
  return CRD_OK;
}

extern "C" int crd_weight_quant_fp8(const void* w_bf16, int32_t Cout, int32_t taps, int32_t Cin, int32_t Cin_out, void* w_fp8, float* scales,
                                    crd_stream_t stream) {
  CRD_CHECK_ARG(w_bf16 && w_fp8 && scales && Cout > 0 && taps > 0 && Cin > 0 && Cin % 4 == 0 && Cin_out >= Cin && Cin_out % 4 == 0,
                "crd_weight_quant_fp8: bad argument");
  hipLaunchKernelGGL(k_weight_quant_fp8, dim3(Cout), dim3(256), 0, as_stream(stream), reinterpret_cast<const bf16_t*>(w_bf16), taps, Cin, Cin_out,
                     reinterpret_cast<unsigned char*>(w_fp8), scales);
  CRD_LAUNCH_CHECK("crd_weight_quant_fp8");
  return CRD_OK;
}

extern "C" int crd_quant_fp8_dev(const void* x, int64_t rows, int32_t ld, int32_t coff, int32_t C, void* y, int32_t y_ld, int32_t y_coff,
                                 const float* scale_dev, crd_stream_t stream) {
  CRD_CHECK_ARG(x && y && scale_dev && rows > 0 && C > 0 && C % 8 == 0 && ld % 8 == 0 && coff % 8 == 0 && y_ld % 8 == 0 && y_coff % 8 == 0,
                "crd_quant_fp8_dev: bad argument (channels in multiples of 8)");
  long long n = (rows * (C / 8) + 255) / 256;
  if (n > 4096) n = 4096;
  hipLaunchKernelGGL(k_quant_fp8_dev, dim3((unsigned)n), dim3(256), 0, as_stream(stream), reinterpret_cast<const bf16_t*>(x) + coff, (long long)rows, ld, C,
                     reinterpret_cast<unsigned char*>(y) + y_coff, y_ld, scale_dev);
  CRD_LAUNCH_CHECK("crd_quant_fp8_dev");
  return CRD_OK;
}

extern "C" int crd_fp8_scale_update(uint32_t* amax_slots, float* scales, int32_t n, float margin, int32_t keep_slots, crd_stream_t stream) {
  CRD_CHECK_ARG(amax_slots && scales && n > 0 && margin > 0.f, "crd_fp8_scale_update: bad argument");
  hipLaunchKernelGGL(k_fp8_scale_update, dim3(n), dim3(64), 0, as_stream(stream), reinterpret_cast<unsigned*>(amax_slots), scales, margin, keep_slots);
  CRD_LAUNCH_CHECK("crd_fp8_scale_update");
  return CRD_OK;
}

extern "C" int crd_conv3x3_fp8_dgrad(const crd_conv_desc* d, const float* w_scales, const float* x_scale_dev, crd_stream_t stream) {
  CRD_CHECK_ARG(d && d->x && d->w && d->y && w_scales && x_scale_dev, "crd_conv3x3_fp8_dgrad: null pointer");
  CRD_UNSUPPORTED(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->gather_mode == 1 && d->out_mode == 0 && d->IH == d->OH &&
                  d->IW == d->OW && !d->y_f32 && !d->bias && !d->act && !d->res && !d->stats && !d->red_x && !d->chan_sums,
                  "crd_conv3x3_fp8_dgrad: 3x3 / stride 1 / pad 1 data gradient with a plain or accumulating bf16 output only");
  CRD_CHECK_ARG(d->Cin % 16 == 0 && d->x_ld % 16 == 0 && d->x_coff % 16 == 0 && d->Cout % 8 == 0 && d->y_ld % 8 == 0 && d->y_coff % 8 == 0,
                "crd_conv3x3_fp8_dgrad: fp8 channels in multiples of 16, output channels of 8");
  CRD_UNSUPPORTED((long long)d->IH * d->IW * d->x_ld < (1ll << 31) && (long long)d->Cout * 9 * d->Cin < (1ll << 31), "crd_conv3x3_fp8_dgrad: tensor too large");
  F8K k;
  k.x = reinterpret_cast<const unsigned char*>(d->x) + d->x_coff; k.x_ld = d->x_ld; k.x_bstride = (long long)d->IH * d->IW * d->x_ld;
  k.Cin = d->Cin; k.Ktot = 9 * d->Cin;
  k.x_scale = 1.f; k.x_scale_dev = x_scale_dev; k.accumulate = d->accumulate; k.H = d->IH; k.W = d->IW;
  k.y_ld = d->y_ld; k.y_bstride = (long long)d->OH * d->OW * d->y_ld;
  k.G16 = 0; k.stats_partial = nullptr;
  hipStream_t st = as_stream(stream);
  // 128-column tiles for the bulk, one narrower launch for what is left (N = 144 / 240 / 304 in the decoder)
  const int main_cols = d->Cout / 128 * 128, rest = d->Cout - main_cols;
  for (int part = 0; part < 2; ++part) {
    const int c0 = part == 0 ? 0 : main_cols, cols = part == 0 ? main_cols : rest;
    if (cols <= 0) continue;
    k.w = reinterpret_cast<const unsigned char*>(d->w) + (long long)c0 * k.Ktot; k.Cout = cols; k.w_scale = w_scales + c0;
    k.y = reinterpret_cast<bf16_t*>(d->y) + d->y_coff + c0;
    const int rc = cols <= 64 ? launch8<2, 1>(k, d->B, st) : cols <= 96 ? launch8<3, 1>(k, d->B, st) : launch8<4, 1>(k, d->B, st);
    if (rc != CRD_OK) return rc;
  }
  return CRD_OK;
}

extern "C" int crd_conv3x3_fp8(const crd_conv_desc* d, const float* w_scales, float x_scale, crd_stream_t stream) {
  CRD_CHECK_ARG(d && d->x && d->w && d->y && w_scales && x_scale > 0.f, "crd_conv3x3_fp8: null pointer / bad scale");
  CRD_UNSUPPORTED(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->gather_mode == 0 && d->out_mode == 0 && d->IH == d->OH &&
                  d->IW == d->OW && !d->y_f32 && !d->bias && !d->act && !d->res && !d->accumulate && !d->red_x && !d->chan_sums,
                  "crd_conv3x3_fp8: 3x3 / stride 1 / pad 1 forward with a plain bf16 output only");
  CRD_CHECK_ARG(d->Cin % 16 == 0 && d->x_ld % 16 == 0 && d->x_coff % 16 == 0 && d->Cout % 8 == 0 && d->y_ld % 8 == 0 && d->y_coff % 8 == 0,
                "crd_conv3x3_fp8: fp8 channels in multiples of 16, output channels of 8");
  CRD_UNSUPPORTED((long long)d->IH * d->IW * d->x_ld < (1ll << 31) && (long long)d->Cout * 9 * d->Cin < (1ll << 31), "crd_conv3x3_fp8: tensor too large");
  F8K k;
  k.x = reinterpret_cast<const unsigned char*>(d->x) + d->x_coff; k.x_ld = d->x_ld; k.x_bstride = (long long)d->IH * d->IW * d->x_ld;
  k.w = reinterpret_cast<const unsigned char*>(d->w); k.Cout = d->Cout; k.Cin = d->Cin; k.Ktot = 9 * d->Cin;
  k.w_scale = w_scales; k.x_scale = x_scale; k.x_scale_dev = nullptr; k.accumulate = 0; k.H = d->IH; k.W = d->IW;
  k.y = reinterpret_cast<bf16_t*>(d->y) + d->y_coff; k.y_ld = d->y_ld; k.y_bstride = (long long)d->OH * d->OW * d->y_ld;
  k.G16 = d->Cout / 16;
  k.stats_partial = nullptr;
  const long long rows = (long long)cdiv(d->IW, TW) * cdiv(d->IH, TH) * NW;
  if (d->stats) {
    CRD_CHECK_ARG(d->Cout % 16 == 0 && d->stats_partial && (long long)d->B * rows * k.G16 * 2 <= d->stats_partial_capacity,
                  "crd_conv3x3_fp8: GroupNorm sums need Cout %% 16 == 0 and a stats_partial buffer of B x ceil(W/32) x ceil(H/16) x 4 x Cout/16 x 2 floats");
    k.stats_partial = d->stats_partial;
  }
  hipStream_t st = as_stream(stream);
  int rc;
  if (d->Cout <= 64) rc = launch8<2>(k, d->B, st);
  else if (d->Cout <= 96) rc = launch8<3>(k, d->B, st);
  else rc = launch8<4>(k, d->B, st);
  if (rc != CRD_OK || !d->stats) return rc;
  hipLaunchKernelGGL(k_stats_finalize, dim3(k.G16, d->B), dim3(256), 0, st, k.stats_partial, (int)rows, k.G16, d->stats);
  CRD_LAUNCH_CHECK("crd_conv3x3_fp8(statistics)");
  return CRD_OK;
}
