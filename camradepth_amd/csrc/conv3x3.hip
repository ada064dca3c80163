// Halo-tile 3x3 convolution (stride 1, pad 1) on MFMA for gfx950 -- the decoder's ShortResBlock convs, which are
// ~85 % of the model's FLOPs (src/utils/utils.py:114-124,211), forward and data gradient.
//
// The generic implicit-GEMM kernel (igemm.hip) re-gathers every input pixel once per tap and re-reads the weight
// tensor for every 128-pixel tile; on the big layers that is ~10 GB of L2->LDS traffic per launch and the kernel is
// bound by it.  Here a workgroup owns an 8 x 32 = 256-pixel 2-D output tile and walks K as (channel chunk of 64) x
// (9 taps):
//   * the (8+2) x (32+2) input halo of the chunk is brought into LDS ONCE (LDS-DMA, double buffered, the next
//     chunk's halo is prefetched in small pieces during taps 0..5) and serves all nine taps -- a tap is just a
//     different row offset into the halo image;
//   * the [Cout tile][64] weight slab of each (chunk, tap) is streamed by LDS-DMA through a 4-slot ring, requested
//     three steps ahead;
//   * 8 waves, each a (TM x TN) grid of 32x32 v_mfma_f32_32x32x16_bf16 tiles; one barrier per (chunk, tap).
// Out-of-image halo pixels, the channel tail and partial tiles are zero-filled by the buffer bounds check, so the
// hot loop has no branches.  MODE 0: forward (weights [Cout][tap][Cin]); MODE 1: data gradient (weights
// [Cin][tap][Cout], taps mirrored).  Same argument block and fused epilogue as igemm.hip.
#include "conv_common.h"

using namespace crdk;

namespace {

constexpr int TH = 8, TW = 32;            // output tile (pixels)
constexpr int HW_ = TW + 2;               // halo width
constexpr int HROWS = (TH + 2) * HW_;     // 340 halo pixels
constexpr int HGROUPS = (HROWS + 7) / 8;  // 43 DMA groups of 8 rows
constexpr int HPAD = HGROUPS * 8;         // 344 rows allocated
constexpr int CK = 64;                    // channels per chunk (8 granules, 128-byte LDS rows)
constexpr int NW = 8;                     // waves per workgroup
constexpr int HTAPS = 6;                  // taps 0..5 of a chunk carry the next chunk's halo (6 x 8 waves = 48 >= 43 groups)
// (weight-slab ring size WS is a kernel template parameter: 4 = the slab of step t+3 is in flight while step t computes;
// 3 for the 160-column tile, whose slabs are 20 KB)

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WM, int WN, int TM, int TN, int MODE, int WS>
__global__ __launch_bounds__(512) void k_conv3x3_w8(ConvK a, int tiles_x) {
  constexpr int D = WS - 1;                       // prefetch distance of the weight slabs, in steps
  static_assert(WM * WN == NW && WM * TM == TH, "8 waves cover the 8 tile rows");
  constexpr int BN = WN * TN * 32;
  constexpr int WGROUPS = BN / 8;                 // weight-slab DMA groups (8 rows each)
  constexpr int WJ = (WGROUPS + NW - 1) / NW;     // groups per wave
  constexpr int PER = WJ + 1;                     // DMA instructions every wave issues per step: WJ weight groups, 1 halo group
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sH = lds;                       // [2][HPAD][CK]
  bf16_t* sW = lds + 2 * HPAD * CK;       // [WS][BN][CK]
  bf16_t* sD = sW + WS * BN * CK;         // [8][CK] landing area of the zero-fill dummies

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int b = blockIdx.z, n0 = blockIdx.y * BN;
  const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int H = a.IH, W = a.IW;           // stride 1, pad 1: output grid == input grid

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + (long long)b * a.x_bstride), 0, (int)(a.x_bstride * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const unsigned OOB = 0x80000000u;

  // ---- per-thread DMA descriptors -------------------------------------------------------------------------
  // Every wave issues the SAME instruction sequence each step -- its WJ groups of the weight slab of step t+3, then one
  // group of the next chunk's halo (a zero-fill dummy, offset out of range, when there is nothing to fetch) -- so one
  // counted wait serves all steps: vmcnt is in order, and behind the slab of step t+1 there are exactly 1 + (D-1)*PER
  // younger DMAs.  A tap step is only ~0.25 us of MFMA work while an L2 hit takes 0.5-1 us, so the slab has to be
  // requested three steps ahead (a two-slot ring stalled every step); the HBM-latency halo groups sit BEHIND the slab
  // groups of their step, where nothing waits on them before the chunk ends.
  // halo: at tap s (0..5) wave w stages group G = 8 s + w: rows 8G + (l>>3), 16-byte slot l&7, which receives channel
  // granule (l&7) ^ ((row>>1)&7) (source-side swizzle).
  unsigned hoff[HTAPS];   // byte offset of the pixel (channel 0 of the chunk) or OOB
  int hch[HTAPS];         // channel offset of this lane's granule inside a chunk
#pragma unroll
  for (int s = 0; s < HTAPS; ++s) {
    const int G = NW * s + wv;
    const int hr = 8 * G + (l >> 3);
    const int hy = hr / HW_, hx = hr - hy * HW_;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const bool ok = G < HGROUPS && hr < HROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hch[s] = ((l & 7) ^ ((hr >> 1) & 7)) * 8;
    hoff[s] = ok ? (unsigned)((iy * W + ix) * a.x_ld * 2) : OOB;
  }
  // weight slab: wave w stages groups g = 8 j + w: rows n = 8 g + (l>>3); (n>>1)&7 = (4 w + (l>>4)) & 7 for every j
  const int wch = ((l & 7) ^ ((4 * wv + (l >> 4)) & 7)) * 8;
  unsigned woff[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int g = NW * j + wv;
    const int n = 8 * g + (l >> 3), ng = n0 + n;
    woff[j] = (g < WGROUPS && ng < a.Cout) ? (unsigned)(ng * a.Ktot * 2) : OOB;
  }
  const int Cin = a.Cin;
  const int nChunks = (Cin + CK - 1) / CK;

  // one halo group; s < 0 or a chunk past the end: zero-fill dummy
  auto stage_halo_piece = [&](int s, int chunk, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool real = s >= 0 && chunk < nChunks && NW * s + wv < HGROUPS;
    const int si = s < 0 ? 0 : s;
    const int ch = chunk * CK + hch[si];
    const unsigned off = (real && ch < Cin) ? hoff[si] + (unsigned)(ch * 2) : OOB;
    bf16_t* dst = real ? sH + buf * HPAD * CK + (NW * si + wv) * 8 * CK : sD;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)dst, 16, real ? (off | (hoff[si] & OOB)) : OOB, 0, 0, 0);
#else
    (void)s; (void)chunk; (void)buf;
#endif
  };
  auto stage_weights = [&](int chunk, int tap, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int ch = chunk * CK + wch;
    const unsigned koff = (unsigned)((tap * Cin + ch) * 2);
    const bool real = chunk < nChunks && ch < Cin;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int g = NW * j + wv;
      const unsigned off = real ? woff[j] + koff : OOB;
      bf16_t* dst = g < WGROUPS ? sW + slot * BN * CK + 8 * g * CK : sD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, g < WGROUPS ? (off | (woff[j] & OOB)) : OOB, 0, 0, 0);
    }
#else
    (void)chunk; (void)tap; (void)slot;
#endif
  };

  // accumulators start at the bias of their column (lane l holds column l&31 of every 32x32 tile): the load overlaps the
  // first DMA instead of adding a dependent memory latency to the epilogue
  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (l & 31);
    const float bias_v = (a.bias && col < a.Cout) ? a.bias[(long long)b * a.bias_bstride + col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bias_v;
  }

  // prologue, shaped like D steps of the loop: (whole halo of chunk 0, slab 0), then (slab s, dummy) for s = 1..D-1
#pragma unroll
  for (int s = 0; s < HTAPS; ++s) stage_halo_piece(s, 0, 0);
  stage_weights(0, 0, 0);
#pragma unroll
  for (int s = 1; s < D; ++s) { stage_weights(0, s, s); stage_halo_piece(-1, 0, 1); }
  wait_vm<(D - 1) * PER>();               // halo of chunk 0 and slab 0 (everything but the last D-1 "steps")
  asm volatile("s_barrier" ::: "memory");

  int step = 0;
  int pc = 0, pt = D;                      // (chunk, tap) of the slab to request: step + D
  int wb = 0, wnext = D % WS;              // ring slots: slab of this step / slab being requested
  for (int chunk = 0; chunk < nChunks; ++chunk) {
    const int hb = chunk & 1;
    int nks = (Cin - chunk * CK + 15) >> 4;
    if (nks > 4) nks = 4;
    for (int tap = 0; tap < 9; ++tap, ++step) {
      stage_weights(pc, pt, wnext);
      stage_halo_piece(tap < HTAPS ? tap : -1, chunk + 1, hb ^ 1);
      if (++pt == 9) { pt = 0; ++pc; }
      const int ky = tap / 3, kx = tap - ky * 3;
      const int oy = (MODE == 0) ? ky : 2 - ky, ox = (MODE == 0) ? kx : 2 - kx;
      const bf16_t* hbase = sH + hb * HPAD * CK;
      const bf16_t* wbase = sW + wb * BN * CK;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < nks) {
          bf16x8 af[TM], bfr[TN];
          const int gi = ks * 2 + (l >> 5);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int hr = (wm * TM + i + oy) * HW_ + (l & 31) + ox;
            af[i] = *reinterpret_cast<const bf16x8*>(hbase + hr * CK + ((gi ^ ((hr >> 1) & 7)) << 3));
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int row = (wn * TN + j) * 32 + (l & 31);
            bfr[j] = *reinterpret_cast<const bf16x8*>(wbase + row * CK + ((gi ^ ((row >> 1) & 7)) << 3));
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      }
      // the slab of step+1 (requested two steps ago) and every older DMA of this wave -- in particular the next chunk's
      // halo groups, requested in taps 0..5 -- have landed; the barrier extends that to all waves.
      // Raw s_barrier: __syncthreads() would drain the whole DMA queue (vmcnt(0)) here.
      wait_vm<1 + (D - 1) * PER>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      wb = wb + 1 == WS ? 0 : wb + 1;
      wnext = wnext + 1 == WS ? 0 : wnext + 1;
    }
  }
  wait_vm<0>();                            // dummy slabs / halo groups still target the LDS the epilogue reuses
  asm volatile("s_barrier" ::: "memory");

  if (a.dbg & 4) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(a.y)[0] = 1.f; return; }
  conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, blockIdx.x, lds,
                                [&](int i, int rr, bool& valid, int& row) {
    const int y = ty0 + wm * TM + i, x = tx0 + rr;
    valid = y < H && x < W;
    row = y * W + x;
  }, [&](int rl, bool& valid, int& row) {
    const int y = ty0 + (rl >> 5), x = tx0 + (rl & 31);
    valid = y < H && x < W;
    row = y * W + x;
  });
}

template <int WM, int WN, int TM, int TN, int WS = 4>
int launch3_w8(const ConvK& k0, int B, hipStream_t st, long long partial_cap) {
  constexpr int BN = WN * TN * 32;
  ConvK k = k0;
  k.n_tiles = cdiv(k.IW, TW) * cdiv(k.IH, TH);
  if ((long long)B * k.n_tiles * k.G16 * 2 > partial_cap) k.stats_partial = nullptr;
  const size_t lds = (size_t)(2 * HPAD * CK + WS * BN * CK + 8 * CK) * sizeof(bf16_t);
  const int tiles_x = cdiv(k.IW, TW), tiles_y = cdiv(k.IH, TH);
  dim3 grid(tiles_x * tiles_y, cdiv(k.Cout, BN), B);
  static bool attr_done[2] = {false, false};
  if (k.gather_mode == 0) {
    if (!attr_done[0]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_w8<WM, WN, TM, TN, 0, WS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done[0] = true;
    }
    hipLaunchKernelGGL((k_conv3x3_w8<WM, WN, TM, TN, 0, WS>), grid, dim3(512), lds, st, k, tiles_x);
  } else {
    if (!attr_done[1]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_w8<WM, WN, TM, TN, 1, WS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done[1] = true;
    }
    hipLaunchKernelGGL((k_conv3x3_w8<WM, WN, TM, TN, 1, WS>), grid, dim3(512), lds, st, k, tiles_x);
  }
  if (k.stats && k.stats_partial)
    hipLaunchKernelGGL(k_stats_finalize, dim3(k.G16, B), dim3(64), 0, st, k.stats_partial, k.n_tiles, k.G16, k.stats);
  CRD_LAUNCH_CHECK("crd_conv_igemm(3x3 halo)");
  return CRD_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Four-wave variant: the same halo-tile scheme with 32-channel chunks, so that a workgroup needs <= 80 KB of LDS and
// 256 VGPRs and TWO of them share a CU.  Measured on the 8-wave kernel above (one workgroup per CU, s_memtime stamps):
// the prologue (first halo + slabs, an HBM burst of every CU at once) and the epilogue (64 KB of stores per tile) took
// 23 % of a workgroup's life on the 304->128 layer and 54 % on the single-chunk 64->240 data gradient, with the matrix
// pipe idle; and inside the loop both waves of a SIMD stalled on the same barrier.  With two independent workgroups
// per CU one computes while the other loads, stores or waits.  Per wave: (TM x TN) 32x32 tiles with TM*WM = 8 rows,
// i.e. 64x128 for the 128-column tile -- 0.75 fragment reads per MFMA instead of 1.
constexpr int QK = 32;                      // channels per chunk: 64-byte LDS rows, 4 granules
constexpr int QNW = 4;                      // waves per workgroup
constexpr int QHG = (HROWS + 15) / 16;      // 22 halo DMA pieces (16 rows x 64 B each)
constexpr int QHPAD = QHG * 16;             // 352 rows allocated
constexpr int QHTAPS = 6;                   // taps 0..5 carry the next chunk's halo: 6 x 4 waves = 24 >= 22 pieces

template <int WM, int WN, int TM, int TN, int MODE, int WS>
__global__ __launch_bounds__(256, 2) void k_conv3x3(ConvK a, int tiles_x) {
  constexpr int D = WS - 1;
  static_assert(WM * WN == QNW && WM * TM == TH, "4 waves cover the 8 tile rows");
  constexpr int BN = WN * TN * 32;
  constexpr int WGROUPS = BN / 16;                // weight-slab DMA pieces (16 rows x 64 B)
  constexpr int WJ = (WGROUPS + QNW - 1) / QNW;   // pieces per wave
  constexpr int PER = WJ + 1;                     // DMA instructions every wave issues per step
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sH = lds;                         // [2][QHPAD][QK]
  bf16_t* sW = lds + 2 * QHPAD * QK;        // [WS][BN][QK]
  bf16_t* sD = sW + WS * BN * QK;           // [16][QK] landing area of the zero-fill dummies

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int b = blockIdx.z, n0 = blockIdx.y * BN;
  const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int H = a.IH, W = a.IW;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + (long long)b * a.x_bstride), 0, (int)(a.x_bstride * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const unsigned OOB = 0x80000000u;

  // DMA descriptors, as in the 8-wave kernel with 64-byte rows: a piece is 16 rows; lane l stages row 16 G + (l>>2),
  // 16-byte slot l&3, which receives channel granule (l&3) ^ ((row>>2)&3) (source-side swizzle: the 16 rows a
  // ds_read_b128 lane group touches are distinct mod 16, so (row&3, slot) -- the 16-byte bank slot -- is distinct too).
  unsigned hoff[QHTAPS];
  int hch[QHTAPS];
#pragma unroll
  for (int s = 0; s < QHTAPS; ++s) {
    const int G = QNW * s + wv;
    const int hr = 16 * G + (l >> 2);
    const int hy = hr / HW_, hx = hr - hy * HW_;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const bool ok = G < QHG && hr < HROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hch[s] = ((l & 3) ^ ((hr >> 2) & 3)) * 8;
    hoff[s] = ok ? (unsigned)((iy * W + ix) * a.x_ld * 2) : OOB;
  }
  // weight slab: wave w stages pieces g = 4 j + w: rows n = 16 g + (l>>2); (n>>2)&3 = (l>>4)&3 for every g
  const int wch = ((l & 3) ^ ((l >> 4) & 3)) * 8;
  unsigned woff[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int g = QNW * j + wv;
    const int n = 16 * g + (l >> 2), ng = n0 + n;
    woff[j] = (g < WGROUPS && ng < a.Cout) ? (unsigned)(ng * a.Ktot * 2) : OOB;
  }
  const int Cin = a.Cin;
  const int nChunks = (Cin + QK - 1) / QK;

  auto stage_halo_piece = [&](int s, int chunk, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool real = s >= 0 && chunk < nChunks && QNW * s + wv < QHG;
    const int si = s < 0 ? 0 : s;
    const int ch = chunk * QK + hch[si];
    const unsigned off = (real && ch < Cin) ? hoff[si] + (unsigned)(ch * 2) : OOB;
    bf16_t* dst = real ? sH + buf * QHPAD * QK + (QNW * si + wv) * 16 * QK : sD;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)dst, 16, real ? (off | (hoff[si] & OOB)) : OOB, 0, 0, 0);
#else
    (void)s; (void)chunk; (void)buf;
#endif
  };
  auto stage_weights = [&](int chunk, int tap, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int ch = chunk * QK + wch;
    const unsigned koff = (unsigned)((tap * Cin + ch) * 2);
    const bool real = chunk < nChunks && ch < Cin;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int g = QNW * j + wv;
      const unsigned off = real ? woff[j] + koff : OOB;
      bf16_t* dst = g < WGROUPS ? sW + slot * BN * QK + 16 * g * QK : sD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, g < WGROUPS ? (off | (woff[j] & OOB)) : OOB, 0, 0, 0);
    }
#else
    (void)chunk; (void)tap; (void)slot;
#endif
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + (l & 31);
    const float bias_v = (a.bias && col < a.Cout) ? a.bias[(long long)b * a.bias_bstride + col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = bias_v;
  }

  // prologue, shaped like D steps of the loop: (whole halo of chunk 0, slab 0), then (slab s, dummy) for s = 1..D-1
#pragma unroll
  for (int s = 0; s < QHTAPS; ++s) stage_halo_piece(s, 0, 0);
  stage_weights(0, 0, 0);
#pragma unroll
  for (int s = 1; s < D; ++s) { stage_weights(0, s, s); stage_halo_piece(-1, 0, 1); }
  wait_vm<(D - 1) * PER>();
  asm volatile("s_barrier" ::: "memory");

  int pc = 0, pt = D;                      // (chunk, tap) of the slab to request: step + D
  int wb = 0, wnext = D % WS;
  for (int chunk = 0; chunk < nChunks; ++chunk) {
    const int hb = chunk & 1;
    const bool two = Cin - chunk * QK > 16;        // second k-step of the chunk holds channels
    for (int tap = 0; tap < 9; ++tap) {
      stage_weights(pc, pt, wnext);
      stage_halo_piece(tap < QHTAPS ? tap : -1, chunk + 1, hb ^ 1);
      if (++pt == 9) { pt = 0; ++pc; }
      const int ky = tap / 3, kx = tap - ky * 3;
      const int oy = (MODE == 0) ? ky : 2 - ky, ox = (MODE == 0) ? kx : 2 - kx;
      const bf16_t* hbase = sH + hb * QHPAD * QK;
      const bf16_t* wbase = sW + wb * BN * QK;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 0 || two) {
          bf16x8 af[TM], bfr[TN];
          const int gi = ks * 2 + (l >> 5);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int hr = (wm * TM + i + oy) * HW_ + (l & 31) + ox;
            af[i] = *reinterpret_cast<const bf16x8*>(hbase + hr * QK + ((gi ^ ((hr >> 2) & 3)) << 3));
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int row = (wn * TN + j) * 32 + (l & 31);
            bfr[j] = *reinterpret_cast<const bf16x8*>(wbase + row * QK + ((gi ^ ((row >> 2) & 3)) << 3));
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      }
      wait_vm<1 + (D - 1) * PER>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      wb = wb + 1 == WS ? 0 : wb + 1;
      wnext = wnext + 1 == WS ? 0 : wnext + 1;
    }
  }
  wait_vm<0>();
  asm volatile("s_barrier" ::: "memory");

  if (a.dbg & 4) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(a.y)[0] = 1.f; return; }
  conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, blockIdx.x, lds,
                                [&](int i, int rr, bool& valid, int& row) {
    const int y = ty0 + wm * TM + i, x = tx0 + rr;
    valid = y < H && x < W;
    row = y * W + x;
  }, [&](int rl, bool& valid, int& row) {
    const int y = ty0 + (rl >> 5), x = tx0 + (rl & 31);
    valid = y < H && x < W;
    row = y * W + x;
  });
}

template <int WM, int WN, int TM, int TN, int WS = 4>
int launch3(const ConvK& k0, int B, hipStream_t st, long long partial_cap) {
  constexpr int BN = WN * TN * 32, BM = TH * TW;
  ConvK k = k0;
  k.n_tiles = cdiv(k.IW, TW) * cdiv(k.IH, TH);
  if ((long long)B * k.n_tiles * k.G16 * 2 > partial_cap) k.stats_partial = nullptr;
  const size_t lds_main = (size_t)(2 * QHPAD * QK + WS * BN * QK + 16 * QK) * sizeof(bf16_t);
  const size_t lds_epi = (size_t)BM * (BN + 8) * sizeof(bf16_t) + 4096;     // conv_epilogue's staging tile + fold scratch
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  const int tiles_x = cdiv(k.IW, TW), tiles_y = cdiv(k.IH, TH);
  dim3 grid(tiles_x * tiles_y, cdiv(k.Cout, BN), B);
  static bool attr_done[2] = {false, false};
  if (k.gather_mode == 0) {
    if (!attr_done[0]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<WM, WN, TM, TN, 0, WS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done[0] = true;
    }
    hipLaunchKernelGGL((k_conv3x3<WM, WN, TM, TN, 0, WS>), grid, dim3(256), lds, st, k, tiles_x);
  } else {
    if (!attr_done[1]) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<WM, WN, TM, TN, 1, WS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_done[1] = true;
    }
    hipLaunchKernelGGL((k_conv3x3<WM, WN, TM, TN, 1, WS>), grid, dim3(256), lds, st, k, tiles_x);
  }
  if (k.stats && k.stats_partial)
    hipLaunchKernelGGL(k_stats_finalize, dim3(k.G16, B), dim3(64), 0, st, k.stats_partial, k.n_tiles, k.G16, k.stats);
  CRD_LAUNCH_CHECK("crd_conv_igemm(3x3 halo)");
  return CRD_OK;
}

}  // namespace

// Called from crd_conv_igemm for 3x3 / stride 1 / pad 1 layers on grids large enough to fill the chip.
int crd_conv3x3_halo(const ConvK& k, int B, hipStream_t st, long long partial_cap) {
  if (k.Cout <= 32) return launch3<4, 1, 2, 1>(k, B, st, partial_cap);
  if (k.Cout <= 64) return launch3<4, 1, 2, 2>(k, B, st, partial_cap);
  if (k.Cout <= 96) return launch3<4, 1, 2, 3>(k, B, st, partial_cap);
  // 129..160 and 257..320 output channels (data gradients towards 136 / 144 / 296 / 304-channel concat inputs): 160-wide
  // tiles instead of a nearly empty last 128-wide one (8-wave kernel: its epilogue tile does not fit in half a CU's LDS)
  if ((k.Cout > 128 && k.Cout <= 160) || (k.Cout > 256 && k.Cout <= 320)) {
    return launch3_w8<8, 1, 1, 5, 3>(k, B, st, partial_cap);
  }
  return launch3<4, 1, 2, 4>(k, B, st, partial_cap);     // (64 x 128 per wave; 128 x 64 spills more and is 6 % slower)
}
