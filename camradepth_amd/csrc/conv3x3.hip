// Halo-tile 3x3 convolution (stride 1, pad 1) on MFMA for gfx950 -- the decoder's ShortResBlock convs, which are
// ~85 % of the model's FLOPs (src/utils/utils.py:114-124,211), forward and data gradient.
//
// The generic implicit-GEMM kernel (igemm.hip) re-gathers every input pixel once per tap and re-reads the weight
// tensor for every 128-pixel tile; on the big layers that is ~10 GB of L2->LDS traffic per launch and the kernel is
// bound by it.  Here a workgroup (4 waves) owns an 8 x 32 = 256-pixel 2-D output tile and walks K as (channel chunk
// of 32) x (9 taps):
//   * the (8+2) x (32+2) input halo of the chunk is brought into LDS ONCE (LDS-DMA, double buffered, the next chunk's
//     halo is requested in one burst at tap 0) and serves all nine taps -- a tap is just a different row offset into
//     the halo image;
//   * the [Cout tile][32] weight slab of each (chunk, tap) is streamed by LDS-DMA through a 4-slot ring, requested
//     three steps ahead;
//   * each wave holds a (TM x TN) grid of 32x32 v_mfma_f32_32x32x16_bf16 tiles; one barrier per (chunk, tap);
//   * <= 80 KB of LDS and <= 256 VGPRs, so that TWO workgroups share a CU: one computes while the other is in its
//     prologue (an HBM burst), its epilogue (64 KB of stores) or a barrier.
// Out-of-image halo pixels, the channel tail and partial tiles are zero-filled by the buffer bounds check, so the
// hot loop has no branches.  MODE 0: forward (weights [Cout][tap][Cin]); MODE 1: data gradient (weights
// [Cin][tap][Cout], taps mirrored).  Same argument block and fused epilogue as igemm.hip.
//
// History (DESIGN.md section 4 has the numbers): an 8-wave / 64-channel-chunk / one-workgroup-per-CU version of this
// kernel spent 23 % (304->128) to 54 % (64->240 data gradient) of a workgroup's life in prologue and epilogue with the
// matrix pipe idle; a two-phase "ping-pong" schedule of its two waves per SIMD did not help.  On this version,
// compile-time ablation (tools/ablate_conv.sh) shows DMA issue, fragment reads and MFMAs to be nearly additive even
// across the two co-resident workgroups (0.20 + 0.19 + 0.25 ms of a 0.67 ms launch), i.e. ~900 TFLOP/s is this
// structure's ceiling from HIP source.
#include "conv_common.h"
#include <cstdlib>

using namespace crdk;
bool crd_conv3x3p_applicable(const ConvK& k, int B);                                      // conv3x3p.hip
int crd_conv3x3p(const ConvK& k, int B, hipStream_t st, int col0, int col1, int tn);
long long crd_conv3x3p_partial_floats(const ConvK& k, int B);
int crd_conv3x3p_finalize(const ConvK& k, int B, hipStream_t st);

namespace {

constexpr int TH = 8, TW = 32;            // output tile (pixels)
constexpr int HW_ = TW + 2;               // halo width
constexpr int HROWS = (TH + 2) * HW_;     // 340 halo pixels
// (weight-slab ring size WS is a kernel template parameter: 4 = the slab of step t+3 is in flight while step t computes)

typedef __attribute__((address_space(3))) void* lds_ptr;

// -DCRD_CONV3_PROF: per-wave cycle split of the main loop of one workgroup (developer instrumentation, tools/bench_conv.py)
#ifdef CRD_CONV3_PROF
__device__ unsigned long long g_prof[4][8];
#define PROF_DECL unsigned long long pt0 = __builtin_readcyclecounter(), pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF_MARK(k)                                                                                  \
  {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
    unsigned long long t_ = __builtin_readcyclecounter();                                             \
    pacc[k] += t_ - pt0;                                                                              \
    pt0 = t_;                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  }
#else
#define PROF_DECL
#define PROF_MARK(k)
#endif
// -DCRD_CONV3_ABLATE=bits: parts of the main loop compiled out (timing experiments; results are wrong)
#ifdef CRD_CONV3_ABLATE
#define ABL(bit) ((CRD_CONV3_ABLATE) & (bit))
#else
#define ABL(bit) false
#endif

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int QK = 32;                      // channels per chunk: 64-byte LDS rows, 4 granules
constexpr int QNW = 4;                      // waves per workgroup
constexpr int QHG = (HROWS + 15) / 16;      // 22 halo DMA pieces (16 rows x 64 B each)
constexpr int QHPAD = QHG * 16;             // 352 rows allocated
constexpr int QHTAPS = 6;                   // halo pieces per wave: 6 x 4 waves = 24 >= 22 pieces

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
  const int b = blockIdx.z, n0 = a.col0 + blockIdx.y * BN;
  const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int H = a.IH, W = a.IW;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.x + (long long)b * a.x_bstride), 0, (int)(a.x_bstride * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const unsigned OOB = 0x80000000u;

  // DMA descriptors.  A piece (one wave instruction, 1 KiB) is 16 rows of 64 bytes; lane l stages row 16 G + (l>>2),
  // 16-byte slot l&3, which receives channel granule (l&3) ^ ((row>>2)&3) (source-side swizzle: the 16 rows a
  // ds_read_b128 lane group touches are distinct mod 16, so (row&3, slot) -- the 16-byte bank slot -- is distinct too).
  // The per-step request code is kept to a handful of instructions (a lone wave issues one instruction per ~4-5 cycles
  // and this code is on the step's critical path): per-lane byte offsets are loop constants -- pixel (or weight row) plus
  // the lane's channel granule, out-of-range lanes flagged out of bounds -- and the (chunk, tap) part of the address is
  // the instruction's scalar offset.  The scalar offset is not bounds-checked, so the channel tail of the last chunk is
  // the one per-step lane test left.
  unsigned hvo[QHTAPS];
  int hch[QHTAPS];
#pragma unroll
  for (int s = 0; s < QHTAPS; ++s) {
    const int G = QNW * s + wv;
    const int hr = 16 * G + (l >> 2);
    const int hy = hr / HW_, hx = hr - hy * HW_;
    const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
    const bool ok = G < QHG && hr < HROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    hch[s] = ((l & 3) ^ ((hr >> 2) & 3)) * 8;
    hvo[s] = ok ? (unsigned)(((iy * W + ix) * a.x_ld + hch[s]) * 2) : OOB;
  }
  // weight slab: wave w stages pieces g = 4 j + w: rows n = 16 g + (l>>2); (n>>2)&3 = (l>>4)&3 for every g
  const int wch = ((l & 3) ^ ((l >> 4) & 3)) * 8;
  unsigned wvo[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int g = QNW * j + wv;
    const int n = 16 * g + (l >> 2), ng = n0 + n;
    wvo[j] = (g < WGROUPS && ng < a.Cout) ? (unsigned)((ng * a.Ktot + wch) * 2) : OOB;
  }
  const int Cin = a.Cin;
  const int nChunks = (Cin + QK - 1) / QK;

  // halo piece s of `chunk` (all of this wave's pieces are requested in one burst, see the loop); a chunk past the end:
  // zero-fill dummy into the landing area
  auto stage_halo_piece = [&](int s, int chunk, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool real = chunk < nChunks && QNW * s + wv < QHG;
    const int tail = Cin - chunk * QK;                       // channels left from this chunk on
    const unsigned vo = (real && hch[s] < tail) ? hvo[s] : OOB;
    bf16_t* dst = real ? sH + buf * QHPAD * QK + (QNW * s + wv) * 16 * QK : sD;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)dst, 16, vo, chunk * QK * 2, 0, 0);
#else
    (void)s; (void)chunk; (void)buf;
#endif
  };
  // slab of (chunk, tap); past the end: the same request for chunk 0 (valid addresses, lands in a ring slot nobody
  // reads any more)
  auto stage_weights = [&](int chunk, int tap, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int c = chunk < nChunks ? chunk : 0;
    const int tail = Cin - c * QK;
    const bool lane_ok = wch < tail;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int g = QNW * j + wv;
      bf16_t* dst = g < WGROUPS ? sW + slot * BN * QK + 16 * g * QK : sD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, lane_ok ? wvo[j] : OOB, (tap * Cin + c * QK) * 2, 0, 0);
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

  // prologue: halo of chunk 0 and slab 0, then slabs 1..D-1
#pragma unroll
  for (int s = 0; s < QHTAPS; ++s) stage_halo_piece(s, 0, 0);
  stage_weights(0, 0, 0);
#pragma unroll
  for (int s = 1; s < D; ++s) stage_weights(0, s, s);
  wait_vm<(D - 1) * WJ>();
  asm volatile("s_barrier" ::: "memory");

  // DMA schedule.  vmcnt retires in order, so a slow request (the halo comes from HBM, the slabs from L2) holds up the
  // counted wait of everything issued after it: spread over the taps, the halo pieces made six of the nine steps of a
  // chunk wait on HBM latency (257 cycles per step on average).  They are now requested in one burst at tap 0, behind
  // that step's slab, and first waited for at the end of tap 3; taps 4..8 wait on L2 only.
  int pc = 0, pt = D;                      // (chunk, tap) of the slab to request: step + D
  int wb = 0, wnext = D % WS;
  PROF_DECL
  for (int chunk = 0; chunk < nChunks; ++chunk) {
    const int hb = chunk & 1;
    const bool two = Cin - chunk * QK > 16;        // second k-step of the chunk holds channels
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (!ABL(256)) {
      stage_weights(pc, pt, wnext);
      if (tap == 0) {
#pragma unroll
        for (int s = 0; s < QHTAPS; ++s) stage_halo_piece(s, chunk + 1, hb ^ 1);
      }
      }
      if (++pt == 9) { pt = 0; ++pc; }
      PROF_MARK(0)   // DMA issue
      const int ky = tap / 3, kx = tap - ky * 3;
      const int oy = (MODE == 0) ? ky : 2 - ky, ox = (MODE == 0) ? kx : 2 - kx;
      const bf16_t* hbase = sH + hb * QHPAD * QK;
      const bf16_t* wbase = sW + wb * BN * QK;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 0 || two) {
          bf16x8 af[TM], bfr[TN];
          const int gi = ks * 2 + (l >> 5);
          if (ABL(64)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = bf16x8{1, 1, 2, 3, 4, 5, 6, 7};
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = bf16x8{1, 1, 2, 3, 4, 5, 6, 7};
          } else {
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
          }
          PROF_MARK(1 + 2 * ks)   // fragment reads (returned)
          if (!ABL(128)) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j][0] += (float)af[i][0] * (float)bfr[j][0];
          }
          PROF_MARK(2 + 2 * ks)   // MFMA issue
        }
      }
      // the slab of step+1 (requested at step+1-D) has landed: everything older than the requests of the last D-1 steps,
      // which include the halo burst while tap <= D-1
      if (!ABL(16)) {
      if (tap <= D - 1) wait_vm<(D - 1) * WJ + QHTAPS>();
      else wait_vm<(D - 1) * WJ>();
      }
      PROF_MARK(5)   // wait for the next slab
      if (!ABL(32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PROF_MARK(6)   // barrier
      wb = wb + 1 == WS ? 0 : wb + 1;
      wnext = wnext + 1 == WS ? 0 : wnext + 1;
    }
  }
  wait_vm<0>();
  asm volatile("s_barrier" ::: "memory");
#ifdef CRD_CONV3_PROF
  if (blockIdx.x == 100 && blockIdx.y == 0 && blockIdx.z == 0 && l == 0)
    for (int k = 0; k < 8; ++k) g_prof[wv][k] = pacc[k];
#endif

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
int launch3(const ConvK& k0, int B, hipStream_t st, long long partial_cap, int col0 = 0, int col1 = -1, bool finalize = true) {
  constexpr int BN = WN * TN * 32, BM = TH * TW;
  ConvK k = k0;
  k.col0 = col0;
  k.lds_bytes = 0;                                 // (the fp32 staging epilogue is an igemm.hip path)
  if (col1 < 0) col1 = k.Cout;                     // this launch computes output columns [col0, col1)
  k.n_tiles = cdiv(k.IW, TW) * cdiv(k.IH, TH);
  if ((long long)B * k.n_tiles * k.G16 * 2 > partial_cap) k.stats_partial = nullptr;
  const size_t lds_main = (size_t)(2 * QHPAD * QK + WS * BN * QK + 16 * QK) * sizeof(bf16_t);
  const size_t lds_epi = (size_t)BM * (BN + 8) * sizeof(bf16_t) + 4096;     // conv_epilogue's staging tile + fold scratch
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  const int tiles_x = cdiv(k.IW, TW), tiles_y = cdiv(k.IH, TH);
  dim3 grid(tiles_x * tiles_y, cdiv(col1 - col0, BN), B);
  static bool attr_done[2] = {false, false};
  if (k.gather_mode == 0) {
    if (!attr_done[0]) {
      crd_reserve_lds(reinterpret_cast<const void*>(&k_conv3x3<WM, WN, TM, TN, 0, WS>), (int)lds, "k_conv3x3");
      attr_done[0] = true;
    }
    hipLaunchKernelGGL((k_conv3x3<WM, WN, TM, TN, 0, WS>), grid, dim3(256), lds, st, k, tiles_x);
  } else {
    if (!attr_done[1]) {
      crd_reserve_lds(reinterpret_cast<const void*>(&k_conv3x3<WM, WN, TM, TN, 1, WS>), (int)lds, "k_conv3x3");
      attr_done[1] = true;
    }
    hipLaunchKernelGGL((k_conv3x3<WM, WN, TM, TN, 1, WS>), grid, dim3(256), lds, st, k, tiles_x);
  }
  if (finalize && k.stats && k.stats_partial)
    hipLaunchKernelGGL(k_stats_finalize, dim3(k.G16, B), dim3(256), 0, st, k.stats_partial, k.n_tiles, k.G16, k.stats);
  CRD_LAUNCH_CHECK("crd_conv_igemm(3x3 halo)");
  return CRD_OK;
}

}  // namespace

#ifdef CRD_CONV3_PROF
extern "C" int crd_dbg_conv3_prof(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof));
}
#endif

static int g_small_thr = -1;      // workgroup count below which narrower column tiles are used (crd_tune_conv3x3_small_grid)

extern "C" int crd_tune_conv3x3_small_grid(int workgroups) {
  const int old = g_small_thr < 0 ? 512 : g_small_thr;
  g_small_thr = workgroups < 0 ? 512 : workgroups;
  return old;
}

// Called from crd_conv_igemm for 3x3 / stride 1 / pad 1 layers on grids large enough to fill the chip.
int crd_conv3x3_halo(const ConvK& k, int B, hipStream_t st, long long partial_cap) {
  // persistent one-wave-per-SIMD kernel (conv3x3p.hip) for the 128-column tiles (96 / 64 columns: one narrower tile).  Its
  // GroupNorm sums need the caller's partial-sum buffer; without one (or a too small one) the launch stays here.
  // (a launch WITH GroupNorm sums whose column count leaves a ragged tail of <= 64 columns -- e.g. Cout 160 / 192 / 288; no
  // layer of the model -- would need its sums split over two kernels: it stays with the two-workgroup kernel below)
  const int rest128 = k.Cout > 96 ? k.Cout % 128 : 0;
  const bool ragged_stats = k.stats && rest128 > 0 && rest128 <= 64;
  if (crd_conv3x3p_applicable(k, B) && !ragged_stats &&
      (!k.stats || (k.stats_partial && crd_conv3x3p_partial_floats(k, B) <= partial_cap))) {
    // a ragged tail of <= 64 columns (data gradients towards 136 / 144 / 296 / 304 channels) goes to the narrow tiles here
    const int N = k.Cout;
    int rc;
    if (N <= 64) rc = crd_conv3x3p(k, B, st, 0, N, 2);
    else if (N <= 96) rc = crd_conv3x3p(k, B, st, 0, N, 3);
    else {
      const int full = N / 128 * 128, rest = N - full;
      // how the ragged tail of a data gradient is cut (CRD_C3P_SPLIT, developer switch: 0 = 128-wide tiles + the two-workgroup
      // kernel for the tail; bits 1 / 2 below: 365.7 -> 354.8 us and 594 -> 584 us on the 256 x 416 level, 3.74 ms -> 3.71 ms per step)
      static int split_mode = -1;
      if (split_mode < 0) split_mode = crd_dev_int("CRD_C3P_SPLIT", 3);
      if (rest == 0 || rest > 64) rc = crd_conv3x3p(k, B, st, 0, N, 4);
      else if ((split_mode & 1) && full == 128) {    // 136 / 144 columns: 96 + 40 / 48 on the persistent kernel's 96- and 64-wide tiles
        rc = crd_conv3x3p(k, B, st, 0, 96, 3);
        if (rc != CRD_OK) return rc;
        return crd_conv3x3p(k, B, st, 96, N, 2);
      } else if ((split_mode & 2) && rest > 32) {    // 296 / 304 columns: the 40 / 48-column tail on the persistent kernel's 64-wide tile
        rc = crd_conv3x3p(k, B, st, 0, full, 4);
        if (rc != CRD_OK) return rc;
        return crd_conv3x3p(k, B, st, full, N, 2);
      } else {
        rc = crd_conv3x3p(k, B, st, 0, full, 4);
        if (rc != CRD_OK) return rc;
        ConvK kt = k;
        kt.stats_partial = nullptr;
        return rest <= 32 ? launch3<4, 1, 2, 1>(kt, B, st, 0, full, N, true) : launch3<4, 1, 2, 2>(kt, B, st, 0, full, N, true);
      }
    }
    if (rc != CRD_OK || !k.stats) return rc;
    return crd_conv3x3p_finalize(k, B, st);
  }
  {   // grids that leave most CUs without a workgroup (the 32x52 decoder level: 64 pixel tiles): narrower column tiles
    if (g_small_thr < 0) g_small_thr = crd_dev_int("CRD_CONV3_SMALL", 512);
    const int thr = g_small_thr;
    const long long tiles = (long long)cdiv(k.IW, TW) * cdiv(k.IH, TH) * B;
    if (tiles * cdiv(k.Cout, 128) < thr && k.Cout > 32) {
      if (tiles * cdiv(k.Cout, 64) >= thr) return launch3<4, 1, 2, 2>(k, B, st, partial_cap);
      return launch3<4, 1, 2, 1>(k, B, st, partial_cap);
    }
  }
  if (k.Cout <= 32) return launch3<4, 1, 2, 1>(k, B, st, partial_cap);
  if (k.Cout <= 64) return launch3<4, 1, 2, 2>(k, B, st, partial_cap);
  if (k.Cout <= 96) return launch3<4, 1, 2, 3>(k, B, st, partial_cap);
  // 129..160 and 257..320 output channels (data gradients towards 136 / 144 / 296 / 304-channel concat inputs):
  if ((k.Cout > 128 && k.Cout <= 160) || (k.Cout > 256 && k.Cout <= 320)) {
    // two launches: 128-wide tiles, then the remaining 16..64 columns with a narrow tile (a 160-wide tile needs more
    // than half a CU's LDS or registers: an 8-wave 160-column kernel was 8-10 % slower than this pair)
    const int cut = k.Cout > 256 ? 256 : 128, rest = k.Cout - cut;
    int rc = launch3<4, 1, 2, 4>(k, B, st, partial_cap, 0, cut, false);
    if (rc != CRD_OK) return rc;
    if (rest <= 32) return launch3<4, 1, 2, 1>(k, B, st, partial_cap, cut, k.Cout, true);
    return launch3<4, 1, 2, 2>(k, B, st, partial_cap, cut, k.Cout, true);
  }
  return launch3<4, 1, 2, 4>(k, B, st, partial_cap);     // (64 x 128 per wave; 128 x 64 spills more and is 6 % slower)
}
