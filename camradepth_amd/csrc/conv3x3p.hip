// Persistent halo-tile 3x3 convolution (stride 1, pad 1) on MFMA for gfx950: the decoder's big ShortResBlock convs at
// the 128x208 and 256x416 levels (src/utils/utils.py:114-124,211), forward and data gradient.
//
// k_conv3x3 (conv3x3.hip) runs two 4-wave workgroups per CU at 256 VGPRs each, one 8 x 32-pixel tile per workgroup: every
// workgroup pays its own prologue (first halo + slabs) and epilogue, hidden only by the other workgroup of the CU.  Here:
//   * ONE persistent workgroup per CU (8 waves = two per SIMD by default, 4 with CRD_CONV3P_WAVES=4: measured the same) walks
//     over 16 x 32 = 512-pixel tiles: one weight slab serves twice the pixels, halo overhead 1.20 instead of 1.33, and the
//     512 x 128 fp32 accumulator tile is exactly the CU's 256 KB of accumulation registers;
//   * the weight-slab ring simply keeps running across tiles and the first halo of the next tile is requested during the
//     last channel chunk of the current one: no exposed prologue (round 1's one-workgroup-per-CU kernel lost 23-54 % there);
//   * each (chunk, tap) step is two half-steps of 16 channels; the fragment reads of the next half-step are issued before the
//     MFMAs of the current one (register double buffer), pinned between the MFMAs with sched_group_barrier;
//   * MFMA operands are swapped (A = weight rows), so a lane holds ONE pixel and 4-channel runs of it; v_permlane32_swap
//     between the half-waves makes 8 consecutive channels = one 16-byte store per lane straight from the accumulators: no LDS
//     staging, no barrier, nothing in flight is waited for.  Accumulate mode = all of a row's loads, one wait, then the stores;
//   * GroupNorm sums: per (tile, wave) partial rows + k_stats_finalize (atomics from 256 synchronised workgroups: +0.4 ms).
// Same K walk as k_conv3x3: (32-channel chunk) x (9 taps), halo of a chunk in LDS once (double buffered), weight slab
// of every (chunk, tap) through a ring requested D steps ahead, counted vmcnt waits, one raw s_barrier per half-step.
// Plain bf16 output (store or accumulate) with optional GroupNorm sums -- everything the decoder's 3x3 ConvLayers and
// their data gradients need; bias / activation / fp32 / residual epilogues stay with k_conv3x3.
// What bounds it (profiles/r02_pmc_kernels.md): MFMA pipe busy 59 % of the clocks at an
// effective 1.5 GHz; ring depth, wave count, barriers, waits, the weights' path (registers + ds_write instead of LDS-DMA),
// XCD-aware tile order, non-temporal stores and rolled tap loops were all built and measured the same (round 2; the
// experiment code is in the history of this file up to commit 871691a, not here).
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>

using namespace crdk;

namespace {

constexpr int TH = 16, TW = 32;            // output tile (pixels)
constexpr int HW_ = TW + 2;                // halo width
constexpr int HROWS = (TH + 2) * HW_;      // 612 halo pixels
constexpr int QK = 32;                     // channels per chunk: 64-byte LDS rows
constexpr int HG = (HROWS + 15) / 16;      // 39 halo DMA pieces (16 rows x 64 B)
constexpr int HPAD = HG * 16;              // 624 rows allocated per halo buffer
// NW waves per workgroup (template parameter): 4 = one wave per SIMD with the whole register file (TM = 4 row tiles of 32
// pixels per wave), 8 = two waves per SIMD at 256 registers each (TM = 2): the second wave's MFMAs fill the issue slots the
// first one spends on DMA requests, fragment reads and waits.

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef __attribute__((ext_vector_type(4))) unsigned u32x4w;
#ifndef CRD_C3P_WS
#define CRD_C3P_WS 5
#endif
constexpr int C3P_WS = CRD_C3P_WS;                    // weight-slab ring slots (prefetch distance WS - 1 steps)

template <int TN, int MODE, int WS, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_conv3x3p(ConvK a, int tiles_x, int tiles_y, int tiles_total) {
  constexpr int TM = TH / NW;                     // 32-pixel row tiles per wave (rows TM w .. TM w + TM - 1 of the tile)
  constexpr int HT = (HG + NW - 1) / NW;          // halo pieces per wave
  constexpr int D = WS - 1;
  static_assert(D >= 2 && D <= 8, "slab prefetch distance");
  constexpr int BN = TN * 32;
  constexpr int WGROUPS = BN / 16;                // weight-slab DMA pieces (16 rows x 64 B)
  constexpr int WJ = (WGROUPS + NW - 1) / NW;     // pieces per wave
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sH = lds;                               // [2][HPAD][QK]
  bf16_t* sW = sH + 2 * HPAD * QK;                // [WS][BN][QK]
  bf16_t* sD = sW + WS * BN * QK;                 // [16][QK] landing area of the zero-fill dummies

  const int t = threadIdx.x, l = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = a.col0 + blockIdx.y * BN;
  const int H = a.IH, W = a.IW, Cin = a.Cin;
  const int nChunks = (Cin + QK - 1) / QK;
  const unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);

  // weight slab: wave w stages pieces g = 4 j + w: rows n = 16 g + (l>>2), 16-byte slot l&3 <- granule (l&3) ^ ((n>>2)&3)
  const int wch = ((l & 3) ^ ((l >> 4) & 3)) * 8;
  unsigned wvo[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int g = NW * j + wv;
    const int n = 16 * g + (l >> 2), ng = n0 + n;
    wvo[j] = (g < WGROUPS && ng < a.Cout) ? (unsigned)((ng * a.Ktot + wch) * 2) : OOB;
  }
  auto stage_weights = [&](int chunk, int tap, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tail = Cin - chunk * QK;
    const bool lane_ok = wch < tail;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int g = NW * j + wv;
      bf16_t* dst = g < WGROUPS ? sW + slot * BN * QK + 16 * g * QK : sD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)dst, 16, lane_ok ? wvo[j] : OOB,
                                               (tap * Cin + chunk * QK) * 2, 0, 0);
    }
#else
    (void)chunk; (void)tap; (void)slot;
#endif
  };
  // halo pieces of one tile: lane l of piece G stages halo row 16 G + (l>>2), slot l&3 <- granule (l&3) ^ ((row>>2)&3)
  unsigned hvo[HT];
  auto halo_offsets = [&](int tile) {             // tile >= tiles_total: everything out of range (zero fill)
    const int bb = tile / (tiles_x * tiles_y), rem = tile - bb * (tiles_x * tiles_y);
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    int lh = l;                                       // opaque: the per-lane (hy, hx) of every piece must not be hoisted out of
    asm volatile("" : "+v"(lh));                      // the tile loop (they get spilled, and a reload waits vmcnt(0))
#pragma unroll
    for (int s = 0; s < HT; ++s) {
      const int G = NW * s + wv;
      const int hr = 16 * G + (lh >> 2);
      const int hy = hr / HW_, hx = hr - hy * HW_;
      const int iy = tyi * TH - 1 + hy, ix = txi * TW - 1 + hx;
      const bool ok = tile < tiles_total && G < HG && hr < HROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int ch = ((lh & 3) ^ ((hr >> 2) & 3)) * 8;
      hvo[s] = ok ? (unsigned)(((iy * W + ix) * a.x_ld + ch) * 2) : OOB;
    }
    return bb;
  };
  auto stage_halo = [&](const __amdgpu_buffer_rsrc_t& rx, int chunk, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tail = Cin - chunk * QK;
#pragma unroll
    for (int s = 0; s < HT; ++s) {
      const int G = NW * s + wv;
      const int ch = ((l & 3) ^ (((16 * G + (l >> 2)) >> 2) & 3)) * 8;
      const bool real = G < HG;
      bf16_t* dst = real ? sH + buf * HPAD * QK + G * 16 * QK : sD;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)dst, 16, (real && ch < tail) ? hvo[s] : OOB, chunk * QK * 2, 0, 0);
    }
#else
    (void)rx; (void)chunk; (void)buf;
#endif
  };
  auto make_rx = [&](int bb) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long long)bb * a.x_bstride), 0, (int)(a.x_bstride * 2), 0x00020000);
  };

  // fragment reads of half-step ks of (halo buffer hb, tap, ring slot wb)
  auto read_frags = [&](int hb, int tap, int wb, int ks, bf16x8 (&af)[TM], bf16x8 (&bfr)[TN]) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int oy = (MODE == 0) ? ky : 2 - ky, ox = (MODE == 0) ? kx : 2 - kx;
    const bf16_t* hbase = sH + hb * HPAD * QK;
    const bf16_t* wbase = sW + wb * BN * QK;
    const int gi = ks * 2 + (l >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int hr = (wv * TM + i + oy) * HW_ + (l & 31) + ox;
      af[i] = *reinterpret_cast<const bf16x8*>(hbase + hr * QK + ((gi ^ ((hr >> 2) & 3)) << 3));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = j * 32 + (l & 31);
      bfr[j] = *reinterpret_cast<const bf16x8*>(wbase + row * QK + ((gi ^ ((row >> 2) & 3)) << 3));
    }
  };

  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // (plain tile order: an XCD-aware order -- the 32 workgroups of an XCD on neighbouring tiles -- measured the same)
  auto map_tile = [&](int q) { return q; };
  int q = blockIdx.x;
  if (q >= tiles_total) return;
  int tile = map_tile(q);
  // ---- prologue: halo of the first tile's chunk 0, slabs of steps 0 .. D-1 ----
  int b = __builtin_amdgcn_readfirstlane(halo_offsets(tile));
  __amdgpu_buffer_rsrc_t rx = make_rx(b);
  stage_halo(rx, 0, 0);
  int pc = 0, pt = 0;                                 // (chunk, tap) of the next slab to request; chunk wraps per tile
#pragma unroll
  for (int s = 0; s < D; ++s) {
    stage_weights(pc, pt, s);
    if (++pt == 9) { pt = 0; if (++pc == nChunks) pc = 0; }
  }
  wait_vm<(D - 1) * WJ>();                            // halo + slab 0 have landed
  asm volatile("s_barrier" ::: "memory");
  int gchunk = 0;                                     // chunks done so far (all tiles): parity = halo buffer
  int wb = 0, wnext = D % WS;
  bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
  read_frags(0, 0, 0, 0, a0, b0);

  for (; q < tiles_total; q += gridDim.x) {
    tile = map_tile(q);
    const int tile_next = map_tile(q + gridDim.x);      // >= tiles_total when there is none (its halo requests read zeros)
    const int rem = tile - b * (tiles_x * tiles_y);
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int ty0 = tyi * TH, tx0 = txi * TW;
    zero_acc();
    for (int chunk = 0; chunk < nChunks; ++chunk, ++gchunk) {
      const int hb = gchunk & 1;
      const bool last = chunk + 1 == nChunks;
      // one (chunk, tap) step; TWO: the chunk's second 16 channels hold data (false only for the 8- / 16-channel tail of
      // Cin = 136 / 144 / 296 / 304).  The LDS reads are never conditional: with a branch around reads in the loop the
      // compiler's lgkmcnt bookkeeping takes the worst case at the join and stalls the first MFMAs of every half-step on
      // the reads issued for the next one (and two copies of the loop body spill 227 VGPRs).
      // MFMAs and everything else of a half-step in ONE scheduling region, with the issue order pinned: an in-order wave
      // that issues its 16 MFMAs back to back spends 512 cycles doing only that, and its loads, address arithmetic and DMA
      // requests ADD to them (measured 2125 cycles per step against 1024 of MFMA issue); placed in the gaps between MFMAs
      // (a wave can issue ~5 other instructions per 32-cycle MFMA for free) they cost nothing.
      auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < TM * TN; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // one MFMA
          if (k < TM + TN) {
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);           // address arithmetic of a fragment read
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // the read
          } else if (k < TM + TN + WJ) {
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // a slab request (VMEM read)
          }
        }
      };
      // one (chunk, tap) step.  Both 16-channel halves of the chunk are always multiplied (the 8- / 16-channel tails of
      // Cin = 136 / 144 / 296 / 304 are zero-filled): a branch around the second half's MFMAs would split the scheduling
      // region, and a branch around its reads makes the compiler's lgkmcnt bookkeeping stall the next MFMAs.
      auto step = [&](int tap) __attribute__((always_inline)) {
        // ---- at tap 0 the next halo (next chunk, or chunk 0 of the next tile) ----
        if (tap == 0) {
          if (last) {                                  // the ring keeps running: first halo of the workgroup's next tile
            const int nb = __builtin_amdgcn_readfirstlane(halo_offsets(tile_next));
            if (tile_next < tiles_total) { b = nb; rx = make_rx(b); }
            stage_halo(rx, 0, hb ^ 1);
          } else {
            stage_halo(rx, chunk + 1, hb ^ 1);
          }
        }
        // ---- half-step 0: MFMAs on (a0, b0); in their shadow the slab request of step + D and the reads of half-step 1 ----
        stage_weights(pc, pt, wnext);
        if (++pt == 9) { pt = 0; if (++pc == nChunks) pc = 0; }
        read_frags(hb, tap, wb, 1, a1, b1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
        interleave();
        // the slab of step+1 (requested D steps before it) has landed: all but the requests of the last D-1 steps (this
        // step's included), which include this chunk's halo burst -- issued BEFORE this step's slab request -- while tap <= D-2
        if (tap <= D - 2) wait_vm<(D - 1) * WJ + HT>();
        else wait_vm<(D - 1) * WJ>();
        const int wbn = wb + 1 == WS ? 0 : wb + 1;
        // (the builtin, not inline asm: the compiler's own lgkmcnt bookkeeping must see that nothing is pending here)
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): (a1, b1) have arrived, all of this wave's LDS accesses are done
        __builtin_amdgcn_s_barrier();
        wb = wbn;
        wnext = wnext + 1 == WS ? 0 : wnext + 1;
        // ---- half-step 1: MFMAs on (a1, b1); in their shadow the reads of the NEXT step's half-step 0 ----
        const int wrap = tap == 8;
        read_frags(hb ^ wrap, wrap ? 0 : tap + 1, wb, 0, a0, b0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
        interleave();
      };
      // (tap loops not unrolled: with nine copies the compiler keeps every tap's fragment addresses live -- 345 spilled
      // VGPRs; the address arithmetic of a step, ~60 VALU instructions, hides under its 32 MFMAs)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) step(tap);
    }
    // ---- epilogue of this wave's 4 x 32 pixels x BN columns, straight from the accumulators ----
    // The MFMA operands are swapped (A = weight rows, B = pixels), so a lane holds ONE pixel (column l&31) and, per 32 x 32
    // tile, the output channels (r&3) + 8 (r>>2) + 4 (l>>5): four runs of 4 consecutive channels.  v_permlane32_swap
    // between the two half-waves (which hold the same pixel) turns two such runs into 8 consecutive channels = one
    // 16-byte store per lane (the CDNA guide's T21).  No LDS staging (the first version's staging strip + barrier-free
    // per-wave copy cost 0.6 ms of a 0.9 ms launch whose main loop takes 0.28 ms), no wait on anything in flight.
    {
      // lane-constant epilogue values must not be hoisted out of the tile loop (they would be spilled across the main loop
      // and every reload waits vmcnt(0), behind the DMA queue): an opaque copy of the lane id makes them per-tile values
      int le = l;
      asm volatile("" : "+v"(le));
      const int half = le >> 5, px = le & 31;
      const int bt = (tile - rem) / (tiles_x * tiles_y);                 // image of THIS tile (b may already be the next one's)
      bf16_t* yb = reinterpret_cast<bf16_t*>(a.y) + (long long)bt * a.y_bstride;
      float s[TN][2], ss[TN][2];                                          // GroupNorm sums of this lane's pixel, per 16-channel slab
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
          __builtin_amdgcn_sched_barrier(0);          // one 32 x 32 tile at a time (hoisting every accumulator read spills)
          uint32_t d[4][2];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            d[g4][0] = pack_bf2(acc[i][j][4 * g4], acc[i][j][4 * g4 + 1]);
            d[g4][1] = pack_bf2(acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
            if (pok) {
              const float v0 = bf_lo(d[g4][0]), v1 = bf_hi(d[g4][0]), v2 = bf_lo(d[g4][1]), v3 = bf_hi(d[g4][1]);
              s[j][g4 >> 1] += (v0 + v1) + (v2 + v3);
              ss[j][g4 >> 1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            }
          }
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            // runs 2 pr (channels 16 pr + 4 half ..) and 2 pr + 1 (16 pr + 8 + 4 half ..): after the swaps the lower half-wave
            // holds channels 16 pr .. 16 pr + 7 of its pixel, the upper one channels 16 pr + 8 .. 16 pr + 15
            auto r0 = __builtin_amdgcn_permlane32_swap(d[2 * pr][0], d[2 * pr + 1][0], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(d[2 * pr][1], d[2 * pr + 1][1], false, false);
            u[j][pr] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pok) {
          // read-modify-write (gradient accumulation): ALL of the row's loads first, one wait, then the stores -- a load
          // inside the per-store branch made the compiler wait vmcnt(0) before every single store
          if (a.accumulate) {
            // (one 16-channel half of every 32-column tile at a time: all TN x 2 old values live at once cost two spilled
            // VGPRs in the 128-column variant, and their reload at the top of every tile waits vmcnt(0) behind the ring)
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
              if (n0 + j * 32 + pr * 16 + half * 8 < a.Cout) {
                *reinterpret_cast<uint4*>(row + j * 32 + pr * 16) = u[j][pr];
              }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (a.stats) {
        // GroupNorm sums of this wave's part of the tile as one row of plain stores: [image][tile][wave][G16][2], summed by
        // k_stats_finalize.  (Atomics straight into stats: the persistent workgroups reach their epilogues together, and
        // 1024 waves x 16 atomics on the same few cache lines stalled every wave's next vmcnt wait -- 0.4 ms of a 1.1 ms launch.)
        float* prow = a.stats_partial + ((((long long)bt * (tiles_x * tiles_y) + rem) * NW + wv) * a.G16) * 2;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) {
            // a lane's runs of slab sl are channels 16 sl + 4 half + {0..3, 8..11}: both half-waves feed both slabs
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
      // the first fragments of the next tile are read (again) here, so that (a0, b0) need not survive the epilogue:
      // 32 more live VGPRs there made the compiler spill, and a spill reload waits vmcnt(0) -- behind the tile's stores
      __builtin_amdgcn_sched_barrier(0);
      read_frags(gchunk & 1, 0, wb, 0, a0, b0);
    }
  }
  wait_vm<0>();
}

template <int TN, int WS, int NW>
int launch_p(const ConvK& k0, int B, hipStream_t st, int col0, int col1) {
  constexpr int BN = TN * 32;
  ConvK k = k0;
  k.col0 = col0;
  const int tiles_x = cdiv(k.IW, TW), tiles_y = cdiv(k.IH, TH);
  const int tiles_total = tiles_x * tiles_y * B;
  const int gy = cdiv(col1 - col0, BN);
  int gx = 256 / gy;
  if (gx < 1) gx = 1;
  if (gx > tiles_total) gx = tiles_total;
  const size_t lds = (size_t)(2 * HPAD * QK + WS * BN * QK + 16 * QK) * sizeof(bf16_t);
  static bool attr_done[2] = {false, false};
  const int m = k.gather_mode == 0 ? 0 : 1;
  if (m == 0) {
    if (!attr_done[0]) { crd_reserve_lds(reinterpret_cast<const void*>(&k_conv3x3p<TN, 0, WS, NW>), (int)lds, "k_conv3x3p"); attr_done[0] = true; }
    hipLaunchKernelGGL((k_conv3x3p<TN, 0, WS, NW>), dim3(gx, gy), dim3(64 * NW), lds, st, k, tiles_x, tiles_y, tiles_total);
  } else {
    if (!attr_done[1]) { crd_reserve_lds(reinterpret_cast<const void*>(&k_conv3x3p<TN, 1, WS, NW>), (int)lds, "k_conv3x3p"); attr_done[1] = true; }
    hipLaunchKernelGGL((k_conv3x3p<TN, 1, WS, NW>), dim3(gx, gy), dim3(64 * NW), lds, st, k, tiles_x, tiles_y, tiles_total);
  }
  CRD_LAUNCH_CHECK("crd_conv_igemm(3x3 persistent)");
  return CRD_OK;
}

// rows of GroupNorm partial sums a launch writes per image (crd_conv_desc.stats_partial must hold B x rows x Cout/16 x 2 floats)
int waves_per_wg() {
  static int nw = -1;
  if (nw < 0) nw = crd_dev_int("CRD_CONV3P_WAVES", 8) == 4 ? 4 : 8;
  return nw;
}
inline long long partial_rows(const ConvK& k) { return (long long)cdiv(k.IW, TW) * cdiv(k.IH, TH) * waves_per_wg(); }

}  // namespace

// Can the persistent kernel take (part of) this launch?  Plain bf16 store / accumulate with optional GroupNorm sums, on
// grids with enough 16 x 32 tiles to occupy the chip.
bool crd_conv3x3p_applicable(const ConvK& k, int B) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CRD_CONV3P"); on = e ? atoi(e) : 1; }     // (a product switch: tests/test_gpu_igemm.py compares both kernels)
  if (!on) return false;
  const long long tiles = (long long)cdiv(k.IW, TW) * cdiv(k.IH, TH) * B;
  return !k.y_f32 && !k.bias && !k.act && !k.res && k.out_mode == 0 && k.vec_ok && (k.y_ld & 7) == 0 && (k.Cout & 7) == 0 &&
         k.red_x == nullptr && !k.chan && tiles >= 192 && k.Cout >= 64;
}

// GroupNorm sums go through per-(tile, wave) partial rows: floats the caller's stats_partial buffer must hold
long long crd_conv3x3p_partial_floats(const ConvK& k, int B) { return (long long)B * partial_rows(k) * k.G16 * 2; }
int crd_conv3x3p_finalize(const ConvK& k, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_stats_finalize, dim3(k.G16, B), dim3(256), 0, st, k.stats_partial, (int)partial_rows(k), k.G16, k.stats);
  CRD_LAUNCH_CHECK("crd_conv_igemm(3x3 persistent, statistics)");
  return CRD_OK;
}

// Output columns [col0, col1) in tiles of TN x 32 (col1 - col0 a multiple of the tile except for a masked last tile)
int crd_conv3x3p(const ConvK& k, int B, hipStream_t st, int col0, int col1, int tn) {
  if (waves_per_wg() == 4) {
    if (tn == 4) return launch_p<4, C3P_WS, 4>(k, B, st, col0, col1);
    if (tn == 3) return launch_p<3, C3P_WS, 4>(k, B, st, col0, col1);
    return launch_p<2, C3P_WS, 4>(k, B, st, col0, col1);
  }
  if (tn == 4) return launch_p<4, C3P_WS, 8>(k, B, st, col0, col1);
  if (tn == 3) return launch_p<3, C3P_WS, 8>(k, B, st, col0, col1);
  return launch_p<2, C3P_WS, 8>(k, B, st, col0, col1);
}
