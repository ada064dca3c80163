// Persistent encoder stage for gfx950: every Block of one SimplifiedTransformer stage in ONE launch
// (reference: Block.forward src/models/simplified_attention.py:141-145, Attention_MaxPool.forward :90-109, Mlp.forward
// :34-43, DWConv.forward :318-323, the stage loops of forward_features :265-306).  API and rationale: include/camradepth_hip.h
// (crd_enc_stage_fwd).
//
// Decomposition.  A sample is owned by G = H / RPW workgroups (RPW = 1 image row each while B x H <= 256, else 2); workgroup g keeps image rows RPW g .. RPW g + RPW - 1 of the
// stage tensor in LDS for the whole launch:
//   sX   fp32 [2W][C]        the residual stream x -> x1 -> x2 (updated in place)
//   sXN  bf16 [.][C + 8]     Block.norm1(x), later Block.norm2(x1): the activation operand of q / sr / k / fc1
//   sQ, sK(, sKRN)           q of the own pixels, k of ALL keys of the sample (and attn.norm(sr(x)) of all keys)
//   sH   bf16 [2W][hid + 8]  fc1 output -> Mlp.norm1 -> depthwise 3x3 -> Mlp.norm2 + GELU, all in place (overlays sQ / sK)
// GEMMs: v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand (rows = output channels, 16-byte loads straight from L2,
// a K chunk ahead) and the LDS-resident activations as B (columns = pixels): a lane ends up with 4 consecutive channels of
// one pixel -- 8-byte LDS stores, and a 16 x 16 tile is exactly one Mlp.norm1 group.
// What a sample's workgroups owe each other per Block -- five all-gathers:
//   E0  per-channel (sum, sum^2) of the block input          -> Block.norm1 statistics, xbar
//   E1  the own keys (sr > 1: raw sr output + partial sums)   -> attn.norm statistics, K of all keys
//   E2  per-group sums of x1                                  -> Block.norm2
//   E3  per-group sums of h1 (+ h1 itself, in global memory)  -> Mlp.norm1, the stencil's neighbour rows
//   E4  per-group sums of h2                                  -> Mlp.norm2
// travel as 8-byte {value, tag} granules: ONE agent-scope (sc1) store publishes a value, agent-scope loads poll it; the tag
// is an epoch counter that never repeats, so nothing is zeroed between exchanges, blocks or launches.  Partials are
// converted to fixed point and added as integers by every reader: the totals are identical in all workgroups and
// independent of arrival order.  No fences, no atomics, no reliance on dispatch order or workgroup -> XCD placement
// (blockIdx = g * sets + set puts a sample's workgroups on one XCD in practice: speed only).
#include <math.h>
#include "common.h"

namespace {

constexpr int NT = 512, NW = 8;       // 8 waves: 256 registers per lane (the 16-wave build spilled ~440 of its 128)
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int EPOCH_WORDS = 256;                 // one epoch word per set at the head of the workspace
constexpr unsigned long long SPIN_LIMIT = 300000000ull;   // 3 s of the 100 MHz wall clock

__device__ __forceinline__ void pub(gu64* p, unsigned tag, unsigned v) {
  __hip_atomic_store(p, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pubf(gu64* p, unsigned tag, float v) { pub(p, tag, __float_as_uint(v)); }
// An opaque copy of a thread index: index arithmetic derived from it stays inside the phase that uses it.  (Derived from the
// plain threadIdx every phase's per-thread offsets are loop invariants of the block loop: the compiler hoists all of them
// to the top of the kernel -- ~500 spilled registers in the first build.)
// Sum over the 64 lanes on the VALU (DPP row shifts + row broadcasts: a fixed tree, so reproducible), total in every lane.  The
// butterfly of common.h (__shfl_xor = ds_bpermute through the LDS crossbar, six dependent round trips) cost 2.8 us per Block in
// fc1's epilogue alone (four sums per 32 x 32 tile).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_src(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += dpp_src<0x111, 0xf>(v);      // row_shr:1
  v += dpp_src<0x112, 0xf>(v);      // row_shr:2
  v += dpp_src<0x114, 0xf>(v);      // row_shr:4
  v += dpp_src<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row of 16 holds the row's sum
  v += dpp_src<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v += dpp_src<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int opq(int v) { asm volatile("" : "+v"(v)); return v; }

// Poll n <= NMAX granules (addr(k)) until every tag matches; values to val[].  Unconditional loads (clamped index): a
// branch around a load makes the compiler wait for each one separately.
template <int NMAX, class Addr>
__device__ __forceinline__ void gather(Addr addr, int n, unsigned tag, unsigned (&val)[NMAX], volatile int* dead, int32_t* status) {
  unsigned long long t0 = 0;
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
      const unsigned long long x = __hip_atomic_load(addr(k < n ? k : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      val[k] = (unsigned)x;
      ok &= (k >= n) | ((unsigned)(x >> 32) == tag);
    }
    if (ok) break;
    if (*dead) break;
    if ((++spins & 31u) == 0) {
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > SPIN_LIMIT) { *dead = 1; atomicExch(status, 1); break; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int C_, int HID_, int HEADS_, int SR_, int WSMAX_, int RPW_>
struct Cfg {
  static constexpr int C = C_, HID = HID_, HEADS = HEADS_, SR = SR_, WSMAX = WSMAX_;
  static constexpr int RPW = RPW_;                       // image rows per workgroup: 2, or 1 when that still fits the chip (B x H <= 256)
  static constexpr int D = C / HEADS, DK = (D + 31) / 32;
  static constexpr int NPXMAX = RPW * WSMAX, RTMAX = (NPXMAX + 31) / 32 * 2, MP = RTMAX * 16;
  static constexpr int XNROWS = (SR > 1 && RPW == 1) ? (2 * WSMAX + 15) / 16 * 16 : MP;   // RPW 1, sr 2: + the partner's row (sr convolution)
  static constexpr int MMAX = 104, KTMAX = 7, MKP = KTMAX * 16;
  static constexpr int GMAX = 16 / RPW;
  static constexpr int LDA = C + 8, LDH = HID + 8, HC = HID / 2, HLD = HC + 8;
  static constexpr int NG = C / 16, NGH = HID / 16, CG = C / 8, HG = HID / 8;
  static constexpr int KC = C / 32;                      // 16-wide k-steps per weight chunk: a chunk = C / 2 input channels
  static constexpr int KCE = KC * 16;                    // elements per chunk
  static constexpr int RT32 = MP / 32;                   // 32-pixel tiles
  // packed per-block vector (fp32, crd_enc_block_desc.vec)
  static constexpr int V_N1G = 0, V_N1B = C, V_BQ = 2 * C, V_BSR = 3 * C, V_NKG = 4 * C, V_NKB = 5 * C, V_BK = 6 * C, V_BP = 7 * C,
                       V_N2G = 8 * C, V_N2B = 9 * C, V_B2 = 10 * C, V_B1 = 11 * C, V_M1G = 11 * C + HID, V_M1B = 11 * C + 2 * HID,
                       V_BDW = 11 * C + 3 * HID, V_M2G = 11 * C + 4 * HID, V_M2B = 11 * C + 5 * HID, V_TOTAL = 11 * C + 6 * HID;
  // exchange areas (granules per workgroup)
  static constexpr int E0N = 2 * C;
  static constexpr int E1KEYS = (SR > 1 ? WSMAX / SR : NPXMAX) * (C / 2);
  static constexpr int EXN = (SR > 1 && RPW == 1) ? WSMAX * (C / 2) : 0;      // the odd row's xn, handed to its even partner
  static constexpr int E1N = E1KEYS + (SR > 1 ? 2 * NG : 0);
  static constexpr int E2N = 2 * NG, E3N = 2 * NGH, E4N = 2 * NG;
  static constexpr int XE0 = 0, XE1 = XE0 + GMAX * E0N, XE2 = XE1 + GMAX * E1N, XE3 = XE2 + GMAX * E2N, XE4 = XE3 + GMAX * E3N;
  static constexpr int XEX = XE4 + GMAX * E4N;
  static constexpr int AREA = XEX + GMAX * EXN;          // granules per set
  // LDS (bytes)
  static constexpr int OFF_X = 0, SZ_X = NPXMAX * C * 4;
  static constexpr int OFF_XN = OFF_X + SZ_X, SZ_XN = XNROWS * LDA * 2;
  static constexpr int OFF_U = OFF_XN + SZ_XN;
  static constexpr int OFF_Q = OFF_U, SZ_Q = MP * LDA * 2;
  static constexpr int OFF_K = OFF_Q + SZ_Q, SZ_K = MKP * LDA * 2;
  static constexpr int OFF_KRN = OFF_K + SZ_K, SZ_KRN = SR > 1 ? MKP * LDA * 2 : 0;
  static constexpr int SZ_H = NPXMAX * LDH * 2, SZ_HOVER = MP * LDH * 2;           // fc2 reads rows up to MP (garbage, discarded)
  static constexpr int SZ_HALO = WSMAX * HLD * 2;
  static constexpr int SZ_ATT = SZ_Q + SZ_K + SZ_KRN;
  static constexpr int SZ_MLP = SZ_H + SZ_HALO > SZ_HOVER ? SZ_H + SZ_HALO : SZ_HOVER;
  static constexpr int SZ_U = SZ_ATT > SZ_MLP ? SZ_ATT : SZ_MLP;
  static constexpr int OFF_H = OFF_U, OFF_HALOB = OFF_U + SZ_H, OFF_HALOA = OFF_XN;
  static constexpr int OFF_TAB = OFF_U + SZ_U, SZ_TAB = C * 8;
  static constexpr int OFF_TABH = OFF_TAB + SZ_TAB, SZ_TABH = HID * 8;             // also the fixed-point channel sums [C][2]
  static constexpr int OFF_FXG = OFF_TABH + SZ_TABH, SZ_FXG = 2 * NGH * 8;
  static constexpr int OFF_GRP = OFF_FXG + SZ_FXG, SZ_GRP = NGH * 8;
  static constexpr int OFF_REDH = OFF_GRP + SZ_GRP, SZ_REDH = NGH * 8 * 2;
  static constexpr int OFF_UV = OFF_REDH + SZ_REDH, SZ_UV = C * 4;
  static constexpr int OFF_XBAR = OFF_UV + SZ_UV, SZ_XBAR = C * 4;
  static constexpr int OFF_S = OFF_XBAR + SZ_XBAR, SZ_S = MP * 4;
  static constexpr int OFF_SMAX = OFF_S + SZ_S, SZ_SMAX = HEADS * MP * 4;
  static constexpr int OFF_REDW = OFF_SMAX + SZ_SMAX, SZ_REDW = 2 * NW * 8;
  static constexpr int OFF_DEAD = OFF_REDW + SZ_REDW;
  static constexpr int OFF_DUMMY = OFF_DEAD + 16;            // 1 KB landing zone of the L2 warm-up requests
  static constexpr int TOTAL = OFF_DUMMY + 1024;
  static_assert((SR == 1 || SR == 2) && (RPW == 1 || RPW == 2), "one or two image rows per workgroup: sr 1 or 2");
  static_assert(SZ_HALO <= SZ_XN, "halo row A lives in the sXN region");
  static_assert(C * 2 * 8 <= SZ_TABH, "fixed-point channel sums alias sTabH");
  static_assert(TOTAL <= 160 * 1024, "LDS");
  static_assert(OFF_U % 16 == 0 && OFF_K % 16 == 0 && OFF_KRN % 16 == 0 && OFF_HALOB % 16 == 0 && OFF_TAB % 16 == 0, "alignment");
  static_assert(NG <= 2 * NW && 2 * C <= NT && 2 * NGH <= NT && MP % 32 == 0 && (C / 2) % 16 == 0 && HC / 8 <= NT / 8, "thread mappings");
};

struct EncK {
  const float* x; const crd_enc_block_desc* blocks; int nblocks, B, H, W;
  bf16_t* xb_out; unsigned long long* ws; int32_t* status; int nsets; float scale;
};

// One 32 x 32 output tile per unit: acc[32 channels][32 pixels] = bias + sum_k Wt[ct * 32 + .][k] * act[rt][k] on
// v_mfma_f32_32x32x16_bf16.  Weights (the A operand: rows = output channels) come straight from global memory, row-major [N][ldw],
// one chunk of KC k-steps requested a chunk ahead; the activations (B operand: columns = pixels) from LDS through
// bptr(rt, chunk, k, lane) -> pointer to the lane's 8 consecutive k of pixel rt * 32 + (lane & 31).  With row strides of 4 * odd
// dwords (C + 8, hid + 8 elements) those 16-byte reads are bank-conflict free.  The first version used 16 x 16 x 32 tiles:
// one ds_read_b128 per 16-cycle MFMA with 2-way conflicts made every GEMM LDS-bound (fc1: 12.6 us per Block for 1.3 us of
// MFMA time).  A lane ends with one pixel and 16 channels: quads q = 0..3 at channel ct * 32 + 8 q + 4 (lane >> 5) + {0..3}.
// Units (ct, rt) are dealt to the 16 waves.
template <int KC, int NCH, int ROT = 0, class BPtr, class Epi>
__device__ __forceinline__ void wg_gemm(const bf16_t* __restrict__ Wf, int kstot, int ntiles, int RT, const float* __restrict__ bias,
                                        int wave, int lane, BPtr bptr, Epi epi) {
  const int wv = (wave + ROT) % NW;          // ROT: which waves take the first units (balances back-to-back GEMMs with few units)
  // A wave's work is the flat sequence of (unit, chunk) items of its units; the weight chunk of item i + NB - 1 is requested
  // before item i is computed, ACROSS unit boundaries: NB - 1 chunks (KC KB each) per wave are always in flight.  NB = 2: deeper
  // rings were SLOWER (stage 3 forward at B = 8: NB 1 / 2 / 3 / 4 = 1340 / 1297 / 1430 / 1397 us) -- the loop body is unrolled NB
  // times with the epilogue inlined in each copy, and this kernel is bound by instruction issue, not by the weights' latency.  What bounds
  // these GEMMs is the rate at which one CU pulls weights out of L2 (every workgroup streams all of a block's weights:
  // 768 KB per Block at stage 3) -- with a single chunk in flight and a wait per chunk the first version reached 16 GB/s.
#ifndef CRD_ENC_NB
#define CRD_ENC_NB 2
#endif
  constexpr int NB = CRD_ENC_NB;
  const int l = opq(lane);
  const int units = ntiles * RT;
  const int nu = units > wv ? (units - wv + NW - 1) / NW : 0;      // units of this wave: wv, wv + NW, ...
  const int n = nu * NCH;
  // weights in FRAGMENT order (crd_pack_frag32): the 64 lanes' 16-byte operands of (32-row tile ct, k-step) are one contiguous
  // kilobyte.  Read from the row-major form a wave instruction touched 32 rows x 32 bytes: with eight waves doing that the 32 KB
  // L1 thrashed and every 128-byte line came from L2 four times (fc1 of stage 3: 205 KB in 11.6 us = 18 GB/s per CU).
  const bf16_t* wlane = Wf + l * 8;
  bf16x8 a[NB][KC];
  auto issue = [&](int i, bf16x8 (&dst)[KC]) {
    const int ii = i < n ? i : 0;                          // (clamped: unconditional loads; a tail request re-reads item 0)
    const int u = wv + (ii / NCH) * NW, ch = ii % NCH;
#if defined(CRD_ENC_ABLATE) && (CRD_ENC_ABLATE & 1)      // developer build: every weight request hits the same kilobytes (L1): what is left is not weight streaming
    const bf16_t* src = wlane + (u * 0 + ch * 0) * 512;
#else
    const bf16_t* src = wlane + ((long long)(u / RT) * kstot + ch * KC) * 512;
#endif
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 512);
  };
#pragma unroll
  for (int j = 0; j < NB - 1; ++j) issue(j, a[j]);
  f32x16 acc, acc1;         // two accumulation chains (even / odd k-steps): a dependent MFMA chain issues at half rate
  f32x4 bv[4];
  for (int i0 = 0; i0 < n; i0 += NB) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = i0 + j;
      issue(i + NB - 1, a[(j + NB - 1) % NB]);
      if (i < n) {                                         // wave-uniform
        const int u = wv + (i / NCH) * NW, ch = i % NCH;
        const int ct = u / RT, rt = u - ct * RT;
        if (ch == 0) {
          // the bias is REQUESTED here and added in the epilogue: as the accumulators' initial value it had to be waited for at
          // once, and memory returns in order -- every weight chunk in flight was drained at the head of every unit
          const float* bp = bias + ct * 32 + (l >> 5) * 4;
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(bp + 8 * q);
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
        }
        const bf16_t* bb = bptr(rt, ch, 0, l);
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(bb + ks * 16);
#ifdef CRD_ENC_ONE_ACC
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][ks], bf, acc, 0, 0, 0);
#else
          if (ks & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][ks], bf, acc1, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][ks], bf, acc, 0, 0, 0);
#endif
        }
        if (ch == NCH - 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + acc1[r]) + bv[r >> 2][r & 3];
          epi(ct, rt, acc, l);
        }
      }
    }
  }
}

__device__ __forceinline__ uint2 packq(const f32x16& v, int q) { return make_uint2(pack_bf2(v[4 * q], v[4 * q + 1]), pack_bf2(v[4 * q + 2], v[4 * q + 3])); }
__device__ __forceinline__ long long fx_stat(unsigned bits) { return to_fx(__uint_as_float(bits), STAT_ONE); }

// The 3 x 3 depthwise stencil of one channel granule (8 channels) for the PJ pixels of this lane (pixel phase pc, pc + 8, ...):
// nine bf16 taps from global memory (requested together), the normalised inputs from LDS (own rows in sH, neighbour rows in
// the two halo buffers), results written back IN PLACE once every lane of the wave has read its inputs (the 8 lanes of a
// granule sit in one wave).  Returns the lane's (sum, sum of squares) of the rounded outputs.
// A function of its own, NOT inlined: inside the stage kernel the register allocator spilled ~300 registers around this
// loop nest whatever its shape (taps or pixels outermost, batches of 2 / 4 / 7 pixels, scheduling barriers); alone it needs ~110.
typedef __attribute__((address_space(3))) bf16_t lds_bf16;
template <int PJ, int LDH, int HLD, int RPW>
__device__ __attribute__((noinline)) float2 dw_stencil(lds_bf16* sHc, const lds_bf16* hA, const lds_bf16* hB, const bf16_t* w9c, int hid,
                                                       const float* bias, int W, int NPX, int pc, bf16_t* h2c) {
  float wf[9][8];          // unpacked once: 72 registers (this function has them), 8 operations per (pixel, tap) less
  {
    u32x4 wt[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) wt[tap] = *reinterpret_cast<const u32x4*>(w9c + tap * hid);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int q = 0; q < 4; ++q) { wf[tap][2 * q] = bf_lo(wt[tap][q]); wf[tap][2 * q + 1] = bf_hi(wt[tap][q]); }
  }
  const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias), b1 = *reinterpret_cast<const f32x4*>(bias + 4);
  // pixels outermost; the nine 16-byte LDS reads of pixel j + 1 are requested before pixel j's arithmetic (two sets of nine in
  // registers), the products as packed FMAs (v_pk_fma_f32: two channels per instruction).  The asm statements pin the order: the
  // compiler otherwise hoists all 63 reads to the top and spills them, or sinks all the arithmetic below the last read.
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  u32x4 outp[PJ];
  const unsigned aH = (unsigned)(uintptr_t)sHc, aA = (unsigned)(uintptr_t)hA, aB = (unsigned)(uintptr_t)hB;
  auto read9 = [&](int j, u32x4 (&u)[9]) {
    const int p = pc + 8 * j, pp = p < NPX ? p : 0;
    const int ly = (RPW > 1 && pp >= W) ? 1 : 0, x = pp - ly * W;
    // per stencil row: base address and pixel stride by SELECTS between values that already exist (the opaque asm keeps the
    // compiler from sinking their computation into an if / else: written as one conditional expression per read this compiled
    // to an exec-mask branch diamond in front of every one of the 63 reads)
    unsigned radr[3], rstr[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = ly + ky - 1;
      unsigned a3 = aH + (unsigned)(yy * W * LDH * 2), sA = (unsigned)(HLD * 2), sH_ = (unsigned)(LDH * 2);
      asm volatile("" : "+v"(a3), "+v"(sA), "+v"(sH_));
      const bool up = yy < 0, dn = yy > RPW - 1;
      radr[ky] = up ? aA : (dn ? aB : a3);
      rstr[ky] = (up || dn) ? sA : sH_;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3 - 1;
      const int xx = x + kx;
      const bool xok = xx >= 0 && xx < W;
      const unsigned off = radr[ky] + (unsigned)(xok ? xx : x) * rstr[ky];
      u[tap] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>((uintptr_t)off);
      if (!xok) u[tap] = u32x4{0u, 0u, 0u, 0u};
    }
  };
  u32x4 ua[9], ub[9];
  read9(0, ua);
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    u32x4 (&cur)[9] = (j & 1) ? ub : ua;
    u32x4 (&nxt)[9] = (j & 1) ? ua : ub;
    if (j + 1 < PJ) read9(j + 1, nxt);
    f32x2 o[4] = {{b0[0], b0[1]}, {b0[2], b0[3]}, {b1[0], b1[1]}, {b1[2], b1[3]}};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2 uv = {bf_lo(cur[tap][q]), bf_hi(cur[tap][q])}, wv2 = {wf[tap][2 * q], wf[tap][2 * q + 1]};
        o[q] = __builtin_elementwise_fma(uv, wv2, o[q]);
      }
    outp[j][0] = pack_bf2(o[0][0], o[0][1]); outp[j][1] = pack_bf2(o[1][0], o[1][1]); outp[j][2] = pack_bf2(o[2][0], o[2][1]); outp[j][3] = pack_bf2(o[3][0], o[3][1]);
    asm volatile("" : "+v"(outp[j][0]), "+v"(outp[j][1]), "+v"(outp[j][2]), "+v"(outp[j][3]) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_wave_barrier();
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int j = 0; j < PJ; ++j) {
    const int p = pc + 8 * j;
    if (p < NPX) {
      const u32x4 u = outp[j];
      *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(sHc + p * LDH) = u;
      if (h2c) *reinterpret_cast<u32x4*>(h2c + (long long)p * hid) = u;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float v0 = bf_lo(u[q]), v1 = bf_hi(u[q]); s += v0 + v1; ss += v0 * v0 + v1 * v1; }
    }
  }
  return make_float2(s, ss);
}

#ifdef CRD_ENC_PROF
// per-phase wall-clock ticks (100 MHz) of workgroup 0, summed over the blocks of a launch (tools/prof_enc_stage.py --phases)
__device__ unsigned long long g_enc_prof[32];
__device__ unsigned g_enc_place[512];           // per workgroup of the last launch: XCC id << 16 | HW_ID bits (CU, SE)
#define ENC_STAMP(i) do { if (blockIdx.x == 0 && tid == 0) { const unsigned long long now_ = wall_clock64(); g_enc_prof[i] += now_ - prof_last; prof_last = now_; } } while (0)
#else
#define ENC_STAMP(i) do {} while (0)
#endif

template <class CF>
__global__ __launch_bounds__(NT) void k_enc_stage(EncK a) {
  constexpr int C = CF::C, HID = CF::HID, HEADS = CF::HEADS, SR = CF::SR, D = CF::D, DK = CF::DK;
  constexpr int LDA = CF::LDA, LDH = CF::LDH, HLD = CF::HLD, HC = CF::HC;
  constexpr int NG = CF::NG, NGH = CF::NGH, CG = CF::CG, HG = CF::HG, KC = CF::KC, RTMAX = CF::RTMAX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sX = reinterpret_cast<float*>(smem + CF::OFF_X);
  bf16_t* sXN = reinterpret_cast<bf16_t*>(smem + CF::OFF_XN);
  bf16_t* sQ = reinterpret_cast<bf16_t*>(smem + CF::OFF_Q);
  bf16_t* sK = reinterpret_cast<bf16_t*>(smem + CF::OFF_K);
  bf16_t* sKRN = reinterpret_cast<bf16_t*>(smem + CF::OFF_KRN);
  bf16_t* sH = reinterpret_cast<bf16_t*>(smem + CF::OFF_H);
  bf16_t* sHaloA = reinterpret_cast<bf16_t*>(smem + CF::OFF_HALOA);
  bf16_t* sHaloB = reinterpret_cast<bf16_t*>(smem + CF::OFF_HALOB);
  float2* sTab = reinterpret_cast<float2*>(smem + CF::OFF_TAB);
  float2* sTabH = reinterpret_cast<float2*>(smem + CF::OFF_TABH);
  long long* sFx = reinterpret_cast<long long*>(smem + CF::OFF_TABH);       // [C][2] fixed-point channel sums (block entry only)
  long long* sFxG = reinterpret_cast<long long*>(smem + CF::OFF_FXG);       // [groups][2]
  float2* sGrp = reinterpret_cast<float2*>(smem + CF::OFF_GRP);             // (mean, rstd) per group
  float2* sRedH = reinterpret_cast<float2*>(smem + CF::OFF_REDH);           // (s, ss) of h1 per 16-channel group
  float* sU = reinterpret_cast<float*>(smem + CF::OFF_UV);
  float* sXbar = reinterpret_cast<float*>(smem + CF::OFF_XBAR);
  float* sS = reinterpret_cast<float*>(smem + CF::OFF_S);
  float* sSmax = reinterpret_cast<float*>(smem + CF::OFF_SMAX);
  float2* sRedW = reinterpret_cast<float2*>(smem + CF::OFF_REDW);           // [2 rounds][16 waves]
  volatile int* sDead = reinterpret_cast<volatile int*>(smem + CF::OFF_DEAD);

  const int tid = threadIdx.x;
#ifdef CRD_ENC_PROF
  unsigned long long prof_last = wall_clock64();
#endif
#ifdef CRD_ENC_PROF
  if (tid == 0 && blockIdx.x < 512) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_enc_place[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
  }
#endif
  int t = tid, l = tid & 63;                 // re-derived from an opaque copy at the head of every phase (opq above)
#define NEWPHASE() do { t = opq(tid); l = t & 63; } while (0)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int RPW = CF::RPW;
  const int W = a.W, H = a.H, G = H / RPW, NPX = RPW * W, N = H * W;
  const int RT = (NPX + 15) >> 4;
  const int set = blockIdx.x % a.nsets, g = blockIdx.x / a.nsets;
  // keys: with sr > 1 a key row needs SR image rows -- the workgroup holding the first of them produces it ("publisher"; with one
  // image row per workgroup its partner hands over its xn row first); without sr every workgroup's pixels are its keys
  const int KPW = SR > 1 ? W / SR : NPX;                  // keys per publishing workgroup
  const bool kpub = SR > 1 ? (g * RPW) % SR == 0 : true;
  const int krow0 = SR > 1 ? (g * RPW / SR) * KPW : g * NPX;        // first key of this workgroup (publishers)
  const int M = SR > 1 ? (H / SR) * KPW : H * W, KT = (M + 15) >> 4;
  const int E1S = KPW * (C / 2) + (SR > 1 ? 2 * NG : 0);  // granules per workgroup slot in E1
  auto kslot = [&](int m, int& j) { const int jy = m / KPW; j = m - jy * KPW; return SR > 1 ? jy * SR / RPW : jy; };   // key -> (slot, index)
  gu64* const xa = (gu64*)a.ws + EPOCH_WORDS + (long long)set * CF::AREA;
  unsigned seq = (unsigned)__hip_atomic_load((gu64*)a.ws + set, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) *sDead = 0;
  __syncthreads();

  // per-channel (sum, sum of squares) of the own pixels of sX -> E0 (thread = (channel, pixel phase): 2 lanes per channel)
  auto publish_chan = [&](unsigned tag) {
    NEWPHASE();
    if (t < 2 * C) {
      const int c = t >> 1, qq = t & 1;
      float s = 0.f, ss = 0.f;
      for (int p = qq; p < NPX; p += 2) { const float v = sX[p * C + c]; s += v; ss += v * v; }
      s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
      pubf(xa + CF::XE0 + g * CF::E0N + c * 2 + qq, tag, qq ? ss : s);
    }
  };

  for (int b = set; b < a.B; b += a.nsets) {
    // ---- the own two image rows of the stage input
    NEWPHASE();
    {
      const float4* src = reinterpret_cast<const float4*>(a.x + ((long long)b * N + g * NPX) * C);
      for (int i = t; i < NPX * C / 4; i += NT) reinterpret_cast<float4*>(sX)[i] = src[i];
    }
    __syncthreads();
    ++seq;
    publish_chan(seq);
    unsigned tagE0 = seq;

    for (int blk = 0; blk < a.nblocks; ++blk) {
      const crd_enc_block_desc* d = a.blocks + blk;
      ENC_STAMP(0);
      const float dps = d->dp ? d->dp[b] : 1.f;
      const float* __restrict__ pv = d->vec;
      // L2 warm-up: the NEXT block's weights and vectors (pf_ptr / pf_bytes) are requested into this XCD's L2 through LDS-DMA into
      // a scratch area (no register destination, nothing waits for them); the sample's G workgroups share the work.  Without it every weight chunk of every GEMM is a first touch from the fabric (~1-2 us).
      if (blk + 1 < a.nblocks) {
        const crd_enc_block_desc* dn = d + 1;
        // (all sixteen descriptor words first: read one range at a time, every iteration waited for its own cold scalar load --
        // this loop took 4-5 us per Block whatever it requested)
        const void* pptr[8];
        int pbytes[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { pptr[r] = dn->pf_ptr[r]; pbytes[r] = dn->pf_bytes[r]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const void* ptr = pptr[r];
          const int bytes = pbytes[r];
          if (ptr == nullptr) continue;
          const crd_rsrc_t rs = make_rsrc(ptr, (unsigned)bytes);
          // ONE dword per half line (a wave request touches 64 x 64 bytes = 4 KB): the data lands in L2, only 4 bytes travel on to
          // the CU.  (Whole lines -- 16 bytes per lane -- made this loop 4.3 us per Block: the CU's own fill rate, ~25 GB/s.)
#ifndef CRD_ENC_WARM
#define CRD_ENC_WARM 64           // bytes between the touched dwords: every half line (fills are 64 bytes).  Stage 3 forward at B = 8, one row per
                                  // workgroup, eval / train: no warm-up 1229 / 1356 us, 128: 1238 / 1337, 64: 1207 / 1308, 16: 1199 / 1298 -- the GEMM
                                  // phases are NOT waiting for first-touch weights (fc1 6.3 us without, 5.0-5.3 with)
#endif
          constexpr int WSTEP = CRD_ENC_WARM > 0 ? CRD_ENC_WARM : 128, WSPAN = 64 * WSTEP;
          if (CRD_ENC_WARM > 0)
          for (int c = g * NW + wv; c * WSPAN < bytes; c += G * NW)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                         :: "s"((unsigned)CF::OFF_DUMMY), "v"((unsigned)(c * WSPAN + (tid & 63) * WSTEP)), "s"(rs) : "memory", "m0");
        }
      }
      const long long rowbase = (long long)b * N + g * NPX;         // first own pixel in [B][N][.] tensors

      ENC_STAMP(17);
      // ================= E0: Block.norm1 statistics, xbar =================
      NEWPHASE();
      if (t < 2 * C) {
        unsigned v[CF::GMAX];
        gather<CF::GMAX>([&](int k) { return xa + CF::XE0 + k * CF::E0N + t; }, G, tagE0, v, sDead, a.status);
        long long tot = 0;
#pragma unroll
        for (int k = 0; k < CF::GMAX; ++k) tot += k < G ? fx_stat(v[k]) : 0ll;
        sFx[t] = tot;
        if (g == 0 && d->ch1) d->ch1[(long long)b * C * 2 + t] = tot;
      }
      __syncthreads();
      ENC_STAMP(18);
      // every thread of a channel adds its group's 16 channel sums itself (broadcast LDS reads) and takes the moments: one barrier
      // instead of three (group sums -> moments -> coefficients)
      if (t < C) {
        long long gs = 0, gss = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { gs += sFx[((t & ~15) + j) * 2]; gss += sFx[((t & ~15) + j) * 2 + 1]; }
        if (g == 0 && d->st1 && (t & 15) == 0) { d->st1[((long long)b * NG + (t >> 4)) * 2] = gs; d->st1[((long long)b * NG + (t >> 4)) * 2 + 1] = gss; }
        float mean, rstd;
        gn_moments(gs, gss, (float)N * 16.f, mean, rstd);
        const float gam = pv[CF::V_N1G + t], bet = pv[CF::V_N1B + t];
        const float ga = gam * rstd;
        const float mc = (float)sFx[2 * t] * (1.f / STAT_ONE) / (float)N;
        const bf16_t qb = f2bf(gam * (mc - mean) * rstd + bet);
        sXbar[t] = bf2f(qb);
        if (g == 0 && d->xbar) reinterpret_cast<bf16_t*>(d->xbar)[(long long)b * C + t] = qb;
        sTab[t] = make_float2(ga, bet - mean * ga);
      }
      __syncthreads();                                         // sFx (aliasing sTabH) is dead from here on
      ENC_STAMP(1);

      // ================= xn = bf16(norm1(x)) =================
      NEWPHASE();
      auto normalise_to_xn = [&](bf16_t* gdst) {
        NEWPHASE();
        for (int i = t; i < NPX * CG; i += NT) {
          const int p = i / CG, cg = i - p * CG;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sX + p * C + cg * 8), v1 = *reinterpret_cast<const f32x4*>(sX + p * C + cg * 8 + 4);
          const float2* tb = sTab + cg * 8;
          u32x4 o;
          o[0] = pack_bf2(v0[0] * tb[0].x + tb[0].y, v0[1] * tb[1].x + tb[1].y);
          o[1] = pack_bf2(v0[2] * tb[2].x + tb[2].y, v0[3] * tb[3].x + tb[3].y);
          o[2] = pack_bf2(v1[0] * tb[4].x + tb[4].y, v1[1] * tb[5].x + tb[5].y);
          o[3] = pack_bf2(v1[2] * tb[6].x + tb[6].y, v1[3] * tb[7].x + tb[7].y);
          *reinterpret_cast<u32x4*>(sXN + p * LDA + cg * 8) = o;
          if (gdst) *reinterpret_cast<u32x4*>(gdst + (rowbase + p) * C + cg * 8) = o;
        }
      };
      const unsigned tagE1 = ++seq;
      normalise_to_xn(reinterpret_cast<bf16_t*>(d->xn));
      if constexpr (SR > 1 && RPW == 1) {
        // one image row per workgroup: the sr convolution's second row belongs to the partner -- the odd workgroup hands its xn row
        // over (XEX), the even one takes it into sXN rows [W, 2W)
        NEWPHASE();
        constexpr int NXH = (CF::WSMAX * (C / 2) + NT - 1) / NT;
        if (!kpub) {
          __syncthreads();                       // (own sXN rows complete)
          for (int i = t; i < W * (C / 2); i += NT) {
            const int p = i / (C / 2), cp = i - p * (C / 2);
            pub(xa + CF::XEX + g * CF::EXN + i, tagE1, *reinterpret_cast<const unsigned*>(sXN + p * LDA + 2 * cp));
          }
        } else {
          unsigned xv[NXH];
          const int ntot = W * (C / 2);
          const int nmine = t < ntot ? (ntot - t + NT - 1) / NT : 0;
          gather<NXH>([&](int k) { return xa + CF::XEX + (g + 1) * CF::EXN + t + k * NT; }, nmine, tagE1, xv, sDead, a.status);
#pragma unroll
          for (int k = 0; k < NXH; ++k) {
            const int e = t + k * NT;
            if (e < ntot) { const int p = e / (C / 2), cp = e - p * (C / 2); *reinterpret_cast<unsigned*>(sXN + (W + p) * LDA + 2 * cp) = xv[k]; }
          }
        }
      }
      __syncthreads();
      ENC_STAMP(2);

      // ================= q (own pixels) and the key path's first GEMM =================
      NEWPHASE();
      constexpr int RT32MAX = CF::RT32;
      const int RT32 = (NPX + 31) >> 5;
      wg_gemm<KC, 2>(reinterpret_cast<const bf16_t*>(d->wq), C / 16, C / 32, RT32, pv + CF::V_BQ, wv, l,
                     [&](int rt, int ch, int k, int l) { return sXN + (rt * 32 + (l & 31)) * LDA + ch * CF::KCE + k + (l >> 5) * 8; },
                     [&](int ct, int rt, const f32x16& acc, int l) {
                       bf16_t* dst = sQ + (rt * 32 + (l & 31)) * LDA + ct * 32 + (l >> 5) * 4;
#pragma unroll
                       for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(dst + 8 * q) = packq(acc, q);
                     });
      ENC_STAMP(25);
      if constexpr (SR > 1) {
        // kr[j][co] = bf16(sum_{tap, ci} Wsr[co][tap][ci] * xn[pixel(j, tap)][ci] + b): one key row per publisher (keys on the columns)
        if (kpub)
        wg_gemm<KC, 2 * SR * SR, NW - 3>(reinterpret_cast<const bf16_t*>(d->wsr), SR * SR * C / 16, C / 32, 1, pv + CF::V_BSR, wv, l,
            [&](int, int ch, int k, int l) {
              const int tap = ch >> 1;
              int j = l & 31;
              j = j < KPW ? j : KPW - 1;
              return sXN + ((tap / SR) * W + SR * j + (tap % SR)) * LDA + (ch & 1) * CF::KCE + k + (l >> 5) * 8;
            },
            [&](int ct, int, const f32x16& acc, int l) {
              const int j = l & 31, c0 = ct * 32 + (l >> 5) * 4;
              float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};         // the tile's two 16-channel groups
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const uint2 pk = packq(acc, q);
                if (j < KPW) {
                  gu64* dst = xa + CF::XE1 + g * E1S + j * (C / 2) + ((c0 + 8 * q) >> 1);
                  pub(dst, tagE1, pk.x);
                  pub(dst + 1, tagE1, pk.y);
                  if (d->kr) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(d->kr) + ((long long)b * M + krow0 + j) * C + c0 + 8 * q) = pk;
                  const float v0 = bf_lo(pk.x), v1 = bf_hi(pk.x), v2 = bf_lo(pk.y), v3 = bf_hi(pk.y);
                  s[q >> 1] += (v0 + v1) + (v2 + v3);
                  ss[q >> 1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                }
              }
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const float ts = wave_sum_dpp(s[h]), tss = wave_sum_dpp(ss[h]);
                if (l < 2) pubf(xa + CF::XE1 + g * E1S + KPW * (C / 2) + (ct * 2 + h) * 2 + l, tagE1, l ? tss : ts);
              }
            });
      } else {
        wg_gemm<KC, 2>(reinterpret_cast<const bf16_t*>(d->wk), C / 16, C / 32, RT32, pv + CF::V_BK, wv, l,
            [&](int rt, int ch, int k, int l) { return sXN + (rt * 32 + (l & 31)) * LDA + ch * CF::KCE + k + (l >> 5) * 8; },
            [&](int ct, int rt, const f32x16& acc, int l) {
              const int p = rt * 32 + (l & 31), c0 = ct * 32 + (l >> 5) * 4;
              if (p < NPX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const uint2 pk = packq(acc, q);
                  gu64* dst = xa + CF::XE1 + g * E1S + p * (C / 2) + ((c0 + 8 * q) >> 1);
                  pub(dst, tagE1, pk.x);
                  pub(dst + 1, tagE1, pk.y);
                }
              }
            });
      }
      ENC_STAMP(26);
      NEWPHASE();
      // ---- u = Wp * xbar in fp64 (per-sample vectors broadcast over every pixel: their rounding error is coherent, see
      // k_attn_xbar_proj), while the other workgroups' keys arrive.  Thread = (row, quarter of the columns).
      for (int r = t >> 2; r < C; r += NT / 4) {
        const int qq = t & 3;
        const bf16_t* wr = reinterpret_cast<const bf16_t*>(d->wp) + (long long)r * C + qq * (C / 4);
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < C / 32; ++i) {
          float wv8[8];
          load8t<0>(wr, i * 8, wv8);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc += (double)wv8[j] * (double)sXbar[qq * (C / 4) + i * 8 + j];
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (qq == 0) {
          sU[r] = (float)acc;
          if (g == 0 && d->u) d->u[(long long)b * C + r] = (float)acc;
        }
      }
      __syncthreads();                          // sQ complete (and sXN free)
      ENC_STAMP(3);
      NEWPHASE();
      if (d->q) {
        for (int i = t; i < NPX * CG; i += NT) {
          const int p = i / CG, cg = i - p * CG;
          *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d->q) + (rowbase + p) * C + cg * 8) = *reinterpret_cast<const u32x4*>(sQ + p * LDA + cg * 8);
        }
      }

      // ================= E1: keys of the whole sample =================
      NEWPHASE();
      constexpr int NK1 = (CF::MMAX * (C / 2) + NT - 1) / NT;
      if constexpr (SR > 1) {
        if (t < 2 * NG) {
          unsigned v[CF::GMAX];
          const int npub = H / SR;
          gather<CF::GMAX>([&](int k) { return xa + CF::XE1 + (k * SR / RPW) * E1S + KPW * (C / 2) + t; }, npub, tagE1, v, sDead, a.status);
          long long tot = 0;
#pragma unroll
          for (int k = 0; k < CF::GMAX; ++k) tot += k < npub ? fx_stat(v[k]) : 0ll;
          sFxG[t] = tot;
          if (g == 0 && d->stk) d->stk[(long long)b * NG * 2 + t] = tot;
        }
        unsigned kv[NK1];
        const int ntot = M * (C / 2);
        const int nmine = t < ntot ? (ntot - t + NT - 1) / NT : 0;
        gather<NK1>([&](int k) {
          const int e = t + k * NT, m = e / (C / 2), cp = e - m * (C / 2);
          int j;
          const int sw = kslot(m, j);
          return xa + CF::XE1 + sw * E1S + j * (C / 2) + cp;
        }, nmine, tagE1, kv, sDead, a.status);
        __syncthreads();
        if (t < C) {
          float mean, rstd;
          gn_moments(sFxG[2 * (t >> 4)], sFxG[2 * (t >> 4) + 1], (float)M * 16.f, mean, rstd);
          const float ga = pv[CF::V_NKG + t] * rstd;
          sTab[t] = make_float2(ga, pv[CF::V_NKB + t] - mean * ga);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NK1; ++k) {
          const int e = t + k * NT;
          if (e < ntot) {
            const int m = e / (C / 2), cp = e - m * (C / 2);
            const float2 t0 = sTab[2 * cp], t1 = sTab[2 * cp + 1];
            *reinterpret_cast<unsigned*>(sKRN + m * LDA + 2 * cp) = pack_bf2(bf_lo(kv[k]) * t0.x + t0.y, bf_hi(kv[k]) * t1.x + t1.y);
          }
        }
        __syncthreads();
        ENC_STAMP(27);
        if (d->krn && kpub) {
          for (int i = t; i < KPW * CG; i += NT) {
            const int j = i / CG, cg = i - j * CG;
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d->krn) + ((long long)b * M + krow0 + j) * C + cg * 8) =
                *reinterpret_cast<const u32x4*>(sKRN + (krow0 + j) * LDA + cg * 8);
          }
        }
        // k = bf16(Wk * krn + bk) for ALL keys (every workgroup: 5 MFLOP, cheaper than another exchange)
        wg_gemm<KC, 2>(reinterpret_cast<const bf16_t*>(d->wk), C / 16, C / 32, (M + 31) >> 5, pv + CF::V_BK, wv, l,
            [&](int rt, int ch, int k, int l) { return sKRN + (rt * 32 + (l & 31)) * LDA + ch * CF::KCE + k + (l >> 5) * 8; },
            [&](int ct, int rt, const f32x16& acc, int l) {
              const int m = rt * 32 + (l & 31);            // (rows past the 112 allocated are read as garbage and not written)
              if (m < CF::MKP) {
                bf16_t* dst = sK + m * LDA + ct * 32 + (l >> 5) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(dst + 8 * q) = packq(acc, q);
              }
            });
        __syncthreads();
      } else {
        unsigned kv[NK1];
        const int ntot = M * (C / 2);
        const int nmine = t < ntot ? (ntot - t + NT - 1) / NT : 0;
        gather<NK1>([&](int k) {
          const int e = t + k * NT, m = e / (C / 2), cp = e - m * (C / 2);
          int j;
          const int sw = kslot(m, j);
          return xa + CF::XE1 + sw * E1S + j * (C / 2) + cp;
        }, nmine, tagE1, kv, sDead, a.status);
#pragma unroll
        for (int k = 0; k < NK1; ++k) {
          const int e = t + k * NT;
          if (e < ntot) { const int m = e / (C / 2), cp = e - m * (C / 2); *reinterpret_cast<unsigned*>(sK + m * LDA + 2 * cp) = kv[k]; }
        }
        __syncthreads();
      }
      if (d->k && kpub) {
        for (int i = t; i < KPW * CG; i += NT) {
          const int j = i / CG, cg = i - j * CG;
          *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d->k) + ((long long)b * M + krow0 + j) * C + cg * 8) =
              *reinterpret_cast<const u32x4*>(sK + (krow0 + j) * LDA + cg * 8);
        }
      }

      ENC_STAMP(4);
      // ================= scores: s_h[n] = max_m bf16(bf16(q_n . k_m) * scale), arg-max; S = sum_h s_h =================
      NEWPHASE();
      {
        for (int un = wv; un < HEADS * RT; un += NW) {
          const int h = un % HEADS, qt = un / HEADS;
          const float scl = a.scale;
          const int qn = qt * 16 + (l & 15);
          bf16x8 qf[DK];
#pragma unroll
          for (int ks = 0; ks < DK; ++ks) {
            const int kk = ks * 32 + (l >> 4) * 8;
            u32x4 u = *reinterpret_cast<const u32x4*>(sQ + qn * LDA + h * D + (kk < D ? kk : 0));
            if (kk >= D) u = u32x4{0u, 0u, 0u, 0u};
            qf[ks] = *reinterpret_cast<bf16x8*>(&u);
          }
          float best = -INFINITY;
          int besti = 0;
          for (int kt = 0; kt < KT; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
              const int kk = ks * 32 + (l >> 4) * 8;
              u32x4 u = *reinterpret_cast<const u32x4*>(sK + (kt * 16 + (l & 15)) * LDA + h * D + (kk < D ? kk : 0));
              if (kk >= D) u = u32x4{0u, 0u, 0u, 0u};
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&u), qf[ks], acc, 0, 0, 0);
            }
            const int m0 = kt * 16 + (l >> 4) * 4;
            const uint32_t p1a = pack_bf2(acc[0], acc[1]), p1b = pack_bf2(acc[2], acc[3]);
            const uint32_t p2a = pack_bf2(bf_lo(p1a) * scl, bf_hi(p1a) * scl), p2b = pack_bf2(bf_lo(p1b) * scl, bf_hi(p1b) * scl);
            float v[4] = {bf_lo(p2a), bf_hi(p2a), bf_lo(p2b), bf_hi(p2b)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float vi = m0 + i < M ? v[i] : -INFINITY;
              const bool tk = vi > best;
              best = tk ? vi : best;
              besti = tk ? m0 + i : besti;
            }
          }
#pragma unroll
          for (int o = 16; o < 64; o <<= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(besti, o);
            if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
          }
          if (l < 16) {
            sSmax[h * CF::MP + qn] = best;
            if (d->idx && qn < NPX) d->idx[(rowbase + qn) * HEADS + h] = (short)besti;
          }
        }
      }
      __syncthreads();
      if (t < NPX) {
        float Ssum = 0.f;
#pragma unroll
        for (int hh = 0; hh < HEADS; ++hh) Ssum += sSmax[hh * CF::MP + t];
        sS[t] = Ssum;
        if (d->ssum) d->ssum[rowbase + t] = Ssum;
      }
      __syncthreads();
      ENC_STAMP(5);

      // ================= x1 = x + dp * bf16(u * S + bp);  E2: Block.norm2 statistics =================
      NEWPHASE();
      for (int i = t; i < NPX * CG; i += NT) {
        const int p = i / CG, cg = i - p * CG;
        const float s = sS[p];
        float* xp = sX + p * C + cg * 8;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(xp), v1 = *reinterpret_cast<const f32x4*>(xp + 4);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(sU + cg * 8), u1 = *reinterpret_cast<const f32x4*>(sU + cg * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(pv + CF::V_BP + cg * 8), b1 = *reinterpret_cast<const f32x4*>(pv + CF::V_BP + cg * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] += dps * bf_round(u0[j] * s + b0[j]); v1[j] += dps * bf_round(u1[j] * s + b1[j]); }
        *reinterpret_cast<f32x4*>(xp) = v0;
        *reinterpret_cast<f32x4*>(xp + 4) = v1;
        if (d->x1) { float* gx = d->x1 + (rowbase + p) * C + cg * 8; *reinterpret_cast<f32x4*>(gx) = v0; *reinterpret_cast<f32x4*>(gx + 4) = v1; }
      }
      __syncthreads();
      ENC_STAMP(6);
      const unsigned tagE2 = ++seq;
      if (t < 2 * C) {                 // 2 lanes per channel: a half-wave = the 16 channels of one group
        const int c = t >> 1, qq = t & 1;
        float s = 0.f, ss = 0.f;
        for (int p = qq; p < NPX; p += 2) { const float v = sX[p * C + c]; s += v; ss += v * v; }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
        if ((l & 31) < 2) pubf(xa + CF::XE2 + g * CF::E2N + (wv * 2 + (l >> 5)) * 2 + (l & 1), tagE2, (l & 1) ? ss : s);
      }
      if (t < 2 * NG) {
        unsigned v[CF::GMAX];
        gather<CF::GMAX>([&](int k) { return xa + CF::XE2 + k * CF::E2N + t; }, G, tagE2, v, sDead, a.status);
        long long tot = 0;
#pragma unroll
        for (int k = 0; k < CF::GMAX; ++k) tot += k < G ? fx_stat(v[k]) : 0ll;
        sFxG[t] = tot;
        if (g == 0 && d->st2) d->st2[(long long)b * NG * 2 + t] = tot;
      }
      __syncthreads();
      if (t < C) {
        float mean, rstd;
        gn_moments(sFxG[2 * (t >> 4)], sFxG[2 * (t >> 4) + 1], (float)N * 16.f, mean, rstd);
        const float ga = pv[CF::V_N2G + t] * rstd;
        sTab[t] = make_float2(ga, pv[CF::V_N2B + t] - mean * ga);
      }
      __syncthreads();
      ENC_STAMP(7);
      normalise_to_xn(reinterpret_cast<bf16_t*>(d->xn2));
      __syncthreads();
      ENC_STAMP(8);

      // ================= fc1 -> h1 (LDS + global), Mlp.norm1 partial sums =================
      NEWPHASE();
      wg_gemm<KC, 2>(reinterpret_cast<const bf16_t*>(d->w1), C / 16, HID / 32, RT32, pv + CF::V_B1, wv, l,
          [&](int rt, int ch, int k, int l) { return sXN + (rt * 32 + (l & 31)) * LDA + ch * CF::KCE + k + (l >> 5) * 8; },
          [&](int ct, int rt, const f32x16& acc, int l) {
            const int p = rt * 32 + (l & 31);
            float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};           // the tile's two Mlp.norm1 groups
            if (p < NPX) {
              bf16_t* dst = sH + p * LDH + ct * 32 + (l >> 5) * 4;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const uint2 pk = packq(acc, q);
                *reinterpret_cast<uint2*>(dst + 8 * q) = pk;
                const float v0 = bf_lo(pk.x), v1 = bf_hi(pk.x), v2 = bf_lo(pk.y), v3 = bf_hi(pk.y);
                s[q >> 1] += (v0 + v1) + (v2 + v3);
                ss[q >> 1] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
              }
            }
#if !(defined(CRD_ENC_ABLATE) && (CRD_ENC_ABLATE & 2))   // developer build: without the per-tile statistics reduction
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float ts = wave_sum_dpp(s[h]), tss = wave_sum_dpp(ss[h]);
              if (l == 0) sRedH[(ct * 2 + h) * RT32MAX + rt] = make_float2(ts, tss);
            }
#endif
          });
      __syncthreads();
      ENC_STAMP(9);
      const unsigned tagE3 = ++seq;
      {
        // h1 of the own rows to global memory with write-through (sc1) stores: the neighbours' stencils read rows of it
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<bf16_t*>(d->h1) + (long long)b * N * HID), 0, N * HID * 2, 0x00020000);
        for (int i = t; i < NPX * HG; i += NT) {
          const int p = i / HG, hg = i - p * HG;
          const u32x4 v = *reinterpret_cast<const u32x4*>(sH + p * LDH + hg * 8);
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(((g * NPX + p) * HID + hg * 8) * 2), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      ENC_STAMP(10);
      if (t < 2 * NGH) {
        float v = 0.f;
        for (int r = 0; r < RT32; ++r) { const float2 pr = sRedH[(t >> 1) * RT32MAX + r]; v += (t & 1) ? pr.y : pr.x; }
        pubf(xa + CF::XE3 + g * CF::E3N + t, tagE3, v);
      }

      // ================= E3: Mlp.norm1 statistics =================
      NEWPHASE();
      if (t < 2 * NGH) {
        unsigned v[CF::GMAX];
        gather<CF::GMAX>([&](int k) { return xa + CF::XE3 + k * CF::E3N + t; }, G, tagE3, v, sDead, a.status);
        long long tot = 0;
#pragma unroll
        for (int k = 0; k < CF::GMAX; ++k) tot += k < G ? fx_stat(v[k]) : 0ll;
        sFxG[t] = tot;
        if (g == 0 && d->sth1) d->sth1[(long long)b * NGH * 2 + t] = tot;
      }
      __syncthreads();
      // Every workgroup of the sample has published: its h1 rows are in memory.  The stencil's neighbour rows (image rows
      // RPW g - 1 and RPW g + RPW, both rounds) are requested NOW, with agent-scope loads, and land while the statistics are finished.
      constexpr int CGR = HC / 8;                                            // channel granules per round
      constexpr int HLR = (2 * CF::WSMAX * CGR + NT - 1) / NT;               // neighbour-row granules per thread and round
      u32x4 hal[2][HLR];
      {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<bf16_t*>(d->h1) + (long long)b * N * HID), 0, N * HID * 2, 0x00020000);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int k = 0; k < HLR; ++k) {
            const int i = t + k * NT;
            const int hr = i / (W * CGR), rem = i - hr * (W * CGR), x = rem / CGR, cgl = rem - x * CGR;
            const int gy = hr ? RPW * g + RPW : RPW * g - 1;
            const bool ok = i < 2 * W * CGR && gy >= 0 && gy < H;
            hal[r][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(((gy * W + x) * HID + r * HC + cgl * 8) * 2) : 0u, 0, 16);
          }
      }
      for (int c = t; c < HID; c += NT) {
        float mean, rstd;
        gn_moments(sFxG[2 * (c >> 4)], sFxG[2 * (c >> 4) + 1], (float)N * 16.f, mean, rstd);
        const float ga = pv[CF::V_M1G + c] * rstd;
        sTabH[c] = make_float2(ga, pv[CF::V_M1B + c] - mean * ga);
      }
      __syncthreads();
      ENC_STAMP(11);

      // ================= depthwise 3x3 on norm1(h1), in place: two rounds of hid / 2 channels =================
      // thread = (channel granule of the round, pixel phase of 8); the 8 lanes of a granule sit in one wave, so the in-place
      // update (all reads of a granule, then its writes) needs no barrier.  Zero padding applies to the NORMALISED tensor.
      // The granule's nine taps (bf16, 16 bytes each) are requested together at the head of the round.
      {
        auto norm8 = [&](const u32x4& u, int c0) {
          const float2* tb = sTabH + c0;
          u32x4 o;
          o[0] = pack_bf2(bf_lo(u[0]) * tb[0].x + tb[0].y, bf_hi(u[0]) * tb[1].x + tb[1].y);
          o[1] = pack_bf2(bf_lo(u[1]) * tb[2].x + tb[2].y, bf_hi(u[1]) * tb[3].x + tb[3].y);
          o[2] = pack_bf2(bf_lo(u[2]) * tb[4].x + tb[4].y, bf_hi(u[2]) * tb[5].x + tb[5].y);
          o[3] = pack_bf2(bf_lo(u[3]) * tb[6].x + tb[6].y, bf_hi(u[3]) * tb[7].x + tb[7].y);
          return o;
        };
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          NEWPHASE();
          const int cgl = t >> 3, pc = t & 7;
          const bool act = cgl < CGR;
          const int c0 = r * HC + (act ? cgl : 0) * 8;
          // neighbour rows of this round's channels -> LDS, normalised
#pragma unroll
          for (int k = 0; k < HLR; ++k) {
            const int i = t + k * NT;
            if (i < 2 * W * CGR) {
              const int hr = i / (W * CGR), rem = i - hr * (W * CGR), x = rem / CGR, cl = rem - x * CGR;
              const int gy = hr ? RPW * g + RPW : RPW * g - 1;
              u32x4 u = norm8(hal[r][k], r * HC + cl * 8);
              if (gy < 0 || gy >= H) u = u32x4{0u, 0u, 0u, 0u};
              *reinterpret_cast<u32x4*>((hr ? sHaloB : sHaloA) + x * HLD + cl * 8) = u;
            }
          }
          if (act) {
            for (int p = pc; p < NPX; p += 8) {
              u32x4* cell = reinterpret_cast<u32x4*>(sH + p * LDH + c0);
              *cell = norm8(*cell, c0);
            }
          }
          __syncthreads();
          ENC_STAMP(19 + 3 * r);
          float s = 0.f, ss = 0.f;
          if (act) {
            constexpr int PJ = (CF::NPXMAX + 7) / 8;
            const float2 r2 = dw_stencil<PJ, LDH, HLD, CF::RPW>(
                (lds_bf16*)(sH + c0), (const lds_bf16*)(sHaloA + cgl * 8), (const lds_bf16*)(sHaloB + cgl * 8),
                reinterpret_cast<const bf16_t*>(d->w9b) + c0, HID, pv + CF::V_BDW + c0, W, NPX, pc,
                d->h2 ? reinterpret_cast<bf16_t*>(d->h2) + rowbase * HID + c0 : nullptr);
            s = r2.x; ss = r2.y;
          }
          ENC_STAMP(20 + 3 * r);
          s = wave_sum_dpp(s); ss = wave_sum_dpp(ss);   // a wave = 8 granules = 64 channels = one Mlp.norm2 group
          if (l == 0) sRedW[r * NW + wv] = make_float2(s, ss);
          __syncthreads();
          ENC_STAMP(21 + 3 * r);
        }
      }
      ENC_STAMP(12);
      // ================= E4: Mlp.norm2 statistics (groups of hid / (C / 16) = 64 channels = one wave of a round) =================
      NEWPHASE();
      const unsigned tagE4 = ++seq;
      if (t < 2 * NG) {
        constexpr int GPR = NG / 2;                      // groups per round = active waves of a round
        const int grp = t >> 1, r = grp / GPR;
        const float2 pa = sRedW[r * NW + (grp - r * GPR)];
        pubf(xa + CF::XE4 + g * CF::E4N + t, tagE4, (t & 1) ? pa.y : pa.x);
        unsigned v[CF::GMAX];
        gather<CF::GMAX>([&](int k) { return xa + CF::XE4 + k * CF::E4N + t; }, G, tagE4, v, sDead, a.status);
        long long tot = 0;
#pragma unroll
        for (int k = 0; k < CF::GMAX; ++k) tot += k < G ? fx_stat(v[k]) : 0ll;
        sFxG[t] = tot;
        // the per-launch kernels keep these sums per 16-channel slab ([hid/16][2], a group = 4 slabs); the backward only ever
        // adds a group's slabs, so the group total goes to the first slab and zeros to the others
        if (g == 0 && d->sth2) {
          crd_sum_t* o = d->sth2 + ((long long)b * NGH + grp * (NGH / NG)) * 2 + (t & 1);
          o[0] = tot;
#pragma unroll
          for (int k = 1; k < NGH / NG; ++k) o[2 * k] = 0;
        }
      }
      __syncthreads();
      for (int c = t; c < HID; c += NT) {
        const int grp = c / (16 * (NGH / NG));
        float mean, rstd;
        gn_moments(sFxG[2 * grp], sFxG[2 * grp + 1], (float)N * 16.f * (NGH / NG), mean, rstd);
        const float ga = pv[CF::V_M2G + c] * rstd;
        sTabH[c] = make_float2(ga, pv[CF::V_M2B + c] - mean * ga);
      }
      __syncthreads();
      ENC_STAMP(13);

      // ================= h3 = bf16(GELU(norm2(h2))) in place =================
      NEWPHASE();
      for (int i = t; i < NPX * HG; i += NT) {
        const int p = i / HG, hg = i - p * HG;
        u32x4* cell = reinterpret_cast<u32x4*>(sH + p * LDH + hg * 8);
        const u32x4 u = *cell;
        const float2* tb = sTabH + hg * 8;
        u32x4 o;
        o[0] = pack_bf2(gelu_exact(bf_lo(u[0]) * tb[0].x + tb[0].y), gelu_exact(bf_hi(u[0]) * tb[1].x + tb[1].y));
        o[1] = pack_bf2(gelu_exact(bf_lo(u[1]) * tb[2].x + tb[2].y), gelu_exact(bf_hi(u[1]) * tb[3].x + tb[3].y));
        o[2] = pack_bf2(gelu_exact(bf_lo(u[2]) * tb[4].x + tb[4].y), gelu_exact(bf_hi(u[2]) * tb[5].x + tb[5].y));
        o[3] = pack_bf2(gelu_exact(bf_lo(u[3]) * tb[6].x + tb[6].y), gelu_exact(bf_hi(u[3]) * tb[7].x + tb[7].y));
        *cell = o;
        if (d->h3) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(d->h3) + (rowbase + p) * HID + hg * 8) = o;
      }
      __syncthreads();
      ENC_STAMP(14);

      // ================= fc2 + residual: x2 = x1 + dp * bf16(W2 h3 + b2) =================
      NEWPHASE();
      wg_gemm<KC, 2 * HID / C>(reinterpret_cast<const bf16_t*>(d->w2), HID / 16, C / 32, RT32, pv + CF::V_B2, wv, l,
          [&](int rt, int ch, int k, int l) { return sH + (rt * 32 + (l & 31)) * LDH + ch * CF::KCE + k + (l >> 5) * 8; },
          [&](int ct, int rt, const f32x16& acc, int l) {
            const int p = rt * 32 + (l & 31);
            if (p < NPX) {
              float* xp = sX + p * C + ct * 32 + (l >> 5) * 4;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(xp + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += dps * bf_round(acc[4 * q + j]);
                *reinterpret_cast<f32x4*>(xp + 8 * q) = v;
              }
            }
          });
      __syncthreads();
      ENC_STAMP(15);
      NEWPHASE();
      const bool last = blk + 1 == a.nblocks;
      if (!last) { tagE0 = ++seq; publish_chan(tagE0); }
      if (d->x2 || (last && a.xb_out)) {
        for (int i = t; i < NPX * CG; i += NT) {
          const int p = i / CG, cg = i - p * CG;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sX + p * C + cg * 8), v1 = *reinterpret_cast<const f32x4*>(sX + p * C + cg * 8 + 4);
          if (d->x2) { float* gx = d->x2 + (rowbase + p) * C + cg * 8; *reinterpret_cast<f32x4*>(gx) = v0; *reinterpret_cast<f32x4*>(gx + 4) = v1; }
          if (last && a.xb_out) {
            u32x4 o;
            o[0] = pack_bf2(v0[0], v0[1]); o[1] = pack_bf2(v0[2], v0[3]); o[2] = pack_bf2(v1[0], v1[1]); o[3] = pack_bf2(v1[2], v1[3]);
            *reinterpret_cast<u32x4*>(a.xb_out + (rowbase + p) * C + cg * 8) = o;
          }
        }
      }
      ENC_STAMP(16);
    }
    __syncthreads();
  }
  if (g == 0 && tid == 0) __hip_atomic_store((gu64*)a.ws + set, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int RPW> using Cfg3 = Cfg<160, 640, 4, 2, 26, RPW>;
template <int RPW> using Cfg4 = Cfg<256, 1024, 8, 1, 13, RPW>;

// Compute units of the current device (hipDeviceAttributeMultiprocessorCount; 256 on a whole MI355X, fewer in a CPX / DPX partition).
// The inter-workgroup exchanges need every workgroup of a launch RESIDENT (one per CU: 124-152 KB of LDS each), so the grid is sized
// from this, never from a literal.  0 when no device can be queried (the entry points then report "unsupported").
int device_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0, n = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 0;
  }
  return cus;
}

template <class CF>
int sets_for(int B, int G, int cus = -1) {
  int cap = (cus < 0 ? device_cus() : cus) / G;      // one workgroup per CU, all resident
  if (cap >= 8) cap = cap / 8 * 8;         // a multiple of the XCD count keeps a sample on one XCD
  return B < cap ? B : cap;
}

// image rows per workgroup: one while the whole batch still fits the chip that way (a workgroup per CU), else two
int rows_per_wg(int B, int H, int want) {
  if (want == 1 || want == 2) return want;
  return B * H <= device_cus() ? 1 : 2;
}

int stage_kind(int H, int W, int C, int hid, int heads, int sr) {
  if (H < 2 || (H & 1) || W < sr || W % sr != 0) return 0;
  if (C == 160 && hid == 640 && heads == 4 && sr == 2 && W <= 26 && H <= 16 && (H / 2) * (W / 2) <= 104) return 3;
  if (C == 256 && hid == 1024 && heads == 8 && sr == 1 && W <= 13 && H <= 16 && H * W <= 104) return 4;
  return 0;
}

template <class CF>
int launch_stage(const crd_enc_stage_desc* d, hipStream_t st) {
  static int attr_rc = -1;
  if (attr_rc != 0) {
    attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_enc_stage<CF>), hipFuncAttributeMaxDynamicSharedMemorySize, CF::TOTAL);
    if (attr_rc != 0) { crd_set_error("crd_enc_stage_fwd: cannot reserve %d bytes of LDS (hip error %d)", CF::TOTAL, attr_rc); return CRD_E_LAUNCH; }
  }
  EncK k;
  k.x = d->x; k.blocks = d->blocks; k.nblocks = d->nblocks; k.B = d->B; k.H = d->H; k.W = d->W;
  k.xb_out = reinterpret_cast<bf16_t*>(d->xb_out); k.ws = reinterpret_cast<unsigned long long*>(d->sync_ws); k.status = d->status;
  const int G = d->H / CF::RPW;
  k.nsets = sets_for<CF>(d->B, G);
  k.scale = (float)pow((double)CF::D, -0.5);      // head_dim ** -0.5 (simplified_attention.py:54)
  hipLaunchKernelGGL((k_enc_stage<CF>), dim3(k.nsets * G), dim3(NT), CF::TOTAL, st, k);
  return 0;
}

}  // namespace

#ifdef CRD_ENC_PROF
extern "C" int crd_dbg_enc_place(unsigned* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_place), sizeof(g_enc_place)); }
extern "C" int crd_dbg_enc_prof(unsigned long long* out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_prof), sizeof(g_enc_prof));
  if (reset) { unsigned long long z[32] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_enc_prof), z, sizeof(z)); }
  return rc;
}
#endif

namespace {
// dst granule (ct, ks, lane) <- src[ct * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 .. + 8]   (bf16 [N][K] row-major -> the order in
// which v_mfma_f32_32x32x16_bf16 takes its A operand: one contiguous kilobyte per (32-row tile, k-step))
__global__ __launch_bounds__(256) void k_pack_frag32(const crd_frag_entry* tab) {
  const crd_frag_entry e = tab[blockIdx.y];
  const int ks_tot = e.K >> 4;
  const long long n = (long long)e.N * e.K / 8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long long r = i >> 6;
    const int ks = (int)(r % ks_tot), ct = (int)(r / ks_tot);
    const bf16_t* src = reinterpret_cast<const bf16_t*>(e.src) + (long long)(ct * 32 + (lane & 31)) * e.K + ks * 16 + (lane >> 5) * 8;
    reinterpret_cast<u32x4*>(e.dst)[i] = *reinterpret_cast<const u32x4*>(src);
  }
}

}  // namespace

extern "C" int crd_pack_frag32(const crd_frag_entry* table_dev, int32_t n, int64_t max_elems, crd_stream_t stream) {
  CRD_CHECK_ARG(table_dev && n > 0 && max_elems > 0, "crd_pack_frag32: bad argument");
  long long blocks = (max_elems / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(k_pack_frag32, dim3((unsigned)blocks, n), dim3(256), 0, as_stream(stream), table_dev);
  CRD_LAUNCH_CHECK("crd_pack_frag32");
  return CRD_OK;
}

extern "C" int crd_enc_stage_supported(int32_t B, int32_t H, int32_t W, int32_t C, int32_t hidden, int32_t heads, int32_t sr) {
  if (B < 1 || !stage_kind(H, W, C, hidden, heads, sr)) return 0;
  const int G = H / rows_per_wg(B, H, 0);
  return device_cus() >= G ? G : 0;          // a sample's G workgroups must be co-resident (one per CU): else not supported here
}

extern "C" int crd_enc_stage_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t hidden, int32_t heads, int32_t sr) {
  const int kind = stage_kind(H, W, C, hidden, heads, sr);
  if (!kind || B < 1) return 0;
  long long need = 0;                        // (either choice of rows per workgroup fits)
  for (int rpw = 1; rpw <= 2; ++rpw) {
    const int G = H / rpw;
    const long long area = kind == 3 ? (rpw == 1 ? Cfg3<1>::AREA : Cfg3<2>::AREA) : (rpw == 1 ? Cfg4<1>::AREA : Cfg4<2>::AREA);
    const long long sets = sets_for<void>(B, G, device_cus() > 256 ? device_cus() : 256);      // (an upper bound: valid without a device too)
    if (sets * area > need) need = sets * area;
  }
  return (int)((EPOCH_WORDS + need) * 8);
}

extern "C" int crd_enc_stage_fwd(const crd_enc_stage_desc* d, crd_stream_t stream) {
  CRD_CHECK_ARG(d && d->x && d->blocks && d->sync_ws && d->status && d->nblocks >= 1, "crd_enc_stage_fwd: null pointer");
  const int kind = stage_kind(d->H, d->W, d->C, d->hidden, d->heads, d->sr);
  CRD_UNSUPPORTED(kind != 0 && d->B >= 1, "crd_enc_stage_fwd: shape not covered (H %d W %d C %d hidden %d heads %d sr %d): see crd_enc_stage_supported",
                  d->H, d->W, d->C, d->hidden, d->heads, d->sr);
  CRD_CHECK_ARG(d->rows_per_wg >= 0 && d->rows_per_wg <= 2, "crd_enc_stage_fwd: rows_per_wg must be 0 (choose), 1 or 2");
  const int rpw = rows_per_wg(d->B, d->H, d->rows_per_wg);
  CRD_UNSUPPORTED(device_cus() >= d->H / rpw, "crd_enc_stage_fwd: %d workgroups per sample cannot be co-resident on %d compute units",
                  d->H / rpw, device_cus());
  hipStream_t st = as_stream(stream);
  const int rc = kind == 3 ? (rpw == 1 ? launch_stage<Cfg3<1>>(d, st) : launch_stage<Cfg3<2>>(d, st))
                           : (rpw == 1 ? launch_stage<Cfg4<1>>(d, st) : launch_stage<Cfg4<2>>(d, st));
  if (rc != 0) return rc;
  CRD_LAUNCH_CHECK("crd_enc_stage_fwd");
  return CRD_OK;
}
