// Developer experiment (round 6, REJECTED -- not part of the library): the depthwise 3x3 convolution on the matrix pipe.
//
// Question: k_dwconv (csrc/encoder_ops.hip) is bound by its vector instruction count (36 v_pk_fma_f32 per output of 8 channels,
// 1.35 TB/s at stage 1); v_mfma_f32_4x4x4_16B_bf16 runs sixteen independent 4x4x4 products, one per channel -- does that remove the bound?
// Answer (MI355X, B = 8, forward with GroupNorm sums, us per launch; this file prints them):
//                                   64x104x512   32x52x1024   16x26x1280
//   k_dwconv (library)                 35.2         19.4          9.6
//   this kernel, one tile / workgroup  41.5         23.2         11.4
//   this kernel, persistent + prefetch 40.8         24.0         13.7
//   ... without the lane permutes      34.1         20.2         11.5     (-DDWM_NOPERM: wrong results, timing only)
// The multiplies disappear (6 MFMAs per 4 x 4 x 16-channel patch) but the operands have to be brought into the MFMA's lane layout
// (block = lane / 4 = channel): per patch 3 ds_read_b64_tr_b16 + 6 ds_bpermute_b32 + 4 ds_write_b16 of the results, 13 LDS
// instructions where the vector kernel issues 3 ds_read_b128 per 8 outputs -- and the LDS pipe is shared by the CU's four SIMDs.
// Per-phase stamps (100 MHz counter, -DDWM_PROF) of a one-tile workgroup at stage 1: halo load + staging 2.9 us, matrix phase 3.6 us,
// store loop 1.25 us, sums 0.6 us.  Requesting the next tile's halo before the matrix phase (the persistent variant below) hides the
// load but the matrix phase grows to 4.6 us per tile: the two workgroups of a CU now overlap their LDS traffic.  Software-pipelining
// the reads and permutes three / two steps ahead did not help (3.6 us per tile, 46.7 us per launch, 256 VGPRs): the phase is bound
// by LDS issue, not by latency.  Even with the permutes removed entirely the launch only ties the vector kernel, so a layout that
// makes them unnecessary (rows interleaved in LDS, ky shifts by DPP quad rotations) was not built.
// Layout facts this rests on: tools/probe_mfma4.hip.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DDWM_PROF -Icamradepth_amd/csrc -Iinclude tools/exp_dwconv_mfma.hip -o tools/exp_dwconv_mfma
// run on the GPU box: ./tools/exp_dwconv_mfma      (checks the results against crd_dwconv3x3, then times both)
#include <cstdio>
#include <cstring>
#include <vector>
#include "../camradepth_amd/csrc/encoder_ops.hip"
// the symbols of api.hip the included file refers to
void crd_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }
void crd_note_attr_failure(const char* k, int bytes, int e) { fprintf(stderr, "LDS reservation refused: %s %d (%d)\n", k, bytes, e); }
int crd_report_attr_failure(const char*) { return 0; }
void crd_register_nonfinite_flag(void* (*)()) {}

namespace {
// ------------------------------------------------------------------------------------------------
// The same convolution on the matrix pipe (tiles 32 columns wide).  k_dwconv above is bound by its vector instruction count -- 36
// v_pk_fma_f32 per output of 8 channels, 1.35 TB/s at stage 1 -- and no re-tiling of a VALU kernel removes the multiplies.
// v_mfma_f32_4x4x4_16B_bf16 runs SIXTEEN independent 4x4x4 products, and a depthwise row convolution is one of them per channel:
//   D[i][j] = sum_k A[i][k] * B[k][j],  A[i][k] = w[ky][k - i] (banded Toeplitz of the 3 taps of row ky, zero elsewhere),
//   B[k][j] = input pixel k of run j,  so D[i][j] = row-ky contribution to output column x0 + i of run j;
// runs j = 4 consecutive output rows, k = the 6 input columns x0-1 .. x0+4 split into two products (k = 0..3 with A1, 4..7 with A2,
// whose only non-zero entries are w[2] / w[1], w[2] in rows 2 / 3).  Six MFMAs (3 tap rows x 2 halves) give a 4 x 4 output patch of
// 16 channels; block = lane / 4 = channel, operand lane % 4 = row i of A / run j of B (layout probed on the GPU: tools/probe_mfma4.hip).
// The halo tile stays pixel-major in LDS exactly as k_dwconv stages it (coalesced 16-byte pieces, Mlp.norm1 applied on the way in);
// a B operand is ds_read_b64_tr_b16 (lane i of a 16-lane group receives 4 consecutive pixels of channel i; the four groups read the
// four runs) followed by two ds_bpermute that move (run j, channel c) from lane 16 j + c to lane 4 c + j.  The second half of one
// 4-column step is the first half of the next, so a step costs 3 transposing reads.  Products of two bf16 are exact in fp32 and the
// accumulation is fp32 like the vector kernel's; taps that are not bf16 values are split into hi + lo bf16 parts (a second set of
// MFMAs, skipped when every lo part of the wave is zero -- always, in the product path, whose taps are rounded by the weight pack).
// The 4 x 4 patches go to an LDS output tile as bf16; the store / GroupNorm-sum / fused-reduce phase then runs pixel-major with the
// thread mapping of k_dwconv.  A wave owns 16 channels of the 64-channel window.
// ------------------------------------------------------------------------------------------------
constexpr int MHW = 36;                                   // LDS halo row in pixels: 34 + 2 zeroed pad pixels (the last step reads columns 32..35)
constexpr int MIN_BYTES = (DTH + 2) * MHW * 128;          // 46080
constexpr int MOUT_ROW = 32 * 128 + 32;                   // output tile row pitch: +8 banks per row, the four runs of a ds_write_b16 hit different banks
constexpr int DWM_LDS = MIN_BYTES + DTH * MOUT_ROW;       // 79104 B: two workgroups per CU

__device__ __forceinline__ s16x4 dwm_tr_read(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ s16x4 dwm_perm(int sel, s16x4 v) {
#ifdef DWM_NOPERM
  return v;      // timing experiment only (wrong results): what the phase costs without the lane permutes
#endif
  typedef __attribute__((ext_vector_type(2))) int i32x2;
  i32x2 a = __builtin_bit_cast(i32x2, v);
  a[0] = __builtin_amdgcn_ds_bpermute(sel, a[0]);
  a[1] = __builtin_amdgcn_ds_bpermute(sel, a[1]);
  return __builtin_bit_cast(s16x4, a);
}

#ifdef DWM_PROF
__device__ unsigned dwm_prof[4096 * 8];       // [workgroup][phase] ticks of the 100 MHz counter
#define DWM_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); dwm_prof[(blockIdx.x & 4095) * 8 + k] += (unsigned)(n_ - stamp_); stamp_ = n_; } } while (0)
#else
#define DWM_STAMP(k) do {} while (0)
#endif

struct DwTile { int b, cw, ty0, tx0; };

template <bool FLIP, bool STATS>
__global__ __launch_bounds__(TPB, 2) void k_dwconv_mfma(const bf16_t* x, int B, int H, int W, int C, const float* w9, const float* bias,
                                                        bf16_t* y, crd_sum_t* stats, int tiles_x, int tiles_y, int per, int total,
                                                        InNorm inn, RedOut red) {
  constexpr int TW = 32, HWD = TW + 2, HPX = (DTH + 2) * HWD, NPC = (HPX * 8 + TPB - 1) / TPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char dwm_lds[];
  uint4* sh = reinterpret_cast<uint4*>(dwm_lds);          // [10][MHW][8 granules]
  unsigned char* so = dwm_lds + MIN_BYTES;                // [8][32 px + pad][64 ch] bf16
  __shared__ float sred[4][16];
  const int t = threadIdx.x;
  const int g = t & 7, xc = t >> 3;                        // staging granule (256 % 8 == 0: the same for every piece) / store-loop column
  const int l = t & 63, wv = t >> 6;
  const int cl = l >> 2, li = l & 3;                       // MFMA block (channel of the wave's 16) and row / run within it
  const int first = blockIdx.x * per;
  const int last = first + per < total ? first + per : total;
  const long long P = (long long)H * W;
#ifdef DWM_PROF
  unsigned long long stamp_ = __builtin_amdgcn_s_memrealtime();
#endif
  // the workgroup walks over `per` consecutive tiles (columns fastest, then rows, samples, channel windows): the halo of tile n + 1
  // is REQUESTED before the matrix phase of tile n and written to LDS after it, so that the memory latency of a tile (2.9 us of the
  // 8.3 us a one-tile workgroup took at stage 1) runs under the arithmetic of the one before
  DwTile cur;
  {
    int r = first;
    const int txi = r % tiles_x; r /= tiles_x;
    const int tyi = r % tiles_y; r /= tiles_y;
    cur.b = r % B; cur.cw = (r / B) * DCW; cur.ty0 = tyi * DTH; cur.tx0 = txi * TW;
  }
  uint4 r[NPC];
  unsigned inmask = 0;
  auto issue_halo = [&](const DwTile& q) {
    const int nG = (C - q.cw) >= DCW ? 8 : (C - q.cw) >> 3;
    const bf16_t* xb = x + (long long)q.b * P * C + q.cw + (g < nG ? g : 0) * 8;
    inmask = 0;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int i = t + k * TPB;
      const int hp = i >> 3;
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const int iy = q.ty0 - 1 + hy, ix = q.tx0 - 1 + hx;
      const bool in = i < HPX * 8 && g < nG && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      inmask |= in ? (1u << k) : 0u;
      // unconditional loads from a clamped address (zeroed by a select when they are stored to LDS)
      const int cy = iy < 0 ? 0 : (iy < H ? iy : H - 1), cx = ix < 0 ? 0 : (ix < W ? ix : W - 1);
      r[k] = *reinterpret_cast<const uint4*>(xb + ((long long)cy * W + cx) * C);
    }
  };
  float na[8], ns[8];
  int nrm_b = -1, nrm_cw = -1, w_cw = -1;
  s16x4 ahi[3][2], alo[3][2];
  float bz = 0.f;
  bool need_lo = false;
  auto stage = [&](const DwTile& q) {
    if (q.b != nrm_b || q.cw != nrm_cw) {
      const int nG = (C - q.cw) >= DCW ? 8 : (C - q.cw) >> 3;
      innorm_coeffs(inn, q.b, C, P, q.cw + g * 8, g < nG, na, ns);
      nrm_b = q.b; nrm_cw = q.cw;
    }
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int i = t + k * TPB;
      const int hp = i >> 3, hy = hp / HWD, hx = hp - hy * HWD;
      const bool in = (inmask >> k) & 1;
      const uint4 v = in ? r[k] : make_uint4(0, 0, 0, 0);
      if (i < HPX * 8) sh[(hy * MHW + hx) * 8 + g] = (inn.stats && in) ? innorm_apply(v, na, ns) : v;   // the zero padding stays zero
    }
  };
  // A operands of channel window cw: row li of the banded tap matrices, k = 0..3 (half 0) and 4..7 (half 1), as hi + lo bf16 parts
  auto load_taps = [&](int cw) {
    const int nG = (C - cw) >= DCW ? 8 : (C - cw) >> 3;
    const bool wok = wv * 2 < nG;
    const int chm = cw + (wok ? wv * 16 + cl : 0);
    float wt[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wt[tp] = w9[(long long)(FLIP ? 8 - tp : tp) * C + chm];
    bz = (bias && wok) ? bias[chm] : 0.f;
    unsigned lo_any = 0;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] = 0.f;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) v[k] = (k + 4 * h - li == kx) ? wt[ky * 3 + kx] : v[k];
        }
        const unsigned h0 = pack_bf2(v[0], v[1]), h1 = pack_bf2(v[2], v[3]);
        const unsigned l0 = pack_bf2(v[0] - bf_lo(h0), v[1] - bf_hi(h0)), l1 = pack_bf2(v[2] - bf_lo(h1), v[3] - bf_hi(h1));
        ahi[ky][h] = __builtin_bit_cast(s16x4, (u32x2{h0, h1}));
        alo[ky][h] = __builtin_bit_cast(s16x4, (u32x2{l0, l1}));
        lo_any |= (l0 | l1) & 0x7fff7fffu;
      }
    need_lo = __any(lo_any != 0);
    w_cw = cw;
  };
  issue_halo(cur);
  load_taps(cur.cw);
  if (t < (DTH + 2) * 16) sh[((t >> 4) * MHW + HWD + ((t >> 3) & 1)) * 8 + (t & 7)] = make_uint4(0, 0, 0, 0);   // the two pad pixels of a row: written once
  stage(cur);
  __syncthreads();
  DWM_STAMP(0);
  for (int id = first; id < last; ++id) {
    const bool has_next = id + 1 < last;
    DwTile nxt = cur;
    if (has_next) {
      nxt.tx0 += TW;
      if (nxt.tx0 >= W) {
        nxt.tx0 = 0; nxt.ty0 += DTH;
        if (nxt.ty0 >= H) { nxt.ty0 = 0; if (++nxt.b == B) { nxt.b = 0; nxt.cw += DCW; } }
      }
    }
    issue_halo(nxt);             // (the last tile requests itself again: the loads stay unconditional)
    const int c_win = cur.cw, ty0 = cur.ty0, tx0 = cur.tx0, b = cur.b;
    const int nG = (C - c_win) >= DCW ? 8 : (C - c_win) >> 3;
    const int c0 = c_win + g * 8;
    const bool gok = g < nG;
    const bool wok = wv * 2 < nG;                          // the wave's 16 channels exist (C is a multiple of 16)
    bf16_t* yb = y + (long long)b * P * C;
    // raw input of the fused reduce at this thread's output positions: requested now, used in the store loop
    uint4 xrv[DTH];
    if (red.xr) {
      const int cgr = gok ? c0 : c_win;
#pragma unroll
      for (int i = 0; i < DTH; ++i) {
        const int oy = ty0 + i, ox = tx0 + xc;
        const int cy = oy < H ? oy : H - 1, cx = ox < W ? ox : W - 1;
        xrv[i] = *reinterpret_cast<const uint4*>(red.xr + (((long long)b * H + cy) * W + cx) * C + cgr);
      }
    }
    if (wok) {
      // transposing read: lane (group gq, row r, quarter q) supplies the address of channels 4q..4q+3 of pixel r of run gq
      const unsigned char* rd0 = dwm_lds + (((l >> 4) * MHW + ((l & 15) >> 2)) * 128 + wv * 32 + (l & 3) * 8);
      const int sel = ((l & 3) * 16 + (l >> 2)) << 2;         // (run j, channel c): lane 16 j + c -> lane 4 c + j
      unsigned char* wr0 = so + li * MOUT_ROW + (wv * 16 + cl) * 2;
#pragma unroll
      for (int yg = 0; yg < 2; ++yg) {
        s16x4 bc[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) bc[ky] = dwm_perm(sel, dwm_tr_read(rd0 + ((4 * yg + ky) * MHW) * 128));
#pragma unroll
        for (int xg = 0; xg < 8; ++xg) {
          s16x4 bn[3];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) bn[ky] = dwm_perm(sel, dwm_tr_read(rd0 + ((4 * yg + ky) * MHW + 4 * (xg + 1)) * 128));
          f32x4 acc = {bz, bz, bz, bz};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ahi[ky][0], bc[ky], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ahi[ky][1], bn[ky], acc, 0, 0, 0);
          }
          if (need_lo) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(alo[ky][0], bc[ky], acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(alo[ky][1], bn[ky], acc, 0, 0, 0);
            }
          }
          // D: lane (channel cl, run li) holds output columns 4 xg .. 4 xg + 3 of row 4 yg + li
          const unsigned u0 = pack_bf2(acc[0], acc[1]), u1 = pack_bf2(acc[2], acc[3]);
          unsigned char* wp = wr0 + (4 * yg) * MOUT_ROW + (4 * xg) * 128;
          *reinterpret_cast<unsigned short*>(wp) = (unsigned short)u0;
          *reinterpret_cast<unsigned short*>(wp + 128) = (unsigned short)(u0 >> 16);
          *reinterpret_cast<unsigned short*>(wp + 256) = (unsigned short)u1;
          *reinterpret_cast<unsigned short*>(wp + 384) = (unsigned short)(u1 >> 16);
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) bc[ky] = bn[ky];
        }
      }
    }
    __syncthreads();             // every wave is done with the halo image: the next one may be written
    DWM_STAMP(1);
    if (has_next) {
      if (nxt.cw != w_cw) load_taps(nxt.cw);
      stage(nxt);
    }
    DWM_STAMP(2);
    // ---- output tile -> memory (+ sums), thread (column xc, granule g) walking down the rows like k_dwconv
    float s = 0.f, ss = 0.f;
    float rs0[8], rs1[8], rmean = 0.f, rrstd = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) rs0[j] = rs1[j] = 0.f;
    if (red.xr && gok) gn_mean_rstd(red.stats + (long long)b * (C >> 4) * 2, c0 >> 4, 1, (float)H * W * 16.f, rmean, rrstd);
#pragma unroll
    for (int i = 0; i < DTH; ++i) {
      const int oy = ty0 + i, ox = tx0 + xc;
      const uint4 u = *reinterpret_cast<const uint4*>(so + i * MOUT_ROW + xc * 128 + g * 16);
      if (gok && oy < H && ox < W) {
        *reinterpret_cast<uint4*>(yb + ((long long)oy * W + ox) * C + c0) = u;
        if (red.xr) {
          const uint4 xv = xrv[i];
          const float gq[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
          const float xq[8] = {bf_lo(xv.x), bf_hi(xv.x), bf_lo(xv.y), bf_hi(xv.y), bf_lo(xv.z), bf_hi(xv.z), bf_lo(xv.w), bf_hi(xv.w)};
#pragma unroll
          for (int j = 0; j < 8; ++j) { rs0[j] += gq[j]; rs1[j] += gq[j] * ((xq[j] - rmean) * rrstd); }
        }
        if (STATS) {
          s += bf_lo(u.x) + bf_hi(u.x) + bf_lo(u.y) + bf_hi(u.y) + bf_lo(u.z) + bf_hi(u.z) + bf_lo(u.w) + bf_hi(u.w);
          ss += bf_lo(u.x) * bf_lo(u.x) + bf_hi(u.x) * bf_hi(u.x) + bf_lo(u.y) * bf_lo(u.y) + bf_hi(u.y) * bf_hi(u.y) +
                bf_lo(u.z) * bf_lo(u.z) + bf_hi(u.z) * bf_hi(u.z) + bf_lo(u.w) * bf_lo(u.w) + bf_hi(u.w) * bf_hi(u.w);
        }
      }
    }
    if (STATS) {
      s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
      if (l < 8 && (l & 1) == 0) { sred[wv][(l >> 1) * 2] = s; sred[wv][(l >> 1) * 2 + 1] = ss; }
      __syncthreads();
      if (t < 8) {
        const float v = sred[0][t] + sred[1][t] + sred[2][t] + sred[3][t];
        const int slab = (c_win >> 4) + (t >> 1);
        if (slab < (C >> 4)) stat_add(&stats[((long long)b * (C >> 4) + slab) * 2 + (t & 1)], v);
      }
    }
    if (red.xr) {
      // fold the 8 columns of the wave, then the 4 waves through LDS (the output tile has been read): [wave][128]
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { rs0[j] += __shfl_xor(rs0[j], o); rs1[j] += __shfl_xor(rs1[j], o); }
      }
      __syncthreads();
      float* fr = reinterpret_cast<float*>(so);
      if (l < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { fr[wv * 128 + (l * 8 + j) * 2] = rs0[j]; fr[wv * 128 + (l * 8 + j) * 2 + 1] = rs1[j]; }
      }
      __syncthreads();
      const int nch = nG * 8;
      if (t < 2 * nch) {
        const float v = fr[t] + fr[128 + t] + fr[256 + t] + fr[384 + t];
        const int c = c_win + (t >> 1);
        grad_add(&red.r[((long long)b * C + c) * 2 + (t & 1)], v);
        fr[512 + t] = v * red.gamma[c];
      }
      __syncthreads();
      if (t < 2 * (nch >> 4)) {
        const int grp = t >> 1, which = t & 1;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) a += fr[512 + (grp * 16 + j) * 2 + which];
        grad_add(&red.r[(long long)B * C * 2 + ((long long)b * (C >> 4) + (c_win >> 4) + grp) * 2 + which], a);
      }
    }
    __syncthreads();             // the next halo image is complete, the output tile and the fold scratch are free
    DWM_STAMP(3);
    cur = nxt;
  }
}


int launch_mfma(const void* x, int B, int H, int W, int C, const float* w9, const float* bias, void* y, crd_sum_t* stats) {
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, DTH);
  const int total = tiles_x * tiles_y * cdiv(C, DCW) * B;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int per = cdiv(total, 2 * cus);
  static bool done = false;
  if (!done) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwconv_mfma<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DWM_LDS); done = true; }
  const InNorm inn{nullptr, nullptr, nullptr, 1};
  const RedOut red{nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL((k_dwconv_mfma<false, true>), dim3(cdiv(total, per)), dim3(TPB), DWM_LDS, 0, reinterpret_cast<const bf16_t*>(x), B, H, W, C, w9, bias,
                     reinterpret_cast<bf16_t*>(y), stats, tiles_x, tiles_y, per, total, inn, red);
  return cdiv(total, per);
}
}  // namespace

int main() {
  const int B = 8;
  const int shapes[3][3] = {{64, 104, 512}, {32, 52, 1024}, {16, 26, 1280}};
  for (auto& sh : shapes) {
    const int H = sh[0], W = sh[1], C = sh[2];
    const size_t n = (size_t)B * H * W * C;
    std::vector<unsigned short> hx(n), y0(n), y1(n);
    for (size_t i = 0; i < n; ++i) hx[i] = 0x3f80 ^ (unsigned short)((i * 2654435761u) >> 23 & 0x81ff);     // +-[1, 4)
    void *x, *y; float *w9, *bias; crd_sum_t* stats;
    const size_t sbytes = (size_t)B * (C / 16) * 2 * sizeof(crd_sum_t);
    hipMalloc(&x, n * 2); hipMalloc(&y, n * 2); hipMalloc(&w9, 9 * C * 4); hipMalloc(&bias, C * 4); hipMalloc(&stats, sbytes);
    hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice);
    std::vector<float> hw(9 * C), hb(C);
    for (int i = 0; i < 9 * C; ++i) hw[i] = (float)((i * 37 % 33) - 16) / 32.f;          // bf16 values, as the weight pack leaves them
    for (int i = 0; i < C; ++i) hb[i] = (float)(i % 7) / 8.f;
    hipMemcpy(w9, hw.data(), 9 * C * 4, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice);
    // results: library kernel against the matrix-pipe kernel
    std::vector<crd_sum_t> s0(sbytes / sizeof(crd_sum_t)), s1(s0.size());
    hipMemset(stats, 0, sbytes); hipMemset(y, 0, n * 2);
    crd_dwconv3x3(x, B, H, W, C, w9, bias, 0, y, stats, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    hipMemcpy(y0.data(), y, n * 2, hipMemcpyDeviceToHost); hipMemcpy(s0.data(), stats, sbytes, hipMemcpyDeviceToHost);
    hipMemset(stats, 0, sbytes); hipMemset(y, 0, n * 2);
    const int nwg = launch_mfma(x, B, H, W, C, w9, bias, y, stats);
    hipMemcpy(y1.data(), y, n * 2, hipMemcpyDeviceToHost); hipMemcpy(s1.data(), stats, sbytes, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t i = 0; i < n; ++i) diff += y0[i] != y1[i];
    printf("%dx%dx%dx%d: %zu of %zu outputs differ from the library kernel's (fp32 summation order), sums %s\n", B, H, W, C, diff, n,
           memcmp(s0.data(), s1.data(), sbytes) == 0 ? "equal" : "differ");
    for (int mfma = 0; mfma < 2; ++mfma) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto go = [&]() {
        if (mfma) launch_mfma(x, B, H, W, C, w9, bias, y, stats);
        else crd_dwconv3x3(x, B, H, W, C, w9, bias, 0, y, stats, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      };
      for (int i = 0; i < 3; ++i) go();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int R = 20;
      for (int i = 0; i < R; ++i) go();
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("  %s: %.2f us/launch", mfma ? "k_dwconv_mfma" : "k_dwconv     ", ms / R * 1e3);
#ifdef DWM_PROF
      if (mfma) {
        std::vector<unsigned> z(4096 * 8, 0);
        hipMemcpyToSymbol(HIP_SYMBOL(dwm_prof), z.data(), z.size() * 4);
        go(); hipDeviceSynchronize();
        hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(dwm_prof), z.size() * 4);
        double ph[4] = {0, 0, 0, 0};
        for (int w = 0; w < nwg && w < 4096; ++w) for (int k = 0; k < 4; ++k) ph[k] += z[w * 8 + k];
        const double wgs = (nwg < 4096 ? nwg : 4096) * 100.;      // ticks -> us
        printf("; per workgroup (us, all its tiles): prologue %.2f  matrix phases %.2f  staging %.2f  store loops + sums %.2f", ph[0] / wgs, ph[1] / wgs, ph[2] / wgs, ph[3] / wgs);
      }
#endif
      printf("\n");
    }
    hipFree(x); hipFree(y); hipFree(w9); hipFree(bias); hipFree(stats);
  }
  return 0;
}
