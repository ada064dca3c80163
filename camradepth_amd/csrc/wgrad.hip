// Convolution weight gradient on MFMA for gfx950.
//
//   dw[co][kf] += sum_p dy[p][co] * xcol[p][kf]        p = (b, oy, ox) flattened, kf = (tap, ci)
//
// Both operands are pixel-major, i.e. the contraction index p is the SLOW dimension of both
// tiles.  v_mfma_f32_16x16x32_bf16 wants 8 consecutive-k values per lane, so the fragments are
// read from the row-major [pixel][channel] LDS tiles with ds_read_b64_tr_b16 (the gfx950
// transposing LDS read: a 16-lane group fetches a 4x16 block and lane i receives column i),
// which costs the same as a plain read and removes any explicit transpose.  The pixel range is
// split across workgroups (split-K); partial tiles meet in a 64-bit fixed-point accumulator (integer
// atomics: the sum does not depend on the order of arrival, see crd_sum_t).
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "common.h"

int crd_wgrad3x3_stream(const crd_wgrad_desc* d, hipStream_t st);   // wgrad3x3.hip
int crd_wgrad3x3_splits(const crd_wgrad_desc* d);

namespace {

struct WgK {
  const bf16_t* x; int x_ld; int IH, IW, Cin;
  const bf16_t* dy; int dy_ld; int OH, OW, Cout;
  int KW, stride, pad, Ktot;
  long long P;           // B*OH*OW
  long long x_bytes, dy_bytes;
  int chunk;             // pixels per split (multiple of 32)
  crd_sum_t* dw;         // fixed-point sums (CRD_GRAD_FRAC_BITS): split-K partial tiles meet here in any order
  crd_sum_t* dbias;      // or nullptr; accumulated by the workgroups of the first kf tile
  int dbg;               // developer experiments (CRD_DBG): 1 skip output atomics, 2 skip loads
};

constexpr int PK = 64;  // pixels per K-step (two 32-pixel MFMA sub-steps)

__device__ __forceinline__ s16x4 tr_read(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

template <int WMc, int WNc, int TMc, int TNc>
__device__ __forceinline__ void wgrad_tile(const WgK& a, const int bx, const int by, const int bz) {
  static_assert(WMc * WNc == 4, "4 waves");
  constexpr int BMc = WMc * TMc * 16, BNk = WNc * TNc * 16;
  static_assert(BNk == 128, "kf tile is 128 wide");
  constexpr int LDY = BMc + 16, LDX = BNk + 16;
  constexpr int GY = BMc / 8;                       // 16-byte granules per dy row
  constexpr int Y_IT = (PK * GY + 255) / 256;
  constexpr int X_IT = PK / 16;                      // 16 granules per row, 16 rows per pass of 256 threads
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * PK * (LDY + LDX)];
  bf16_t* sY = lds;
  bf16_t* sX = lds + 2 * PK * LDY;

  const int t = threadIdx.x, l = t & 63, wave = t >> 6;
  const int wm = wave / WNc, wn = wave % WNc;
  const int n0 = bx * BNk, m0 = by * BMc;
  const long long p_begin = (long long)bz * a.chunk;
  long long p_end = p_begin + a.chunk;
  if (p_end > a.P) p_end = a.P;
  if (p_begin >= p_end) return;
  const int nK = (int)((p_end - p_begin + PK - 1) / PK);

  // hardware-bounds-checked buffer loads (out-of-range offset -> zeros): padding, K tail and partial tiles
  // need no branch in the load path
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rys = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
  // ---- X gather state: fixed kf granule per thread, two pixel rows ----
  const int xg = t & 15, xr0 = t >> 4;
  const int kf = n0 + xg * 8;
  const bool kok = kf < a.Ktot;
  const int tap = kf / a.Cin, kc = kf - tap * a.Cin;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
  int xb[X_IT], xoy[X_IT], xox[X_IT];
  const int OHW = a.OH * a.OW;
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    long long p = p_begin + xr0 + 16 * i;
    int b = (int)(p / OHW);
    int rem = (int)(p - (long long)b * OHW);
    xb[i] = b; xoy[i] = rem / a.OW; xox[i] = rem - xoy[i] * a.OW;
  }
  // dy tile: thread -> (row, granule) fixed for the whole loop
  int yr[Y_IT], yc[Y_IT];
  bool yok[Y_IT];
#pragma unroll
  for (int i = 0; i < Y_IT; ++i) {
    const int idx = t + 256 * i;
    yr[i] = idx / GY;
    yc[i] = idx - yr[i] * GY;
    yok[i] = idx < PK * GY && (m0 + yc[i] * 8) < a.Cout;
  }

  u32x4 rx[X_IT], ry[Y_IT];
  int kt_load = 0;
  auto gload = [&]() {
    const long long pbase = p_begin + (long long)kt_load * PK;
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const long long p = pbase + xr0 + 16 * i;
      const int iy = xoy[i] * a.stride - a.pad + ky, ix = xox[i] * a.stride - a.pad + kx;
      const bool ok = kok && p < p_end && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const unsigned off = ok ? (unsigned)((((xb[i] * a.IH + iy) * a.IW + ix) * a.x_ld + kc) * 2) : OOB;
      rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rxs, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const long long p = pbase + yr[i];
      const unsigned off = (yok[i] && p < p_end) ? (unsigned)((p * a.dy_ld + m0 + yc[i] * 8) * 2) : OOB;
      ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rys, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      xox[i] += PK;
      while (xox[i] >= a.OW) { xox[i] -= a.OW; if (++xoy[i] == a.OH) { xoy[i] = 0; ++xb[i]; } }
    }
    ++kt_load;
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < X_IT; ++i)
      *reinterpret_cast<u32x4*>(&sX[buf * PK * LDX + (xr0 + 16 * i) * LDX + xg * 8]) = rx[i];
#pragma unroll
    for (int i = 0; i < Y_IT; ++i)
      if (t + 256 * i < PK * GY) *reinterpret_cast<u32x4*>(&sY[buf * PK * LDY + yr[i] * LDY + yc[i] * 8]) = ry[i];
  };

  f32x4 acc[TMc][TNc];
#pragma unroll
  for (int i = 0; i < TMc; ++i)
#pragma unroll
    for (int j = 0; j < TNc; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool do_bias = a.dbias != nullptr && bx == 0 && t < BMc;
  float bsum = 0.f;
  gload();
  lstore(0);
  __syncthreads();
  // per-lane fragment addressing: 16-lane group q covers pixels 8q..8q+7; inside a read the lane
  // supplies the address of 4 consecutive channels of pixel row (l&15)>>2.
  const int q = l >> 4, prow = (l & 15) >> 2, pcol = (l & 3) * 4;
  for (int kt = 0; kt < nK; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nK && !(a.dbg & 2)) gload();
#pragma unroll
    for (int sub = 0; sub < PK / 32; ++sub) {
      bf16x8 af[TMc], bfr[TNc];
#pragma unroll
      for (int i = 0; i < TMc; ++i) {
        const bf16_t* base = &sY[cur * PK * LDY + (32 * sub + 8 * q + prow) * LDY + (wm * TMc + i) * 16 + pcol];
        s16x4 lo = tr_read(base), hi = tr_read(base + 4 * LDY);
        union { bf16x8 v; s16x4 h[2]; } u;
        u.h[0] = lo; u.h[1] = hi;
        af[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < TNc; ++j) {
        const bf16_t* base = &sX[cur * PK * LDX + (32 * sub + 8 * q + prow) * LDX + (wn * TNc + j) * 16 + pcol];
        s16x4 lo = tr_read(base), hi = tr_read(base + 4 * LDX);
        union { bf16x8 v; s16x4 h[2]; } u;
        u.h[0] = lo; u.h[1] = hi;
        bfr[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < TMc; ++i)
#pragma unroll
        for (int j = 0; j < TNc; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias) {       // column sums of the dy tile (rows beyond p_end were loaded as zeros)
#pragma unroll 8
      for (int r = 0; r < PK; ++r) bsum += bf2f(sY[cur * PK * LDY + r * LDY + t]);
    }
    if (kt + 1 < nK) lstore(cur ^ 1);
    __syncthreads();
  }
  // (fx_add casts to address space 1: the grouped kernel loads its descriptor from memory, so the compiler no longer knows
  // that dw / dbias are global pointers and would emit FLAT atomics)
  if (do_bias && m0 + t < a.Cout) grad_add(a.dbias + m0 + t, bsum);

  // D layout (16x16): col = lane&15 -> kf, row = (lane>>4)*4 + r -> co
#pragma unroll
  for (int i = 0; i < TMc; ++i)
#pragma unroll
    for (int j = 0; j < TNc; ++j) {
      const int kfo = n0 + (wn * TNc + j) * 16 + (l & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = m0 + (wm * TMc + i) * 16 + (l >> 4) * 4 + r;
        if (co < a.Cout && kfo < a.Ktot && (!(a.dbg & 1) || acc[i][j][r] == 123.456f)) grad_add(a.dw + (long long)co * a.Ktot + kfo, acc[i][j][r]);
      }
    }
}

template <int WMc, int WNc, int TMc, int TNc>
__global__ __launch_bounds__(256) void k_wgrad(WgK a) {
  wgrad_tile<WMc, WNc, TMc, TNc>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Grouped launch: many small weight-gradient problems in ONE dispatch.  items[i] = (problem, kf tile, co tile, split);
// the problem descriptors live in device memory (built once by crd_wgrad_group_build).  The encoder's ~190 small
// wgrads are each latency-bound on their own (a few dozen workgroups, 4 serial K-steps); together they fill the chip.
template <int WMc, int WNc, int TMc, int TNc>
__global__ __launch_bounds__(256) void k_wgrad_grouped(const WgK* __restrict__ probs, const int4* __restrict__ items) {
  const int4 it = items[blockIdx.x];
  if (it.x < 0) return;                               // padding of the XCD-interleaved item list
  const WgK a = probs[it.x];
  wgrad_tile<WMc, WNc, TMc, TNc>(a, it.y, it.z, it.w);
}

// Split-K plan: sets k.chunk, returns the number of splits.  `want` = splits that would fill the chip, `min_steps` =
// K-steps a workgroup must at least run.
long long plan_splits(WgK& k, int tiles, long long want, int min_steps) {
  long long steps = (k.P + PK - 1) / PK;
  long long max_splits = (steps + min_steps - 1) / min_steps;
  long long splits = want < 1 ? 1 : want;
  if (splits > max_splits) splits = max_splits;
  // every split adds Cout x Ktot 64-bit atomics: keep the total around 1M per launch (2M with the fp32 atomics of round 2: 19.90 vs 19.72 ms per step) (L2 sustains ~170 G atomics/s)
  static long long atom_budget = -1;
  if (atom_budget < 0) atom_budget = crd_dev_int("CRD_WGRAD_ATOMS", 1 << 20);
  const long long atom_cap = atom_budget / ((long long)k.Cout * k.Ktot) + 1;
  if (splits > atom_cap) splits = atom_cap;
  if (splits < 1) splits = 1;
  long long chunk_steps = (steps + splits - 1) / splits;
  k.chunk = (int)(chunk_steps * PK);
  return (steps + chunk_steps - 1) / chunk_steps;
}

template <int WMc, int WNc, int TMc, int TNc>
int launch(const WgK& k0, hipStream_t st) {
  WgK k = k0;
  constexpr int BMc = WMc * TMc * 16, BNk = WNc * TNc * 16;
  const int tn = cdiv(k.Ktot, BNk), tm = cdiv(k.Cout, BMc);
  const long long splits = plan_splits(k, tn * tm, (1536 + tn * tm - 1) / (tn * tm), 4);   // ~6 workgroups per CU in flight
  dim3 grid(tn, tm, (unsigned)splits);
  hipLaunchKernelGGL((k_wgrad<WMc, WNc, TMc, TNc>), grid, dim3(256), 0, st, k);
  CRD_LAUNCH_CHECK("crd_conv_wgrad");
  return CRD_OK;
}

int fill(const crd_wgrad_desc* d, WgK& k) {
  CRD_CHECK_ARG(d && d->x && d->dy && d->dw, "crd_conv_wgrad: null pointer");
  CRD_CHECK_ARG(d->Cin % 8 == 0 && d->x_ld % 8 == 0 && d->x_coff % 8 == 0, "crd_conv_wgrad: x channels must be multiples of 8");
  CRD_CHECK_ARG(d->dy_ld % 8 == 0 && d->dy_coff % 8 == 0, "crd_conv_wgrad: dy_ld/dy_coff must be multiples of 8");
  CRD_CHECK_ARG(d->Cout % 8 == 0 || d->dy_ld - d->dy_coff >= ((d->Cout + 7) / 8) * 8,
                "crd_conv_wgrad: dy rows must hold Cout rounded up to 8 channels");
  k.x = reinterpret_cast<const bf16_t*>(d->x) + d->x_coff; k.x_ld = d->x_ld; k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin;
  k.dy = reinterpret_cast<const bf16_t*>(d->dy) + d->dy_coff; k.dy_ld = d->dy_ld; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout;
  k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.Ktot = d->KH * d->KW * d->Cin;
  k.P = (long long)d->B * d->OH * d->OW; k.chunk = 0; k.dw = d->dw;
  k.x_bytes = (long long)d->B * d->IH * d->IW * d->x_ld * 2;
  k.dy_bytes = k.P * d->dy_ld * 2;
  k.dbias = d->dbias; k.dbg = 0;
  return CRD_OK;
}
int check_generic(const WgK& k) {   // the generic split-K kernel addresses the whole batch with 32-bit byte offsets
  CRD_UNSUPPORTED(k.x_bytes < (1ll << 31) && k.dy_bytes < (1ll << 31), "crd_conv_wgrad: tensor too large for 32-bit byte offsets");
  return CRD_OK;
}

// tile configuration of the generic kernel by output-channel count (index into the launch tables)
int cfg_of(int cout) { return cout <= 32 ? 0 : cout <= 64 ? 1 : cout <= 96 ? 2 : 3; }
const int CFG_BM[4] = {32, 64, 96, 128};

}  // namespace

extern "C" int crd_wgrad_group_build(const crd_wgrad_desc* descs, int32_t n, void* host_table, int64_t capacity,
                                     crd_wgrad_group_info* info) {
  CRD_CHECK_ARG(descs && info && n > 0, "crd_wgrad_group_build: null pointer / empty group");
  // Item order (round 6): workgroups are dealt to the 8 XCDs round-robin by index, and the tiles of one (problem, K split) re-read the same
  // x and dy rows -- tn x tm of them, e.g. 30 for a stage-3 fc layer.  Dealt in problem order each XCD's L2 fetched its own copy
  // (the grouped launches fetched 457 MB each on average, the stage-3 group's unique operands are ~280 MB: profiles/r06_pmc_traffic.json).  So the units
  // (problem, split) are binned onto the XCDs -- largest first onto the least-loaded bin -- and the item list interleaves the bins:
  // item 8 j + x is the j-th item of XCD x's bin; bins shorter than the longest end in no-op items (problem index -1).  Where whole
  // units do not balance over the XCDs the pieces get finer (rows of tiles, then single tiles = the plain order).
  constexpr int NX = 8;
  const long long head = ((long long)n * sizeof(WgK) + 15) / 16 * 16;
  struct Unit { int prob, split, tn, tm; };
  std::vector<WgK> ks((size_t)n);
  std::vector<Unit> units[4];
  for (int i = 0; i < n; ++i) {
    WgK k;
    int rc = fill(&descs[i], k);
    if (rc == CRD_OK) rc = check_generic(k);
    if (rc != CRD_OK) return rc;
    const int c = cfg_of(k.Cout);
    const int tn = cdiv(k.Ktot, 128), tm = cdiv(k.Cout, CFG_BM[c]);
    // the group as a whole fills the chip: long K runs per workgroup (8 steps) keep the atomics per problem low.  (No K split at all for a
    // launch whose tiles alone are two workgroups per CU -- stage 3: 690 tiles -- measured 218 -> 204 us for that launch and nothing in
    // the step; not kept: one round of long-K workgroups quantises badly at other sizes.)
    const long long splits = plan_splits(k, tn * tm, 1 << 20, 8);
    ks[(size_t)i] = k;
    for (long long sp = 0; sp < splits; ++sp) units[c].push_back(Unit{i, (int)sp, tn, tm});
  }
  std::vector<int4> lists[4];
  long long off = 0;
  for (int c = 0; c < 4; ++c) {
    std::vector<Unit>& u = units[c];
    std::stable_sort(u.begin(), u.end(), [](const Unit& p, const Unit& q) { return p.tn * p.tm > q.tn * q.tm; });
    // granularity 0: a whole (problem, split) per bin entry; 1: one row of tiles (same dy columns, all of x) when the whole units do not
    // balance -- a group of a few large problems must not be confined to a few XCDs; 2: single tiles, i.e. the plain round-robin order
    for (int gran = 0; gran < 3; ++gran) {
      std::vector<int4> bins[NX];
      for (const Unit& w : u) {
        const int pieces = gran == 0 ? 1 : gran == 1 ? w.tm : w.tm * w.tn;
        for (int pc = 0; pc < pieces; ++pc) {
          int best = 0;
          for (int x = 1; x < NX; ++x)
            if (bins[x].size() < bins[best].size()) best = x;
          const int m0 = gran == 0 ? 0 : gran == 1 ? pc : pc / w.tn, m1 = gran == 0 ? w.tm : m0 + 1;
          const int t0 = gran == 2 ? pc % w.tn : 0, t1 = gran == 2 ? t0 + 1 : w.tn;
          for (int m = m0; m < m1; ++m)
            for (int t = t0; t < t1; ++t) bins[best].push_back(make_int4(w.prob, t, m, w.split));
        }
      }
      size_t longest = 0, total = 0;
      for (int x = 0; x < NX; ++x) { longest = bins[x].size() > longest ? bins[x].size() : longest; total += bins[x].size(); }
      if (gran < 2 && longest * NX > total + total / 8 + NX) continue;          // more than ~12 % of the launch would idle: finer pieces
      lists[c].clear();
      for (size_t j = 0; j < longest; ++j)
        for (int x = 0; x < NX; ++x) lists[c].push_back(j < bins[x].size() ? bins[x][j] : make_int4(-1, 0, 0, 0));
      while (!lists[c].empty() && lists[c].back().x < 0) lists[c].pop_back();    // (trailing no-ops)
      break;
    }
    info->item_offset[c] = (int32_t)off;
    info->n_items[c] = (int32_t)lists[c].size();
    off += (long long)lists[c].size();
  }
  info->n_problems = n;
  info->bytes = head + off * (long long)sizeof(int4);
  if (host_table == nullptr) return CRD_OK;                                // size query
  CRD_CHECK_ARG(capacity >= info->bytes, "crd_wgrad_group_build: the table needs %lld bytes, capacity is %lld", (long long)info->bytes,
                (long long)capacity);                                      // (was taken for a size query: rc 0, nothing written)
  WgK* probs = reinterpret_cast<WgK*>(host_table);
  for (int i = 0; i < n; ++i) probs[i] = ks[(size_t)i];
  int4* items = reinterpret_cast<int4*>(reinterpret_cast<char*>(host_table) + head);
  for (int c = 0; c < 4; ++c)
    for (size_t j = 0; j < lists[c].size(); ++j) items[info->item_offset[c] + (long long)j] = lists[c][j];
  return CRD_OK;
}

extern "C" int crd_conv_wgrad_grouped(const void* dev_table, const crd_wgrad_group_info* info, crd_stream_t stream) {
  CRD_CHECK_ARG(dev_table && info, "crd_conv_wgrad_grouped: null pointer");
  const WgK* probs = reinterpret_cast<const WgK*>(dev_table);
  const long long head = ((long long)info->n_problems * sizeof(WgK) + 15) / 16 * 16;
  const int4* items = reinterpret_cast<const int4*>(reinterpret_cast<const char*>(dev_table) + head);
  hipStream_t st = as_stream(stream);
  if (info->n_items[0]) hipLaunchKernelGGL((k_wgrad_grouped<1, 4, 2, 2>), dim3(info->n_items[0]), dim3(256), 0, st, probs, items + info->item_offset[0]);
  if (info->n_items[1]) hipLaunchKernelGGL((k_wgrad_grouped<1, 4, 4, 2>), dim3(info->n_items[1]), dim3(256), 0, st, probs, items + info->item_offset[1]);
  if (info->n_items[2]) hipLaunchKernelGGL((k_wgrad_grouped<2, 2, 3, 4>), dim3(info->n_items[2]), dim3(256), 0, st, probs, items + info->item_offset[2]);
  if (info->n_items[3]) hipLaunchKernelGGL((k_wgrad_grouped<2, 2, 4, 4>), dim3(info->n_items[3]), dim3(256), 0, st, probs, items + info->item_offset[3]);
  CRD_LAUNCH_CHECK("crd_conv_wgrad_grouped");
  return CRD_OK;
}

static bool uses_stream3(const crd_wgrad_desc* d) {
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->IH == d->OH && d->IW == d->OW && d->IW >= 32 &&
         d->IH >= 8 && !crd_dev_int("CRD_NO_WGRAD3", 0);
}

extern "C" int crd_conv_wgrad_splits(const crd_wgrad_desc* d) {
  if (!d || !uses_stream3(d)) return 0;
  return crd_wgrad3x3_splits(d);
}

extern "C" int crd_conv_wgrad(const crd_wgrad_desc* d, crd_stream_t stream) {
  WgK k;
  { int rc = fill(d, k); if (rc != CRD_OK) return rc; }
  hipStream_t st = as_stream(stream);
  // 3x3 / stride 1 / pad 1 on grids at least one 32-pixel strip wide: streaming halo-row kernel (wgrad3x3.hip)
  if (uses_stream3(d)) return crd_wgrad3x3_stream(d, st);
  CRD_CHECK_ARG(d->dw_partials == nullptr, "crd_conv_wgrad: dw_partials is only supported where crd_conv_wgrad_splits() > 0");
  { int rc = check_generic(k); if (rc != CRD_OK) return rc; }
  { static int dbg = -1; if (dbg < 0) dbg = crd_dev_int("CRD_DBG", 0); k.dbg = dbg; }
  if (d->Cout <= 32) return launch<1, 4, 2, 2>(k, st);
  if (d->Cout <= 64) return launch<1, 4, 4, 2>(k, st);
  if (d->Cout <= 96) return launch<2, 2, 3, 4>(k, st);
  return launch<2, 2, 4, 4>(k, st);
}
