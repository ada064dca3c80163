// GroupNorm-apply folded into the A-operand load of an MFMA GEMM (gfx950, bf16 operands, fp32 accumulate).
//
//   y = epilogue( W * act( GN(x) ) )        x: raw (un-normalised) fp32 or bf16 pixel-major tensor
//
// for the pointwise / non-overlapping-patch convolutions of the encoder blocks (reference:
// src/models/simplified_attention.py:34-43 fc1 / fc2 behind Mlp.norm2 + GELU, :96-100 attn.sr / attn.k behind Block.norm1
// and attn.norm, :142-145 fc1 behind Block.norm2).  Each of these was a crd_gn_apply launch followed by a crd_conv_igemm
// launch: a full extra pass over the tensor and one more link in the encoder's latency-bound launch chain.  Here the
// GEMM's A rows take the register path instead of the LDS-DMA path: global load -> x * scale[c] + shift[c] (per sample
// and channel, from the GroupNorm sums the PRODUCER's epilogue left in `stats`) -> optional exact GELU -> bf16 ->
// ds_write_b128 into the same XOR-swizzled LDS image k_igemm's DMA builds.  The normalised tensor is still written once
// (by the column-tile 0 workgroups) when a later weight-gradient needs it.  Weights stream through an LDS-DMA ring as in
// k_igemm; the epilogue (bias, residual + DropPath scale, GroupNorm sums of the output, fp32 / bf16 stores) is shared.
#include <stdlib.h>
#include "conv_common.h"

using namespace crdk;

namespace {

constexpr int BK = 64;
typedef __attribute__((address_space(3))) void* lds_ptr;

struct GnIn {
  const void* x; int x_f32;               // raw input [B][IH*IW][x_ld] (+ channel offset applied), fp32 or bf16
  const float* stats; int gmul;           // [B][Cin/16][2] slab sums of x; a group = gmul slabs
  const float* gamma; const float* beta;  // [Cin]
  float inv_count;                        // 1 / (pixels per sample * channels per group)
  bf16_t* xn; int xn_ld; long long xn_bstride;    // optional store of act(GN(x)) (bf16), nullptr = none
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One slab's worth of a thread's A rows: AIT rows x 8 channels, raw, as loaded
template <int AIT, int XF32>
struct ARegs {
  float4 lo[AIT];
  float4 hi[XF32 ? AIT : 1];
};

template <int WM, int WN, int TM, int TN, int NSA, int NSB, int XF32, int ACT>
__global__ __launch_bounds__(256) void k_gngemm(ConvK a, GnIn gi) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  bf16_t* sA = lds;                                   // [NSA][BM][BK]
  bf16_t* sB = sA + NSA * BM * BK;                    // [NSB][BN][BK]
  float2* tab = reinterpret_cast<float2*>(sB + NSB * BN * BK);   // [Cin] (scale, shift) of this sample

  const int t = threadIdx.x, l = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int b = blockIdx.z, m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.Cout * a.Ktot * 2, 0x00020000);
  const unsigned OOB = 0x80000000u;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int r0 = 8 * wv + (l >> 3);
  const int g = (l & 7) ^ ((r0 >> 1) & 7);             // K granule this thread supplies to slot l&7 of its rows (k_igemm's swizzle)

  // ---- weight ring: issue the first slabs before anything else (they depend on nothing) ----
  unsigned woff[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int n = r0 + 32 * j, ng = n0 + n;
    woff[j] = (n < BN && ng < a.Cout) ? (unsigned)(ng * a.Ktot * 2) : OOB;
  }
  const int nK = (a.Ktot + BK - 1) / BK;
  auto stage_b = [&](int kt, int buf) {
    const int kf = kt * BK + g * 8;
    const bool kok = kf < a.Ktot;
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
      const unsigned off = kok ? woff[j] + (unsigned)(kf * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(sB + buf * BN * BK + (8 * wv + 32 * j) * BK), 16, off | (woff[j] & OOB), 0, 0, 0);
#endif
    }
  };
#pragma unroll
  for (int s = 0; s < NSB - 1; ++s) stage_b(s, s);

  // ---- A rows of this thread: source pixel of tap (0,0); patch convs (k = stride, pad 0) add the tap offset per slab ----
  long long pbase[A_IT];
  bool rok[A_IT];
  const char* xb = reinterpret_cast<const char*>(gi.x) + (long long)b * a.x_bstride * (XF32 ? 4 : 2);
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + r0 + 32 * i;
    const int oy = m / a.OW, ox = m - oy * a.OW;
    rok[i] = m < a.OHW;
    pbase[i] = ((long long)(oy * a.stride) * a.IW + ox * a.stride) * a.x_ld;
  }
  auto load_a = [&](int kt, ARegs<A_IT, XF32>& r) {
    const int kf = kt * BK + g * 8;
    int kc = kf, tap = 0;
    if (a.KW > 1) { tap = kf / a.Cin; kc = kf - tap * a.Cin; }
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const long long toff = ((long long)ky * a.IW + kx) * a.x_ld + kc;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      if (rok[i] && kf < a.Ktot) {
        if (XF32) {
          const float4* p = reinterpret_cast<const float4*>(xb + (pbase[i] + toff) * 4);
          r.lo[i] = p[0];
          r.hi[i] = p[1];
        } else {
          r.lo[i] = *reinterpret_cast<const float4*>(xb + (pbase[i] + toff) * 2);
        }
      } else {
        r.lo[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (XF32) r.hi[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  // transform + write one slab's rows into LDS stage `buf` (and to xn)
  auto commit_a = [&](int kt, int buf, const ARegs<A_IT, XF32>& r) {
    const int kf = kt * BK + g * 8;
    int kc = kf, tap = 0;
    if (a.KW > 1) { tap = kf / a.Cin; kc = kf - tap * a.Cin; }
    const bool kok = kf < a.Ktot;
    float sc[8], sh[8];
    if (kok) {
      const float4* tp = reinterpret_cast<const float4*>(tab + kc);
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float4 v = tp[j]; sc[2 * j] = v.x; sh[2 * j] = v.y; sc[2 * j + 1] = v.z; sh[2 * j + 1] = v.w; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[j] = sh[j] = 0.f;
    }
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      float v[8];
      if (XF32) {
        v[0] = r.lo[i].x; v[1] = r.lo[i].y; v[2] = r.lo[i].z; v[3] = r.lo[i].w;
        v[4] = r.hi[i].x; v[5] = r.hi[i].y; v[6] = r.hi[i].z; v[7] = r.hi[i].w;
      } else {
        const uint4 u = *reinterpret_cast<const uint4*>(&r.lo[i]);
        v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
        v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
      }
      const bool ok = rok[i] && kok;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float y = v[j] * sc[j] + sh[j];
        if (ACT == 1) y = gelu_exact(y);
        v[j] = ok ? y : 0.f;
      }
      uint4 q;
      q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]); q.z = pack_bf2(v[4], v[5]); q.w = pack_bf2(v[6], v[7]);
      *reinterpret_cast<uint4*>(sA + buf * BM * BK + (r0 + 32 * i) * BK + (l & 7) * 8) = q;
      if (gi.xn && ok && blockIdx.y == 0) {
        const int m = m0 + r0 + 32 * i;
        const int oy = m / a.OW, ox = m - oy * a.OW;
        const long long pix = (long long)(oy * a.stride + ky) * a.IW + (ox * a.stride + kx);
        *reinterpret_cast<uint4*>(gi.xn + (long long)b * gi.xn_bstride + pix * gi.xn_ld + kc) = q;
      }
    }
  };

  // A loads of the first NSA-1 slabs go out before the scale/shift table is built: the table needs the statistics
  // (a dependent load), the rows do not
  ARegs<A_IT, XF32> pre[NSA - 1];
#pragma unroll
  for (int s = 0; s < NSA - 1; ++s) load_a(s, pre[s]);

  {   // per-sample (scale, shift) of every input channel
    const float* stb = gi.stats + (long long)b * (a.Cin >> 4) * 2;
    for (int c = t; c < a.Cin; c += 256) {
      float mean, rstd;
      gn_mean_rstd(stb, ((c >> 4) / gi.gmul) * gi.gmul, gi.gmul, gi.inv_count, mean, rstd);
      const float ga = gi.gamma[c] * rstd;
      tab[c] = make_float2(ga, gi.beta[c] - mean * ga);
    }
  }

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
  __syncthreads();                                  // table visible
#pragma unroll
  for (int s = 0; s < NSA - 1; ++s) commit_a(s, s, pre[s]);

  ARegs<A_IT, XF32> nxt;
  for (int kt = 0; kt < nK; ++kt) {
    // everything issued so far has landed: weight slabs up to kt + NSB - 2 and (register path) nothing is pending
    wait_vm<0>();
    __syncthreads();                                // A slab kt written by everyone, slab kt-1's buffers free
    if (kt + NSB - 1 < nK) stage_b(kt + NSB - 1, (kt + NSB - 1) % NSB);
    const bool more = kt + NSA - 1 < nK;
    if (more) load_a(kt + NSA - 1, nxt);
    const int ca = kt % NSA, cb = kt % NSB;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[TM], bfr[TN];
      const int gi2 = ks * 2 + (l >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + (l & 31);
        af[i] = *reinterpret_cast<const bf16x8*>(&sA[ca * BM * BK + row * BK + ((gi2 ^ ((row >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + (l & 31);
        bfr[j] = *reinterpret_cast<const bf16x8*>(&sB[cb * BN * BK + row * BK + ((gi2 ^ ((row >> 1) & 7)) << 3)]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) commit_a(kt + NSA - 1, (kt + NSA - 1) % NSA, nxt);
  }
  wait_vm<0>();
  __syncthreads();

  conv_epilogue<TM, TN, WM, WN>(a, acc, b, l, wm, wn, n0, blockIdx.x, lds,
                                [&](int i, int rr, bool& valid, int& row) {
    row = m0 + (wm * TM + i) * 32 + rr;
    valid = row < a.OHW;
  }, [&](int rl, bool& valid, int& row) {
    row = m0 + rl;
    valid = row < a.OHW;
  });
}

template <int WM, int WN, int TM, int TN, int NSA, int NSB, int XF32, int ACT>
int launch_k(const ConvK& k0, const GnIn& gi, int B, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  ConvK k = k0;
  k.n_tiles = cdiv(k.OHW, BM);
  k.stats_partial = nullptr;
  size_t lds = (size_t)(NSA * BM + NSB * BN) * BK * sizeof(bf16_t) + (size_t)k.Cin * sizeof(float2);
  const size_t epi = (size_t)BM * (BN + 4) * 4 + 256 * 8 * 4 + 1024;       // fp32 staging tile + the folds behind it
  if (lds < epi) lds = epi;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gngemm<WM, WN, TM, TN, NSA, NSB, XF32, ACT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  k.lds_bytes = (int)lds;
  dim3 grid(k.n_tiles, cdiv(k.Cout, BN), B);
  hipLaunchKernelGGL((k_gngemm<WM, WN, TM, TN, NSA, NSB, XF32, ACT>), grid, dim3(256), lds, st, k, gi);
  CRD_LAUNCH_CHECK("crd_gn_conv");
  return CRD_OK;
}

template <int XF32, int ACT>
int dispatch(const ConvK& k, const GnIn& gi, int B, hipStream_t st) {
  const long long big_tiles = (long long)cdiv(k.OHW, 128) * cdiv(k.Cout, 128) * B;
  if (k.Cout <= 64 && big_tiles >= 192) return launch_k<2, 2, 2, 1, 2, 3, XF32, ACT>(k, gi, B, st);     // 128 x 64
  if (big_tiles >= 192) return launch_k<2, 2, 2, 2, 2, 2, XF32, ACT>(k, gi, B, st);                       // 128 x 128
  return launch_k<2, 2, 1, 1, 4, 4, XF32, ACT>(k, gi, B, st);                                             // 64 x 64, deep prefetch
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
  CRD_UNSUPPORTED((long long)d->Cout * d->KH * d->KW * d->Cin < (1ll << 30) && d->Cin <= 4096, "crd_gn_conv: weight tensor too large");
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
  k.red_gmul = 1; k.red_act = 0; k.red_r = nullptr; k.dbg = 0;
  GnIn gi;
  gi.x_f32 = n->x_f32;
  gi.x = n->x_f32 ? (const void*)(reinterpret_cast<const float*>(d->x) + d->x_coff) : (const void*)(reinterpret_cast<const bf16_t*>(d->x) + d->x_coff);
  CRD_CHECK_ARG((reinterpret_cast<uintptr_t>(gi.x) & 15) == 0 && (!n->x_f32 || d->x_ld % 4 == 0), "crd_gn_conv: x rows must be 16-byte aligned");
  gi.stats = n->stats; gi.gmul = n->gmul; gi.gamma = n->gamma; gi.beta = n->beta;
  gi.inv_count = 1.f / ((float)d->IH * (float)d->IW * 16.f * (float)n->gmul);
  gi.xn = reinterpret_cast<bf16_t*>(n->xn); gi.xn_ld = n->xn_ld; gi.xn_bstride = (long long)d->IH * d->IW * n->xn_ld;
  hipStream_t st = as_stream(stream);
  if (n->x_f32) return n->act ? dispatch<1, 1>(k, gi, d->B, st) : dispatch<1, 0>(k, gi, d->B, st);
  return n->act ? dispatch<0, 1>(k, gi, d->B, st) : dispatch<0, 0>(k, gi, d->B, st);
}
