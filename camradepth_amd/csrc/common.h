// Shared device/host helpers for libcamradepth_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/camradepth_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CRD_WAVE 64
#define GN_EPS 1e-5f

void crd_set_error(const char* fmt, ...);

#define CRD_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      crd_set_error(__VA_ARGS__);           \
      return CRD_E_INVALID;                 \
    }                                       \
  } while (0)

#define CRD_UNSUPPORTED(cond, ...)          \
  do {                                      \
    if (!(cond)) {                          \
      crd_set_error(__VA_ARGS__);           \
      return CRD_E_UNSUPPORTED;             \
    }                                       \
  } while (0)

// Dynamic-LDS reservations (hipFuncSetAttribute) are checked: a refusal is remembered (thread-local, api.hip) and reported by the
// CRD_LAUNCH_CHECK that follows the launch, by kernel name and size, instead of surfacing as an opaque "invalid argument" launch failure.
void crd_note_attr_failure(const char* kernel, int bytes, int hip_err);
int crd_report_attr_failure(const char* entry_point);      // 1 if a refusal was pending: crd_last_error() is set
// Tuning knobs whose verdict is recorded in DESIGN.md are CONSTANTS in the product build; a developer build (-DCRD_DEV_SWITCHES, e.g.
// CRD_EXTRA_FLAGS=-DCRD_DEV_SWITCHES python -m camradepth_amd.build) reads them from the environment again, for tools/sweep_knobs.sh
// and the ablation scripts under tools/.
#ifdef CRD_DEV_SWITCHES
#include <stdlib.h>
static inline int crd_dev_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int crd_dev_int(const char*, int dflt) { return dflt; }
#endif

static inline void crd_reserve_lds(const void* fn, int bytes, const char* kernel) {
  const hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e_ != hipSuccess) crd_note_attr_failure(kernel, bytes, (int)e_);
}

#define CRD_LAUNCH_CHECK(name)                                                   \
  do {                                                                           \
    if (crd_report_attr_failure(name)) { (void)hipGetLastError(); return CRD_E_LAUNCH; } \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      crd_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
      return CRD_E_LAUNCH;                                                       \
    }                                                                            \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN not special-cased: the hot path never produces it on purpose)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf_round(float f) { return bf2f(f2bf(f)); }

// An LDS-DMA request (16 bytes per lane, lane-linear at LDS byte address lds_addr) the compiler does not know about: for
// kernels whose LDS reads are ds_read_tr builtins, in front of which the compiler would otherwise wait for every request in
// flight.  rsrc = {base lo, base hi (48-bit address), bytes, 0x00020000}; m0 takes the LDS address (one wait state before
// the DMA reads it).  Completion is the caller's counted s_waitcnt vmcnt + barrier, as with the builtin.
typedef __attribute__((ext_vector_type(4))) int crd_rsrc_t;
__device__ __forceinline__ crd_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  crd_rsrc_t r = {(int)(unsigned)p, (int)(unsigned)((p >> 32) & 0xffffu), (int)bytes, 0x00020000};
  r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
  r[2] = __builtin_amdgcn_readfirstlane(r[2]);
  return r;
}
__device__ __forceinline__ void lds_dma16(const crd_rsrc_t& rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rsrc) : "memory", "m0");
}

// Workgroup barrier for kernels that keep LDS-DMA requests (buffer_load ... lds) in flight across it.  __syncthreads()
// is a workgroup-scope fence, which the compiler implements as s_waitcnt vmcnt(0) lgkmcnt(0) before s_barrier: every
// request of an N-stage ring is drained at every step and the ring degenerates to one stage (found in the ISA of k_igemm:
// the kernel's own counted s_waitcnt vmcnt(N) was followed by a vmcnt(0)).  This one waits for the wave's LDS accesses only
// (the builtin, so that the compiler's own lgkmcnt bookkeeping sees it); what has to have landed is the caller's business
// (counted s_waitcnt vmcnt).  The "memory" clobber keeps the compiler from moving LDS accesses across the barrier.
// Two related traps: an LDS read through a HIP vector STRUCT (uint4, float4) or the ds_read_tr builtins makes the compiler
// put s_waitcnt vmcnt(0) in front of it when LDS-DMA is in flight; reads through ext_vector_type pointers do not.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0); vmcnt / expcnt untouched
  asm volatile("s_barrier" ::: "memory");
}
// two fp32 -> packed bf16 pair with the hardware converter (v_cvt_pk_bf16_f32, round-to-nearest-even)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<uint32_t*>(&r);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// Exact (erf) GELU of nn.GELU() -- the libm erff costs ~40 VALU ops and made the GroupNorm+GELU passes VALU-bound.
// erf via Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level): one v_rcp, one v_exp, five FMAs;
// the same exponential e^{-x^2/2} also gives the Gaussian density needed by the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& Phi, float& E) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  E = __expf(-z * z);                                   // = exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32 (1 ulp); __frcp_rn expands to a full IEEE divide
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * E;
  Phi = 0.5f * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float gelu_exact(float x) {
  float Phi, E;
  gelu_parts(x, Phi, E);
  return x * Phi;
}
__device__ __forceinline__ float gelu_grad(float x) {
  float Phi, E;
  gelu_parts(x, Phi, E);
  return Phi + x * 0.39894228040143267794f * E;
}
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// load 8 consecutive channels as floats from a bf16 or fp32 row
__device__ __forceinline__ void load8(const void* base, int64_t elem_off, int is_f32, float (&v)[8]) {
  if (is_f32) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
    float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(base) + elem_off);
    v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
    v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
  }
}
template <int F32>
__device__ __forceinline__ void load8t(const void* base, long long elem_off, float (&v)[8]) {
  if (F32) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
    float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(base) + elem_off);
    v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
    v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
  }
}
__device__ __forceinline__ void store8_bf16(void* base, int64_t elem_off, const float (&v)[8]) {
  uint4 u;
  u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
  *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(base) + elem_off) = u;
}
// OCP e4m3 (gfx950's fp8): saturating conversion, round to nearest even
constexpr float E4M3_MAX = 448.f;
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -E4M3_MAX), E4M3_MAX); b = fminf(fmaxf(b, -E4M3_MAX), E4M3_MAX);
  c = fminf(fmaxf(c, -E4M3_MAX), E4M3_MAX); d = fminf(fmaxf(d, -E4M3_MAX), E4M3_MAX);
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);       // bytes 0, 1
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);            // bytes 2, 3
  return (unsigned)p;
}
// the fp8 copy of a bf16 tensor: e4m3(bf16(v) * inv_scale) -- rounded to bf16 first, so that it IS the quantised bf16 value
// whichever kernel (crd_quant_fp8 on the stored tensor, or a producer's fused fp8 output) wrote it
__device__ __forceinline__ void store8_fp8(void* base, int64_t elem_off, const float (&v)[8], float inv_scale) {
  // (the bf16 rounding through the hardware converter, two values per instruction: as bf_round per value it was ~5 integer operations each)
  const uint32_t w0 = pack_bf2(v[0], v[1]), w1 = pack_bf2(v[2], v[3]), w2 = pack_bf2(v[4], v[5]), w3 = pack_bf2(v[6], v[7]);
  uint2 q;
  q.x = pack_fp8x4(bf_lo(w0) * inv_scale, bf_hi(w0) * inv_scale, bf_lo(w1) * inv_scale, bf_hi(w1) * inv_scale);
  q.y = pack_fp8x4(bf_lo(w2) * inv_scale, bf_hi(w2) * inv_scale, bf_lo(w3) * inv_scale, bf_hi(w3) * inv_scale);
  *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(base) + elem_off) = q;
}
__device__ __forceinline__ void store8_f32(float* base, int64_t elem_off, const float (&v)[8]) {
  float4* p = reinterpret_cast<float4*>(base + elem_off);
  p[0] = make_float4(v[0], v[1], v[2], v[3]);
  p[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// ---- order-independent sums (include/camradepth_hip.h: crd_sum_t) -------------------------------------------------------
// A partial is rounded once to the fixed-point grid (the power-of-two scaling is exact in fp32) and added with a 64-bit
// integer atomic; the total does not depend on the order of arrival.
constexpr float STAT_ONE = (float)(1 << CRD_STAT_FRAC_BITS);
constexpr float GRAD_ONE = (float)(1ll << CRD_GRAD_FRAC_BITS);
typedef __attribute__((address_space(1))) unsigned long long gsum_raw_t;
// float -> fixed point, round to nearest even.  The general conversion (__float2ll_rn) expands to ~20 VALU operations; while
// |v * one| < 2^50 the fp64 adder does the rounding: x + 1.5 * 2^52 has an ulp of exactly 1, so its mantissa bits are the
// integer (the low dword of the magic constant's bit pattern is zero: taking it off again is one 32-bit subtract).  Both
// paths round the exact product v * one to nearest even, so they agree bit for bit and the choice is invisible.
// NON-FINITE / OUT-OF-RANGE partials.  An integer accumulator cannot hold NaN or infinity: a NaN partial would convert to 0 and an
// infinite one would saturate and wrap -- a diverged run would show a finite, too-small loss where the reference shows NaN
// (ADVICE r3).  Such a partial (NaN, +-inf, or |v * one| >= 2^62) adds NOTHING and raises this translation unit's sticky flag
// instead; crd_nonfinite_status() (api.hip) reads and clears the flags of all translation units, and the Python side turns the
// affected values into NaN (TrainStep.losses(), the loss modules).  The flag is raised on the rare path only (one plain store).
static __device__ int crd_tu_nonfinite;
__device__ __forceinline__ long long to_fx(float v, float one) {
  const float s = v * one;                              // exact (power-of-two scale) unless it overflows to inf
  if (fabsf(s) < 1125899906842624.f) {                  // 2^50
    const double d = (double)s + 6755399441055744.0;    // 1.5 * 2^52
    return __double_as_longlong(d) - 0x4338000000000000ll;
  }
  if (!(fabsf(s) < 4611686018427387904.f)) {            // 2^62; also NaN (every comparison with NaN is false)
    crd_tu_nonfinite = 1;
    return 0;
  }
  return __float2ll_rn(s);
}
__device__ __forceinline__ void fx_add(crd_sum_t* p, long long q) {
  // address space 1 stated explicitly: through a descriptor loaded from memory the compiler would emit a FLAT atomic
  __hip_atomic_fetch_add((gsum_raw_t*)p, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stat_add(crd_sum_t* p, float v) { fx_add(p, to_fx(v, STAT_ONE)); }
__device__ __forceinline__ void grad_add(crd_sum_t* p, float v) { fx_add(p, to_fx(v, GRAD_ONE)); }
__device__ __forceinline__ float stat_get(const crd_sum_t* p) { return (float)(*p) * (1.f / STAT_ONE); }
__device__ __forceinline__ float grad_get(const crd_sum_t* p) { return (float)(*p) * (1.f / GRAD_ONE); }

// mean / rstd of GroupNorm group from g16 slab sums: group = gmul consecutive slabs starting at slab0
// mean and 1/sqrt(var + eps) of `count` elements from their fixed-point sums.  E[x^2] - mean^2 in fp64: in fp32 the difference
// loses mean^2 / var of its 24 bits (groups whose mean is 30x their deviation kept 14), fp64 keeps what the integer sums hold.
// count (pixels x channels of the group) is exact in a float up to 2^28 elements (it is a multiple of 16).
__device__ __forceinline__ void gn_moments(long long s, long long ss, float count, float& mean, float& rstd) {
  const double inv = 1.0 / ((double)count * (double)STAT_ONE);
  const double m = (double)s * inv;
  const double var = fmax((double)ss * inv - m * m, 0.0);
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(var + (double)GN_EPS));
}

__device__ __forceinline__ void gn_mean_rstd(const crd_sum_t* stats_b, int slab0, int gmul, float count, float& mean, float& rstd) {
  typedef __attribute__((ext_vector_type(2))) long long ll2;
  long long s = 0, ss = 0;
  if (gmul == 1) {
    const ll2 v = *reinterpret_cast<const ll2*>(stats_b + slab0 * 2);
    s = v[0]; ss = v[1];
  } else {
    // the slabs of the group, four 16-byte loads in flight per pass (clamped index + select): as `for (i < gmul) s += stats[...]`
    // every load was waited for before the next was issued -- up to eight dependent memory latencies at the head of every kernel
    // that normalises with 128-channel groups (Mlp.norm2 of stages 1-2; four at stages 3-4)
    for (int i0 = 0; i0 < gmul; i0 += 4) {          // (four per pass: eight cost 18-27 more registers in the streaming kernels)
      ll2 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const ll2*>(stats_b + (slab0 + (i0 + k < gmul ? i0 + k : 0)) * 2);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = i0 + k < gmul;
        s += ok ? v[k][0] : 0ll;
        ss += ok ? v[k][1] : 0ll;
      }
    }
  }
  gn_moments(s, ss, count, mean, rstd);
}

// sum over the B samples of the per-sample (sum g, sum g*xhat) pairs of channel c in r[B][C][2]: eight independent 16-byte loads in
// flight per pass (clamped index + select, no branch) -- as `for (bb) g += r[bb]` the compiler waited for every load before issuing
// the next (s_waitcnt vmcnt(0) in the loop): B dependent memory latencies at the head of the sample-0 workgroups of every
// GroupNorm-backward apply launch.
__device__ __forceinline__ void sum_samples(const crd_sum_t* r, int B, int C, int c, long long& g0, long long& g1) {
  typedef __attribute__((ext_vector_type(2))) long long ll2;
  g0 = 0; g1 = 0;
  for (int b0 = 0; b0 < B; b0 += 8) {
    ll2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bb = b0 + k < B ? b0 + k : B - 1;
      v[k] = *reinterpret_cast<const ll2*>(r + ((long long)bb * C + c) * 2);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = b0 + k < B;
      g0 += ok ? v[k][0] : 0ll;
      g1 += ok ? v[k][1] : 0ll;
    }
  }
}

// every translation unit registers the DEVICE ADDRESS of its sticky flag with api.hip: crd_nonfinite_status() gathers all of them with
// one 64-thread launch and one copy (round 5, ADVICE r4: it used to be one blocking hipMemcpyFromSymbol per translation unit and call)
void crd_register_nonfinite_flag(void* (*addr_of_flag)());
static void* crd_tu_nonfinite_addr() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(crd_tu_nonfinite)) != hipSuccess) return nullptr;
  return p;
}
namespace { struct CrdNonfiniteRegistrar { CrdNonfiniteRegistrar() { crd_register_nonfinite_flag(&crd_tu_nonfinite_addr); } }; static CrdNonfiniteRegistrar crd_nonfinite_registrar; }

static inline hipStream_t as_stream(crd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
