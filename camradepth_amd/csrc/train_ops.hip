// Training-step kernels for gfx950: masked losses, the multi-tensor diffGradNorm optimizer and the
// table-driven weight pack / gradient unpack between the reference's parameter layout and the
// bf16 [Cout][tap][Cin] layout the MFMA kernels read.  All are HBM-bound streaming kernels.
#include <math.h>
#include "common.h"

namespace {

constexpr int TPB = 256;

// workgroup sums -> the fixed-point accumulators (order-independent: the loss value and its gradient scale reproduce)
__device__ __forceinline__ void block_atomic3(float a, float b, float c, crd_sum_t* acc) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  __shared__ float sm[3][4];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (l == 0) { sm[0][w] = a; sm[1][w] = b; sm[2][w] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stat_add(acc + 0, sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3]);
    stat_add(acc + 1, sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3]);
    stat_add(acc + 2, sm[2][0] + sm[2][1] + sm[2][2] + sm[2][3]);
  }
}

// MaskedSmoothL1Loss / MaskedMSELoss partial sums (loss_funcs.py:40-46, 83-91)
__global__ __launch_bounds__(TPB) void k_masked_l1_fwd(const float* pred, const float* target, long long n, crd_sum_t* acc) {
  float s = 0.f, cnt = 0.f, sq = 0.f;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const float t = target[i];
    if (t > 0.f) {
      const float e = pred[i] - t, ae = fabsf(e);
      s += ae < 1.f ? 0.5f * e * e : ae - 0.5f;
      sq += e * e;
      cnt += 1.f;
    }
  }
  block_atomic3(s, cnt, sq, acc);
}

// Trainer.test metrics (runner.py:443-465), per frame f: pred clipped to [0,1] and both scaled by max_depth, ground truth
// beyond max_distance dropped; acc[f] = (sum |e|, sum e^2, sum |e|/gt, count)
__global__ __launch_bounds__(TPB) void k_test_metrics(const float* pred, const float* gt, long long n, float max_depth,
                                                      float max_distance, crd_sum_t* acc) {
  const int f = blockIdx.y;
  const float* p = pred + (long long)f * n;
  const float* g = gt + (long long)f * n;
  float sa = 0.f, sq = 0.f, sr = 0.f, cnt = 0.f;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    float t = g[i] * max_depth;
    if (t > max_distance) t = 0.f;
    if (t > 0.f) {
      const float e = fminf(fmaxf(p[i], 0.f), 1.f) * max_depth - t;
      sa += fabsf(e); sq += e * e; sr += fabsf(e) / t; cnt += 1.f;
    }
  }
  sa = wave_sum(sa); sq = wave_sum(sq); sr = wave_sum(sr); cnt = wave_sum(cnt);
  __shared__ float sm[TPB / 64][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[wave][0] = sa; sm[wave][1] = sq; sm[wave][2] = sr; sm[wave][3] = cnt; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int w = 0; w < TPB / 64; ++w) v += sm[w][threadIdx.x];
    stat_add(&acc[f * 4 + threadIdx.x], v);
  }
}

// Confusion matrix of one frame for the Jaccard index of Trainer.test (runner.py:432-436): prediction = arg-max over the C
// logits of a pixel (NCHW fp32, first maximal class), confmat[f][target][pred] += 1; labels outside [0, C) are counted in
// oor[f] and skipped (torchmetrics 0.10.2 raises on them, which the reference catches: that frame's IoU stays NaN).
__global__ __launch_bounds__(TPB) void k_seg_confusion(const float* logits, const long long* labels, int C, long long HW,
                                                       unsigned long long* confmat, unsigned long long* oor) {
  extern __shared__ unsigned int hist[];      // C * C + 1
  const int f = blockIdx.y;
  for (int i = threadIdx.x; i <= C * C; i += TPB) hist[i] = 0;
  __syncthreads();
  const float* lg = logits + (long long)f * C * HW;
  const long long* lb = labels + (long long)f * HW;
  for (long long p = (long long)blockIdx.x * TPB + threadIdx.x; p < HW; p += (long long)gridDim.x * TPB) {
    const long long t = lb[p];
    if (t < 0 || t >= C) { atomicAdd(&hist[C * C], 1u); continue; }
    float best = lg[p];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = lg[(long long)c * HW + p];
      if (v > best) { best = v; arg = c; }
    }
    atomicAdd(&hist[(int)t * C + arg], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += TPB)
    if (hist[i]) atomicAdd(&confmat[(long long)f * C * C + i], (unsigned long long)hist[i]);
  if (threadIdx.x == 0 && hist[C * C]) atomicAdd(&oor[f], (unsigned long long)hist[C * C]);
}

__global__ __launch_bounds__(TPB) void k_masked_l1_bwd(const float* pred, const float* target, long long n, const crd_sum_t* acc,
                                                       const float* gout, float gmul, float* dpred) {
  const float g = gmul * (gout ? gout[0] : 1.f) / stat_get(acc + 1);
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const float t = target[i];
    float d = 0.f;
    if (t > 0.f) {
      const float e = pred[i] - t;
      d = g * fminf(fmaxf(e, -1.f), 1.f);
    }
    dpred[i] = d;
  }
}

// Cross entropy over NCHW fp32 logits, labels int64 [B][HW], ignore_index 255 (loss_funcs.py:22,27)
__global__ __launch_bounds__(TPB) void k_ce_fwd(const float* logits, const long long* labels, int C, long long HW, long long rows,
                                                crd_sum_t* acc) {
  float s = 0.f, cnt = 0.f;
  for (long long r = (long long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long long)gridDim.x * TPB) {
    const long long lab = labels[r];
    if (lab == 255) continue;
    const long long b = r / HW, p = r - b * HW;
    const float* base = logits + (b * C) * HW + p;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, base[(long long)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(base[(long long)c * HW] - mx);
    s += mx + logf(se) - base[lab * HW];
    cnt += 1.f;
  }
  block_atomic3(s, cnt, 0.f, acc);
}

// focal on the scalar mean CE: F=(1-e^-ce)^2 ce ; dF/dce = 2(1-pt)pt ce + (1-pt)^2
__global__ __launch_bounds__(TPB) void k_ce_focal_bwd(const float* logits, const long long* labels, int C, long long HW,
                                                      long long rows, const crd_sum_t* acc, const float* gout, float gmul,
                                                      float* dlogits) {
  const float cnt = stat_get(acc + 1);
  const float ce = stat_get(acc) / cnt;
  const float pt = expf(-ce);
  const float dF = 2.f * (1.f - pt) * pt * ce + (1.f - pt) * (1.f - pt);
  const float g = gmul * (gout ? gout[0] : 1.f) * dF / cnt;
  for (long long r = (long long)blockIdx.x * TPB + threadIdx.x; r < rows; r += (long long)gridDim.x * TPB) {
    const long long lab = labels[r];
    const long long b = r / HW, p = r - b * HW;
    const float* base = logits + (b * C) * HW + p;
    float* dbase = dlogits + (b * C) * HW + p;
    if (lab == 255) {
      for (int c = 0; c < C; ++c) dbase[(long long)c * HW] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, base[(long long)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(base[(long long)c * HW] - mx);
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) {
      float sm = expf(base[(long long)c * HW] - mx) * inv;
      dbase[(long long)c * HW] = g * (sm - (c == lab ? 1.f : 0.f));
    }
  }
}

// ---- diffGradNorm (src/models/diffGradNorm.py:73-110) -------------------------------------------
constexpr int OPT_CHUNK = 4096;  // elements per workgroup

// hp (device, optional): [beta1, beta2, eps, weight_decay, step_size] -- lets a captured HIP graph follow the
// per-iteration OneCycleLR schedule without re-capture
__global__ __launch_bounds__(TPB) void k_dgn_norm(const float* p, const float* g, const long long* seg_off, const int* blk2seg,
                                                  const int* blk2chunk, float wd, const float* hp, float* norm_sq,
                                                  const unsigned char* active) {
  if (hp) wd = hp[3];
  const int t = blk2seg[blockIdx.x];
  // A skipped tensor (`p.grad is None`, diffGradNorm.py:54-55) has NO gradient storage behind its segment when the gradients are used in
  // place (optim.diffGradNorm with separately allocated tensors: g = the first active gradient's pointer minus its offset): reading its
  // segment walked past the end of that allocation -- a memory access fault whenever it was the last block of an allocator segment
  // (round 6: the full GPU suite hit it once; round 1-5's k_dgn_update already skipped such tensors, this kernel did not).
  if (active && !active[t]) {
    if (threadIdx.x == 0) norm_sq[blockIdx.x] = 0.f;
    return;
  }
  const long long beg = seg_off[2 * t] + (long long)blk2chunk[blockIdx.x] * OPT_CHUNK;
  long long end = beg + OPT_CHUNK;
  if (end > seg_off[2 * t + 1]) end = seg_off[2 * t + 1];
  float s = 0.f;
  for (long long i = beg + threadIdx.x; i < end; i += TPB) {
    float gv = g[i];
    if (wd != 0.f) gv += wd * p[i];
    s += gv * gv;
  }
  s = wave_sum(s);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_sq[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];     // this workgroup's part (plain store)
}

// per tensor: ||g||^2 = its workgroups' parts added in a FIXED order (lane-strided, then the butterfly), so the e > n
// branch below sees the same norm on every run; e <- 0.95 e + 0.05 n ; factor = e > n ? e/(n+1e-8) : 1.
// One wave per tensor.  The tensor's first workgroup is found by a 64-ary search in blk2seg (non-decreasing): two or
// three dependent loads for the ~6000 workgroups of the model.
__global__ __launch_bounds__(256) void k_dgn_scalar(float* exp_grad_norm, const float* norm_part, float* factor, const unsigned char* active,
                                                    int n, const long long* seg_off, const int* blk2seg, int n_blocks) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n) return;
  if (active && !active[t]) { if (lane == 0) factor[t] = 1.f; return; }
  int lo = 0, len = n_blocks;                      // the first workgroup of tensor t lies in [lo, lo + len]
  while (len > 0) {
    const int stride = (len + 63) >> 6;
    const int pos = lo + lane * stride;
    const bool less = pos < lo + len && blk2seg[pos] < t;
    const int k = __popcll(__ballot(less));        // blk2seg is sorted: the probes below t are the first k lanes
    if (k == 0) break;
    const int end = lo + len;
    lo += (k - 1) * stride + 1;
    len = min(stride - 1, end - lo);
  }
  const int cnt = (int)((seg_off[2 * t + 1] - seg_off[2 * t] + OPT_CHUNK - 1) / OPT_CHUNK);
  float s = 0.f;
  for (int i = lane; i < cnt; i += 64) s += norm_part[lo + i];
  s = wave_sum(s);
  if (lane == 0) {
    const float nrm = sqrtf(s);
    const float e = 0.95f * exp_grad_norm[t] + 0.05f * nrm;
    factor[t] = e > nrm ? e / (nrm + 1e-8f) : 1.f;
    exp_grad_norm[t] = e;
  }
}

__global__ __launch_bounds__(TPB) void k_dgn_update(float* p, const float* g, float* m, float* v, float* pg, const float* factor,
                                                    const long long* seg_off, const int* blk2seg, const int* blk2chunk,
                                                    const unsigned char* active, float beta1, float beta2, float eps, float wd,
                                                    float step_size, const float* hp) {
  if (hp) { beta1 = hp[0]; beta2 = hp[1]; eps = hp[2]; wd = hp[3]; step_size = hp[4]; }
  const int t = blk2seg[blockIdx.x];
  if (active && !active[t]) return;
  const long long beg = seg_off[2 * t] + (long long)blk2chunk[blockIdx.x] * OPT_CHUNK;
  long long end = beg + OPT_CHUNK;
  if (end > seg_off[2 * t + 1]) end = seg_off[2 * t + 1];
  const float f = factor[t];
  for (long long i = beg + threadIdx.x; i < end; i += TPB) {
    float gv = g[i];
    const float pv = p[i];
    if (wd != 0.f) gv += wd * pv;
    const float g1 = gv * f;
    const float mv = beta1 * m[i] + (1.f - beta1) * g1;
    const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
    const float dfc = 1.f / (1.f + expf(-fabsf(pg[i] - gv)));
    m[i] = mv; v[i] = vv; pg[i] = gv;
    p[i] = pv - step_size * (mv * dfc) / (sqrtf(vv) + eps);
  }
}

// ---- weight pack / gradient unpack ------------------------------------------------------------------
// One pass over the fp32 parameters: a workgroup stages a tile of 16 output channels x CI_T internal input channels x all
// taps in LDS (source reads run along the reference layout [co][ci][tap], so they are contiguous and every parameter is
// fetched ONCE), then writes the tile in each requested packed order with the destination's fastest index on the lanes.
// (The previous kernel walked every destination form separately and gathered the source with a `taps`-float stride per
// lane: 728 MB of fetches per step for 88 MB of parameters.)
constexpr int WP_CO = 16, WP_ELEMS = 512;
__global__ __launch_bounds__(TPB) void k_weight_pack(const crd_pack_entry* tab) {
  __shared__ float tile[WP_CO * WP_ELEMS];
  const crd_pack_entry e = tab[blockIdx.y];
  int ci_t = (WP_ELEMS / e.taps) & ~7;
  if (ci_t < 8) ci_t = 8;                                   // taps <= 64: 8 channels x 64 taps = 512
  if (ci_t > 64) ci_t = 64;
  const int n_ci = (e.Cin_pad + ci_t - 1) / ci_t, n_co = (e.Cout_pad + WP_CO - 1) / WP_CO;
  const int per_co = ci_t * e.taps;                         // <= WP_ELEMS
  for (int tl = blockIdx.x; tl < n_ci * n_co; tl += gridDim.x) {
    const int co0 = (tl / n_ci) * WP_CO, ci0 = (tl % n_ci) * ci_t;
    __syncthreads();                                        // the previous tile has been written out
    for (int i = threadIdx.x; i < WP_CO * per_co; i += TPB) {
      const int co = co0 + i / per_co, r = i % per_co;
      const int ci = ci0 + r / e.taps, tap = r % e.taps;
      float v = 0.f;
      if (co < e.Cout && ci < e.Cin_pad) {
        const int cr = e.cmap ? e.cmap[ci] : (ci < e.Cin_ref ? ci : -1);
        if (cr >= 0) v = e.src[((long long)co * e.Cin_ref + cr) * e.taps + tap];
      }
      tile[i] = v;
    }
    __syncthreads();
    if (e.dst_fwd) {                                        // [co][tap][ci]: ci on the lanes
      for (int i = threadIdx.x; i < WP_CO * per_co; i += TPB) {
        const int cl = i % ci_t, r = i / ci_t;
        const int tap = r % e.taps, col = r / e.taps;
        const int co = co0 + col, ci = ci0 + cl;
        if (co < e.Cout && ci < e.Cin_pad) {
          const float v = tile[col * per_co + cl * e.taps + tap];
          const long long o = ((long long)co * e.taps + tap) * e.Cin_pad + ci;
          if (e.dst_f32) reinterpret_cast<float*>(e.dst_fwd)[o] = e.dst_f32 == 2 ? bf_round(v) : v;
          else reinterpret_cast<bf16_t*>(e.dst_fwd)[o] = f2bf(v);
        }
      }
    }
    if (e.dst_dgrad || e.dst_scatter) {                     // [ci][tap][co_pad] / [tap][ci][co_pad]: co on the lanes
      for (int i = threadIdx.x; i < WP_CO * per_co; i += TPB) {
        const int col = i % WP_CO, r = i / WP_CO;
        const int tap = r % e.taps, cl = r / e.taps;
        const int co = co0 + col, ci = ci0 + cl;
        if (co < e.Cout_pad && ci < e.Cin_pad) {
          const bf16_t q = f2bf(tile[col * per_co + cl * e.taps + tap]);      // zero beyond Cout (never loaded)
          if (e.dst_dgrad) {
            if (e.dgrad_ld == 0) reinterpret_cast<bf16_t*>(e.dst_dgrad)[((long long)ci * e.taps + tap) * e.Cout_pad + co] = q;
            else {                                             // K-concatenated form: a row range of this layer's input channels
              const int rr = ci - e.dgrad_row0;
              if (rr >= 0 && rr < e.dgrad_rows)
                reinterpret_cast<bf16_t*>(e.dst_dgrad)[((long long)rr * e.taps + tap) * e.dgrad_ld + e.dgrad_coff + co] = q;
            }
          }
          if (e.dst_scatter) reinterpret_cast<bf16_t*>(e.dst_scatter)[((long long)tap * e.Cin_pad + ci) * e.Cout_pad + co] = q;
        }
      }
    }
  }
}

// Copies of a replicated accumulator are added in index order (fixed: reproducible); fixed-point sources (src_sum) are
// added as integers and converted once.
__global__ __launch_bounds__(TPB) void k_wgrad_unpack(const crd_unpack_entry* tab, int accumulate) {
  const crd_unpack_entry e = tab[blockIdx.y];
  const long long n = (long long)e.Cout * e.taps * e.Cin_pad;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    const int ci = (int)(i % e.Cin_pad);
    const long long r = i / e.Cin_pad;
    const int tap = (int)(r % e.taps), co = (int)(r / e.taps);
    const int cr = e.cmap ? e.cmap[ci] : (ci < e.Cin_ref ? ci : -1);
    if (cr < 0) continue;
    float* d = e.dst + ((long long)co * e.Cin_ref + cr) * e.taps + tap;
    float v;
    if (e.src_sum) {
      const crd_sum_t* src = reinterpret_cast<const crd_sum_t*>(e.src);
      long long q = src[i], q1 = 0, q2 = 0, q3 = 0;          // independent chains keep several loads in flight
      int rp = 1;
      for (; rp + 3 <= e.replicas; rp += 3) {
        q1 += src[(long long)rp * e.replica_stride + i];
        q2 += src[(long long)(rp + 1) * e.replica_stride + i];
        q3 += src[(long long)(rp + 2) * e.replica_stride + i];
      }
      for (; rp < e.replicas; ++rp) q += src[(long long)rp * e.replica_stride + i];
      v = (float)(q + q1 + q2 + q3) * (1.f / GRAD_ONE);
    } else {
      const float* src = reinterpret_cast<const float*>(e.src);
      float v1 = 0.f, v2 = 0.f, v3 = 0.f;
      v = src[i];
      int rp = 1;
      for (; rp + 3 <= e.replicas; rp += 3) {
        v1 += src[(long long)rp * e.replica_stride + i];
        v2 += src[(long long)(rp + 1) * e.replica_stride + i];
        v3 += src[(long long)(rp + 2) * e.replica_stride + i];
      }
      for (; rp < e.replicas; ++rp) v += src[(long long)rp * e.replica_stride + i];
      v += (v1 + v2) + v3;
    }
    *d = accumulate ? *d + v : v;
  }
}

// out[r][c] = u(r,c) < keep[r] ? 1/keep[r] : 0 with a counter-based hash RNG; *counter advances once per launch,
// so a captured graph draws fresh masks on every replay.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(TPB) void k_dropout_masks(float* out, const float* keep, int rows, int cols, unsigned long long seed,
                                                       unsigned long long* counter) {
  const unsigned long long epoch = *counter;
  const long long total = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int r = (int)(i / cols);
    const float kp = keep[r];
    const unsigned long long h = splitmix64(splitmix64(seed ^ (epoch * 0xD1342543DE82EF95ull)) + (unsigned long long)i);
    const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    out[i] = u < kp ? 1.0f / kp : 0.0f;
  }
}
__global__ void k_counter_inc(unsigned long long* counter) { *counter += 1; }

inline int blocks_for(long long total, int cap = 2048) {
  long long n = (total + TPB - 1) / TPB;
  if (n > cap) n = cap;
  if (n < 1) n = 1;
  return (int)n;
}

}  // namespace

extern "C" int crd_masked_l1_fwd(const float* pred, const float* target, int64_t n, crd_sum_t* acc, crd_stream_t stream) {
  CRD_CHECK_ARG(pred && target && acc && n > 0, "crd_masked_l1_fwd: bad argument");
  hipLaunchKernelGGL(k_masked_l1_fwd, dim3(blocks_for(n, 512)), dim3(TPB), 0, as_stream(stream), pred, target, (long long)n, acc);
  CRD_LAUNCH_CHECK("crd_masked_l1_fwd");
  return CRD_OK;
}

extern "C" int crd_test_metrics(const float* pred, const float* gt, int32_t frames, int64_t n, float max_depth, float max_distance,
                                crd_sum_t* acc, crd_stream_t stream) {
  CRD_CHECK_ARG(pred && gt && acc && frames > 0 && n > 0, "crd_test_metrics: bad argument");
  hipLaunchKernelGGL(k_test_metrics, dim3(blocks_for(n, 64), frames), dim3(TPB), 0, as_stream(stream), pred, gt, (long long)n,
                     max_depth, max_distance, acc);
  CRD_LAUNCH_CHECK("crd_test_metrics");
  return CRD_OK;
}

extern "C" int crd_seg_confusion(const float* logits, const int64_t* labels, int32_t frames, int32_t C, int64_t HW, int64_t* confmat,
                                 int64_t* out_of_range, crd_stream_t stream) {
  CRD_CHECK_ARG(logits && labels && confmat && out_of_range && frames > 0 && C > 0 && C <= 64 && HW > 0, "crd_seg_confusion: bad argument");
  hipLaunchKernelGGL(k_seg_confusion, dim3(blocks_for(HW, 128), frames), dim3(TPB), (C * C + 1) * sizeof(unsigned int), as_stream(stream),
                     logits, reinterpret_cast<const long long*>(labels), C, (long long)HW,
                     reinterpret_cast<unsigned long long*>(confmat), reinterpret_cast<unsigned long long*>(out_of_range));
  CRD_LAUNCH_CHECK("crd_seg_confusion");
  return CRD_OK;
}

extern "C" int crd_masked_l1_bwd(const float* pred, const float* target, int64_t n, const crd_sum_t* acc, const float* gout,
                                 float gmul, float* dpred, crd_stream_t stream) {
  CRD_CHECK_ARG(pred && target && acc && dpred && n > 0, "crd_masked_l1_bwd: bad argument");
  hipLaunchKernelGGL(k_masked_l1_bwd, dim3(blocks_for(n)), dim3(TPB), 0, as_stream(stream), pred, target, (long long)n, acc, gout,
                     gmul, dpred);
  CRD_LAUNCH_CHECK("crd_masked_l1_bwd");
  return CRD_OK;
}

extern "C" int crd_ce_fwd(const float* logits, const int64_t* labels, int32_t B, int32_t C, int64_t HW, crd_sum_t* acc,
                          crd_stream_t stream) {
  CRD_CHECK_ARG(logits && labels && acc && B > 0 && C > 0 && HW > 0, "crd_ce_fwd: bad argument");
  const long long rows = (long long)B * HW;
  hipLaunchKernelGGL(k_ce_fwd, dim3(blocks_for(rows, 1024)), dim3(TPB), 0, as_stream(stream), logits,
                     reinterpret_cast<const long long*>(labels), C, (long long)HW, rows, acc);
  CRD_LAUNCH_CHECK("crd_ce_fwd");
  return CRD_OK;
}

extern "C" int crd_ce_focal_bwd(const float* logits, const int64_t* labels, int32_t B, int32_t C, int64_t HW, const crd_sum_t* acc,
                                const float* gout, float gmul, float* dlogits, crd_stream_t stream) {
  CRD_CHECK_ARG(logits && labels && acc && dlogits && B > 0 && C > 0 && HW > 0, "crd_ce_focal_bwd: bad argument");
  const long long rows = (long long)B * HW;
  hipLaunchKernelGGL(k_ce_focal_bwd, dim3(blocks_for(rows)), dim3(TPB), 0, as_stream(stream), logits,
                     reinterpret_cast<const long long*>(labels), C, (long long)HW, rows, acc, gout, gmul, dlogits);
  CRD_LAUNCH_CHECK("crd_ce_focal_bwd");
  return CRD_OK;
}

extern "C" int crd_diffgradnorm_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* prev_grad,
                                     float* exp_grad_norm, float* norm_sq, float* factor, const int64_t* seg_off,
                                     const int32_t* blk2seg, const int32_t* blk2chunk, int32_t n_tensors, int32_t n_blocks,
                                     const uint8_t* active, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int32_t step, const float* hp_dev, crd_stream_t stream) {
  CRD_CHECK_ARG(p && g && exp_avg && exp_avg_sq && prev_grad && exp_grad_norm && norm_sq && factor && seg_off && blk2seg &&
                    blk2chunk && n_tensors > 0 && n_blocks > 0 && step >= 1,
                "crd_diffgradnorm_step: bad argument");
  hipStream_t st = as_stream(stream);
  const long long* so = reinterpret_cast<const long long*>(seg_off);
  hipLaunchKernelGGL(k_dgn_norm, dim3(n_blocks), dim3(TPB), 0, st, p, g, so, blk2seg, blk2chunk, weight_decay, hp_dev, norm_sq, active);
  hipLaunchKernelGGL(k_dgn_scalar, dim3(cdiv(n_tensors, 4)), dim3(256), 0, st, exp_grad_norm, norm_sq, factor, active, n_tensors, so, blk2seg,
                     n_blocks);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * sqrt(bc2) / (bc1 + 1e-8));
  hipLaunchKernelGGL(k_dgn_update, dim3(n_blocks), dim3(TPB), 0, st, p, g, exp_avg, exp_avg_sq, prev_grad, factor, so, blk2seg,
                     blk2chunk, active, beta1, beta2, eps, weight_decay, step_size, hp_dev);
  CRD_LAUNCH_CHECK("crd_diffgradnorm_step");
  return CRD_OK;
}

extern "C" int crd_dropout_masks(float* out, const float* keep, int32_t rows, int32_t cols, uint64_t seed, uint64_t* counter,
                                 crd_stream_t stream) {
  CRD_CHECK_ARG(out && keep && counter && rows > 0 && cols > 0, "crd_dropout_masks: bad argument");
  hipLaunchKernelGGL(k_dropout_masks, dim3(blocks_for((long long)rows * cols, 64)), dim3(TPB), 0, as_stream(stream), out, keep, rows,
                     cols, (unsigned long long)seed, reinterpret_cast<unsigned long long*>(counter));
  hipLaunchKernelGGL(k_counter_inc, dim3(1), dim3(1), 0, as_stream(stream), reinterpret_cast<unsigned long long*>(counter));
  CRD_LAUNCH_CHECK("crd_dropout_masks");
  return CRD_OK;
}

extern "C" int crd_weight_pack(const crd_pack_entry* table_dev, int32_t n, int64_t max_elems, crd_stream_t stream) {
  CRD_CHECK_ARG(table_dev && n > 0 && max_elems > 0, "crd_weight_pack: bad argument");
  long long tiles = (max_elems + WP_CO * WP_ELEMS - 1) / (WP_CO * WP_ELEMS) * 2;      // padded tiles: about twice the dense count
  if (tiles < 1) tiles = 1;
  if (tiles > 96) tiles = 96;
  hipLaunchKernelGGL(k_weight_pack, dim3((unsigned)tiles, n), dim3(TPB), 0, as_stream(stream), table_dev);
  CRD_LAUNCH_CHECK("crd_weight_pack");
  return CRD_OK;
}

extern "C" int crd_wgrad_unpack(const crd_unpack_entry* table_dev, int32_t n, int64_t max_elems, int32_t accumulate,
                                crd_stream_t stream) {
  CRD_CHECK_ARG(table_dev && n > 0 && max_elems > 0, "crd_wgrad_unpack: bad argument");
  hipLaunchKernelGGL(k_wgrad_unpack, dim3(blocks_for(max_elems, 2048), n), dim3(TPB), 0, as_stream(stream), table_dev, accumulate);
  CRD_LAUNCH_CHECK("crd_wgrad_unpack");
  return CRD_OK;
}
