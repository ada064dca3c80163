/*
 * camradepth_hip.h -- C ABI of libcamradepth_hip.so (gfx950 / MI355X).
 *
 * The reference (TUMFTM/CamRaDepth) has no FFI or operator registry: its hot path is PyTorch ATen
 * calls made from Python modules.  This header is therefore the boundary a maintainer would bind
 * with ctypes to replace those calls.  Every entry point cites the reference code it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless named host_*.  The library never allocates, frees or
 *     synchronises; the caller owns every buffer and passes the HIP stream to enqueue on.
 *   - Return value: 0 = ok, negative = error (CRD_E_*); crd_last_error() gives a message
 *     (thread-local).  No exceptions cross the ABI.
 *   - Activations are "pixel-major" (NHWC): element (b,y,x,c) of a tensor lives at
 *     base[((b*H + y)*W + x)*ld + coff + c]; `ld` is the channel stride of the underlying buffer,
 *     so a producer can write into a channel slice of a wider buffer (this is how torch.cat of the
 *     reference disappears).  bf16 tensors need ld, coff and C to be multiples of 8 (16-byte rows).
 *   - The encoder's [B,C,N] tensors of the reference are the same layout with H*W = N.
 */
#ifndef CAMRADEPTH_HIP_H
#define CAMRADEPTH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* crd_stream_t; /* hipStream_t */

#define CRD_OK 0
#define CRD_E_INVALID (-1)     /* bad argument (null pointer, misaligned stride, ...) */
#define CRD_E_UNSUPPORTED (-2) /* shape outside what the kernels implement */
#define CRD_E_LAUNCH (-3)      /* HIP launch failure */

/* Order-independent sums.  Every accumulator that several workgroups add into -- GroupNorm statistics, per-channel sums,
 * loss / metric sums, the reduce sums of the GroupNorm backward, weight- and bias-gradient accumulators -- is a 64-bit
 * FIXED-POINT integer added to with integer atomics: integer addition is associative, so the result does not depend on
 * the order in which the workgroups arrive and two runs on the same inputs are bit-identical (fp32 atomics are not:
 * VERDICT r2 measured 1-3 % run-to-run spread downstream of them).  value = acc * 2^-FRAC_BITS.
 *   CRD_STAT_FRAC_BITS (20): forward statistics and loss sums -- range +-8.8e12, every partial rounded to 2^-21
 *     (<= 1e-9 absolute on a mean / mean square of >= 64 elements, against GroupNorm's eps = 1e-5);
 *   CRD_GRAD_FRAC_BITS (44): gradient-magnitude sums -- range +-5.2e5, partials rounded to 2^-45 (2.8e-14; operands are
 *     bf16, i.e. 4e-3 relative, and diffGradNorm's eps = 1e-8 swamps gradients below 1e-9).
 * Buffers of this type are zeroed by the caller (all-zero bits = 0.0). */
typedef int64_t crd_sum_t;
#define CRD_STAT_FRAC_BITS 20
#define CRD_GRAD_FRAC_BITS 44
/* NaN, infinity and magnitudes outside the range cannot be represented in a crd_sum_t.  A partial sum that is not finite, or
 * whose scaled magnitude reaches 2^62, adds NOTHING to its accumulator and raises a sticky device-side flag instead (a run
 * that has diverged would otherwise report a finite, too-small loss where the reference reports NaN, and bias / GroupNorm
 * gradients of 0 next to NaN weight gradients).  crd_nonfinite_status(reset, stream) runs its query ON `stream` -- behind every
 * kernel the caller enqueued there -- and waits for that stream; returns 1 if any such partial was dropped since the flag was last
 * cleared (0 if none, negative on error) and clears it when reset != 0.  reset == 2: clear only -- enqueued on `stream`, nothing is read
 * back, no wait, returns 0 (the eager model forward opens a step with it).  (Work on OTHER streams is ordered by the caller.)
 * TrainStep.losses(), the loss modules and camradepth_amd.lib.stat_checked() report NaN when it is set.  Sums whose TOTAL
 * leaves the range while every partial stays inside it still wrap (|statistic sum| > 8.8e12, |gradient sum| > 5.2e5). */
int crd_nonfinite_status(int32_t reset, crd_stream_t stream);

const char* crd_last_error(void);
#define CRD_ABI_VERSION 6        /* bumped whenever a struct layout or signature changes: the binding refuses a stale library */
int crd_version(void);          /* CRD_ABI_VERSION of the library that was built */
const char* crd_arch(void);     /* "gfx950" */

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (bf16 in, fp32 accumulate).
 * Replaces torch conv1d/conv2d on the hot path: OverlapPatchEmbed.proj
 * (src/models/simplified_attention.py:158,185), Attention_MaxPool.{q,k,proj,sr} (:59-68,92-102),
 * Mlp.{fc1,fc2} (:17,20,35,41), ConvLayer conv (src/utils/utils.py:211,225), Depth_Activation
 * convs (utils.py:282-289) and the seg head convs (src/models/CamRaDepth.py:88-94), and -- with
 * gather_mode 1 / out_mode 1 -- their data gradients (autograd of the same calls).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x;      /* bf16 activations, pixel-major */
  int32_t x_ld, x_coff;
  int32_t B, IH, IW, Cin;       /* Cin multiple of 8 (pad channels must carry zero weights) */
  const void* w;      /* bf16 weights [Cout][KH*KW][Cin], tap index = ky*KW + kx */
  int32_t Cout, KH, KW, stride, pad;
  int32_t OH, OW;               /* output grid */
  int32_t gather_mode;          /* 0: y[oy,ox] += w[ky,kx] x[oy*s-p+ky, ox*s-p+kx]   (forward)
                                   1: y[iy,ix] += w[ky,kx] x[(iy+p-ky)/s, (ix+p-kx)/s] (data gradient) */
  void* y;            /* output, pixel-major */
  int32_t y_ld, y_coff;
  int32_t y_f32;                /* 0: bf16 output, 1: fp32 output */
  int32_t out_mode;             /* 0: y[b,oy,ox,n]; 1: patch scatter: n=(ky*pk+kx)*pc+c ->
                                   y[b, oy*pk+ky, ox*pk+kx, c] (data gradient of a k=s conv) */
  int32_t patch_k, patch_c;
  const float* bias;  /* [Cout] or NULL; element (b, co) is bias[b*bias_bstride + co] */
  int32_t bias_bstride;         /* 0: shared by all images; >0: one bias row per image */
  int32_t act;                  /* 0 none, 1 sigmoid */
  const float* res;   /* fp32, same indexing as y (ld = res_ld): y = res + res_scale[b]*v, or NULL */
  int32_t res_ld;
  const float* res_scale; /* [B] or NULL (=1) */
  int32_t accumulate;           /* 1: y += v (read-modify-write in y's dtype) */
  crd_sum_t* stats;   /* [B][Cout/16][2] += (sum, sum of squares) of the ROUNDED outputs (CRD_STAT_FRAC_BITS), or NULL */
  float* stats_partial;           /* optional scratch: per-tile partial sums are written here with plain stores and
                                     folded into `stats` by a small second kernel (avoids contended atomics) */
  int64_t stats_partial_capacity; /* floats available in stats_partial; needs B*ceil(OH*OW/64)*(Cout/16)*2 */
  /* Optional (red_x != NULL; bf16 output, no residual, Cout > 160, not the 3x3 halo shapes): this launch produces the dy
   * of a GroupNorm (+GELU) backward -- e.g. the data gradient of Mlp.fc2 feeding Mlp.norm2 -- and also computes that
   * backward's reduce phase: red_x is the GroupNorm's raw input (bf16, or fp32 with red_x_f32) laid out like y (pixel-major, red_x_ld channels per
   * pixel, same pixel grid), red_stats its g16 sums, red_gmul its slabs per group, red_act 1 for GELU; red_r is the
   * buffer crd_gn_bwd_reduce would fill (crd_sum_t [B*Cout*2 + B*(Cout/(16*red_gmul))*2], zeroed by the caller).  Only
   * crd_gn_bwd_apply remains to be called. */
  const void* red_x;
  int32_t red_x_ld, red_gmul, red_act, red_x_f32;   /* red_x_f32 = 1: red_x holds fp32 (the encoder's residual stream) */
  const crd_sum_t* red_stats;
  const float* red_gamma;
  const float* red_beta;
  crd_sum_t* red_r;            /* CRD_GRAD_FRAC_BITS */
  crd_sum_t* chan_sums;               /* optional, with stats and an fp32 or residual output: per-channel (sum, sumsq) of the
                                     stored output, crd_sum_t [B][Cout][2] (CRD_STAT_FRAC_BITS) (what crd_gn_stats'
                                     chan_sums would hold for the output tensor) */
} crd_conv_desc;

int crd_conv_igemm(const crd_conv_desc* d, crd_stream_t stream);

/* The same convolution with a GroupNorm (+ exact GELU) applied to its INPUT while the A operand is loaded:
 *   y = epilogue( W * act( GroupNorm(x) ) ),   GroupNorm(x)[b,p,c] = (x - mean[b,g]) * rstd[b,g] * gamma[c] + beta[c]
 * -- nn.GroupNorm followed by Conv1d / Conv2d in Block / Attention_MaxPool / Mlp (simplified_attention.py:34-43 Mlp.norm2 +
 * GELU + fc2, :96-100 attn.sr behind Block.norm1 and attn.k behind attn.norm, :142-145 fc1 behind Block.norm2) without
 * the pass that would store the normalised tensor first.  d->x is the RAW tensor (fp32 when x_f32, else bf16); statistics
 * as raw sums per 16-channel slab, crd_sum_t [B][Cin/16][2] (what crd_gn_stats / a producer's `stats` epilogue leave), a
 * group being `gmul` consecutive slabs.  xn != NULL: act(GroupNorm(x)) is also stored as bf16 [B][IH*IW][xn_ld] (the
 * operand a later weight-gradient call needs).  Pointwise (1x1) and non-overlapping patch convolutions only
 * (KH == KW == stride, pad 0, gather_mode 0, out_mode 0); every epilogue option of crd_conv_desc except red_x and
 * stats_partial. */
typedef struct {
  int32_t x_f32;            /* d->x holds fp32 (1) or bf16 (0) */
  int32_t gmul;             /* 16-channel slabs per group */
  const crd_sum_t* stats;   /* [B][Cin/16][2] */
  const float* gamma;       /* [Cin] */
  const float* beta;        /* [Cin] */
  int32_t act;              /* 0 none, 1 exact GELU (after the affine) */
  int32_t xn_ld;
  void* xn;                 /* optional bf16 output */
} crd_gn_input;
int crd_gn_conv(const crd_conv_desc* d, const crd_gn_input* n, crd_stream_t stream);
/* Two crd_gn_conv problems of the same batch in ONE launch (round 6): attn.q and the attn.sr patch convolution of a Block, which both
 * read Block.norm1(x) (src/models/simplified_attention.py:96-100).  Both read the fp32 residual stream behind a GroupNorm without
 * activation and both take the 64 x 64 tiles; every option of crd_gn_conv applies to each (bias, output sums, n->xn). */
int crd_gn_conv2(const crd_conv_desc* d0, const crd_gn_input* n0, const crd_conv_desc* d1, const crd_gn_input* n1, crd_stream_t stream);
/* Backward twin of crd_gn_conv (round 6): the APPLY phase of a GroupNorm (+ exact GELU) backward folded into the A-operand load of
 * the pointwise data-gradient GEMM that consumes the gradient (csrc/xfgemm.hip).  d->x holds dy of the GroupNorm (bf16
 * [B][P][x_ld] + x_coff, Cin = the GroupNorm's channels); the GEMM multiplies d->w with
 *   dx = (gamma * g - S1 - xhat * S2) * rstd,  g = dy * mask * act'(xhat * gamma + beta)
 * -- what crd_gn_bwd_apply(x = gx, dy, stats, r) stores -- and the descriptor's epilogue options apply to the product (bias,
 * accumulate, out_mode = 1 patch scatter, stats, the fused reduce red_*).  Replaces crd_gn_bwd_apply + crd_conv_igemm in the backward
 * chain of an encoder Block: Mlp.norm1 in front of fc1's data gradient, attn.norm in front of the sr patch scatter (the autograd of
 * src/models/simplified_attention.py:38-41,96-100).  KH = KW = stride = 1, pad 0; bf16 output. */
typedef struct {
  const void* gx; int32_t gx_f32, gx_ld;   /* the GroupNorm's INPUT [B][P][gx_ld] from channel 0, bf16 or fp32 */
  int32_t gmul, act;                       /* group = gmul 16-channel slabs; act = 1: exact GELU behind the GroupNorm */
  const crd_sum_t* stats;                  /* forward slab sums of gx [B][Cin/16][2] */
  const float* gamma; const float* beta;   /* [Cin] */
  const float* mask;                       /* Dropout2d mask [B][Cin] or NULL */
  const crd_sum_t* r;                      /* crd_gn_bwd_reduce's sums: [B][Cin][2] then [B][Cin/(16 gmul)][2] */
  void* dx; int32_t dx_ld;                 /* optional: dx stored as bf16 [B][P][dx_ld] (what the weight gradients read), or NULL */
  float* dgamma; float* dbeta;             /* optional (both or neither): += the GroupNorm's parameter gradients */
} crd_gn_bwd_input;
int crd_gn_bwd_conv(const crd_conv_desc* d, const crd_gn_bwd_input* n, crd_stream_t stream);
/* 1 when crd_conv_igemm / crd_gn_conv send a 1x1 layer of this shape (input channels, output channels, pixels per sample) to the
 * narrow streaming kernel (round 5, csrc/pw_narrow.hip: Mlp.fc2 and the data gradient of Mlp.fc1 at encoder stages 1-2,
 * simplified_attention.py:17,20,35,41 -- hidden 512 / 1024 -> 64 / 128 channels): the weights of a workgroup's 16-column
 * slices live in registers, the hidden tensor streams through LDS once.  With crd_gn_conv (act = 1) Mlp.norm2 + the exact GELU
 * are applied to the rows on their way in and crd_gn_input.xn receives the activated tensor: callers use this to decide whether
 * the separate crd_gn_apply pass over the hidden tensor can be dropped. */
int crd_pw_narrow_supported(int32_t Cin, int32_t Cout, int32_t pixels);
/* Tuning / test knob of that kernel: on = 0 / 1 switches it off / on (2: on, and also for the K = 1024 launches without a GroupNorm
 * in front, which the generic tiles serve faster -- benchmarks only) for launches recorded from now on (returns the previous
 * setting; plans built before keep what they chose); on < 0: returns the number of launches that took the narrow kernel so far
 * (modulo 2^31) -- the tests use it to assert that the native kernel ran and not the generic tiles behind it. */
int crd_tune_pw_narrow(int32_t on);
/* Tuning knob of the 3x3 halo kernel: launches whose 128-column tiling would give fewer than `workgroups` workgroups use
 * 64- or 32-column tiles instead (default 512; 0 disables; < 0 restores the default).  Returns the previous value.
 * Not needed for correctness -- the tests use it to reach every tile configuration with small inputs. */
int crd_tune_conv3x3_small_grid(int workgroups);
/* Round 5: the 64 x 64-tile launches of crd_conv_igemm with a bf16 output in the plain layout (+ bias, sigmoid, accumulate, GroupNorm
 * sums, the fused GroupNorm-backward reduce) use a REGISTER epilogue (csrc/conv_common.h: conv_epilogue_reg -- swapped MFMA operands,
 * 16-byte stores straight from the accumulators, the epilogue's inputs prefetched at kernel start) instead of the LDS-staged one.
 * on = 0 / 1 switches it off / on for the following launches (returns the previous setting); on < 0 returns the number of launches
 * that took it so far (modulo 2^31).  Same results up to the summation order of the fused reduce's fp32 partials. */
int crd_tune_igemm_reg_epilogue(int32_t on);

/* ---- fp8 (OCP e4m3) inference path of the decoder's 3x3 ConvLayers (BASELINE.json config 5) ------------------------------
 * y = bf16( x_scale * w_scales[co] * sum_k x8[k] * w8[co][k] ) on the block-scaled MFMA (all block scales 2^0) at twice the
 * bf16 matrix rate.  d as for crd_conv_igemm with x = fp8 activations [B][IH*IW][x_ld] (1 byte per channel, Cin / x_ld /
 * x_coff multiples of 16) and w = fp8 weights [Cout][9][Cin]; 3x3 / stride 1 / pad 1 forward, plain bf16 output, optional
 * GroupNorm sums (stats + a stats_partial buffer of B x ceil(W/32) x ceil(H/16) x 4 x Cout/16 x 2 floats).  No reference
 * counterpart: the reference computes these convolutions under fp16 autocast (runner.py:191). */
int crd_conv3x3_fp8(const crd_conv_desc* d, const float* w_scales, float x_scale, crd_stream_t stream);
/* amax[0] = max(amax[0], max |x|) over channels [coff, coff + C) of a pixel-major bf16 tensor (atomic max on the float's
 * bits: amax must hold a non-negative float, e.g. 0).  Calibration of x_scale = amax / 448. */
int crd_amax_bf16(const void* x, int64_t rows, int32_t ld, int32_t coff, int32_t C, float* amax, crd_stream_t stream);
/* y[r][y_coff + c] = e4m3(clamp(x[r][coff + c] / scale, +-448)), round to nearest even; C, y_ld, y_coff multiples of 8 */
int crd_quant_fp8(const void* x, int64_t rows, int32_t ld, int32_t coff, int32_t C, void* y, int32_t y_ld, int32_t y_coff,
                  float scale, crd_stream_t stream);
/* Producers that write the e4m3 copy directly (what crd_quant_fp8 would make of their bf16 output: e4m3(bf16(v) / y_scale)):
 * GroupNorm(+GELU) of a ConvLayer (utils.py:210-228) and the decoder's 2x bicubic upsample (utils.py:249-257).  Arguments as
 * crd_gn_apply / crd_bicubic2x with y an fp8 tensor (y_ld, y_coff in bytes = channels).  y_bf16 (optional, NULL = none):
 * the bf16 output itself as well -- a training step keeps it for the backward pass. */
int crd_gn_apply_fp8(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                     int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask, void* y_fp8, int32_t y_ld,
                     int32_t y_coff, float y_scale, void* y_bf16, int32_t yb_ld, int32_t yb_coff, crd_stream_t stream);
int crd_bicubic2x_fp8(const void* x, int32_t x_ld, int32_t x_coff, int32_t B, int32_t H, int32_t W, int32_t C, void* y_fp8,
                      int32_t y_ld, int32_t y_coff, float y_scale, void* y_bf16, int32_t yb_ld, int32_t yb_coff, crd_stream_t stream);
/* per output channel: scales[co] = max |w[co]| / 448 (1 for an all-zero row), w_fp8[co][tap][c] = e4m3(w[co][tap][c] / scales[co])
 * for c < Cin and 0 for Cin <= c < Cin_out (crd_conv3x3_fp8 wants Cin_out % 16 == 0); w_bf16 is the packed forward weight
 * [Cout][taps][Cin] */
int crd_weight_quant_fp8(const void* w_bf16, int32_t Cout, int32_t taps, int32_t Cin, int32_t Cin_out, void* w_fp8, float* scales,
                         crd_stream_t stream);

/* ---- fp8 DATA GRADIENTS of the same ConvLayers (round 5; config 5 as a training step) ----------------------------------------
 * dx[b,iy,ix,ci] (+)= bf16( *x_scale_dev * w_scales[ci] * sum_{ky,kx,co} dy8[b, iy+1-ky, ix+1-kx, co] * w8[ci][ky*3+kx][co] ):
 * the autograd of crd_conv3x3_fp8's convolution w.r.t. its input (utils.py:211 under autograd), with dy = the GroupNorm
 * backward's output quantised to e4m3 (crd_gn_bwd_apply_fp8) and w8 = the packed data-gradient weights [Cin_pad][9][Cout_pad]
 * quantised per INPUT channel (crd_weight_quant_fp8 on crd_pack_entry.dst_dgrad).  d: gather_mode 1, x = dy8 (fp8, Cin here =
 * the layer's output channels rounded up to 16), w = w8, Cout = the channels of dx to produce (multiple of 8; 128-column tiles
 * + one narrower launch for the rest), accumulate 0 / 1 (dx's read-modify-write in bf16, as crd_conv_igemm), no statistics.
 * The activation scale is read from DEVICE memory (one float): it is this step's or the previous step's amax / 448
 * (crd_fp8_scale_update), never a host constant -- gradients shrink by orders of magnitude over a run. */
int crd_conv3x3_fp8_dgrad(const crd_conv_desc* d, const float* w_scales, const float* x_scale_dev, crd_stream_t stream);
/* crd_gn_bwd_apply (bf16 dx, no accumulation, no dx2) that ALSO writes dx_fp8 = e4m3(bf16(dx) / *scale_dev) and folds max |bf16(dx)|
 * into amax_slots[CRD_FP8_AMAX_SLOTS] (u32 max on the float's bits; one slot per workgroup, spread so that the atomics do not
 * serialise on one address).  The bf16 dx is still written: the weight gradient reads it. */
#define CRD_FP8_AMAX_SLOTS 64
int crd_gn_bwd_apply_fp8(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                         int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                         int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                         const crd_sum_t* r, float* dgamma, float* dbeta, void* dx, int32_t dx_ld, int32_t dx_coff,
                         void* dx_fp8, int32_t dx8_ld, int32_t dx8_coff, const float* scale_dev, uint32_t* amax_slots,
                         crd_stream_t stream);
/* For each of n tensors: a = max over amax_slots[i][0..CRD_FP8_AMAX_SLOTS); if a > 0: scales[i] = margin * a / 448; the slots are
 * zeroed unless keep_slots.  Delayed scaling: called once per step at the head of the backward pass; with n = 1 right behind
 * crd_gn_bwd_apply_fp8 it is just-in-time scaling (the calibration iteration and the eager module path: followed by
 * crd_quant_fp8_dev).  Round 6: one scale per decoder STAGE (its three K-concatenated gradient slices): just-in-time updates behind the
 * first two layers keep the slots (running max), the third clears them. */
int crd_fp8_scale_update(uint32_t* amax_slots, float* scales, int32_t n, float margin, int32_t keep_slots, crd_stream_t stream);
/* crd_quant_fp8 with the scale read from device memory */
int crd_quant_fp8_dev(const void* x, int64_t rows, int32_t ld, int32_t coff, int32_t C, void* y, int32_t y_ld, int32_t y_coff,
                      const float* scale_dev, crd_stream_t stream);

/* Weight gradient of the same convolutions: dw[co][tap][ci] += sum_{b,oy,ox} dy[b,oy,ox,co] *
 * x[b, oy*s-p+ky, ox*s-p+kx, ci]  (fixed-point atomics, see crd_sum_t; caller zeroes dw).  Optionally also
 * dbias[co] += sum dy.  (autograd of the conv calls listed above.) */
typedef struct {
  const void* x; int32_t x_ld, x_coff; int32_t B, IH, IW, Cin;
  const void* dy; int32_t dy_ld, dy_coff; int32_t OH, OW, Cout;
  int32_t KH, KW, stride, pad;
  crd_sum_t* dw;      /* [Cout][KH*KW][Cin], CRD_GRAD_FRAC_BITS (untouched when dw_partials is used) */
  crd_sum_t* dbias;   /* [Cout] or NULL, CRD_GRAD_FRAC_BITS */
  float* dw_partials; /* optional, only where crd_conv_wgrad_splits(d) = S > 0 (the streaming 3x3 kernel): fp32
                         [S][Cout][9][Cin], contents don't-care.  Every pixel-range split stores its weight-gradient
                         block into its own copy with plain stores and dw is NOT touched: the gradient is the sum of the
                         S copies (crd_wgrad_unpack, replicas = S).  NULL: the splits add into dw with fp32 atomics. */
  int32_t dw_partial_capacity;   /* copies dw_partials holds.  It also CAPS the number of splits (= workgroups per channel
                                    chunk): the kernel uses min(natural split count, capacity) splits and writes exactly
                                    crd_conv_wgrad_splits(d) copies */
  int32_t wg_budget;             /* workgroups the streaming 3x3 kernel may use in total (0 = one per CU): a caller that runs
                                    other kernels next to this one passes fewer to leave them CUs.  The splits follow from it:
                                    budget / channel chunks, with a short last chunk (Cin = 136 / 144 / 200) given fewer
                                    workgroups and more rows each */
} crd_wgrad_desc;

int crd_conv_wgrad(const crd_wgrad_desc* d, crd_stream_t stream);
/* Number of pixel-range splits the 3x3 streaming kernel uses for this problem (honouring d->dw_partial_capacity > 0 as
 * a cap), 0 if the generic kernel handles it. */
int crd_conv_wgrad_splits(const crd_wgrad_desc* d);

/* Grouped weight gradients: the ~190 small wgrads of the encoder blocks (autograd's per-layer conv2d_backward weight
 * calls behind simplified_attention.py:32-43,95-132) are each too small to fill the chip, so the host collects them
 * per backward segment and runs them as ONE dispatch once their inputs exist.  crd_wgrad_group_build plans the group
 * on the host (problem descriptors + one work item per workgroup) into host_table; the caller copies the table to
 * device memory once and replays it with crd_conv_wgrad_grouped.  Calling build with host_table = NULL only fills
 * info->bytes (size query); a non-NULL table smaller than that is an error (CRD_E_INVALID).  The inputs named by the descriptors must stay valid until the grouped call.
 * Round 6: the items of a configuration are ordered for the 8 XCDs (workgroup index mod 8 = XCD): all tiles of one (problem, K split)
 * -- they re-read the same x and dy rows -- sit at indices of one residue, so the re-reads hit that XCD's L2; the list may contain
 * no-op items (problem index -1) where the XCDs' shares differ in length, and n_items counts them. */
typedef struct crd_wgrad_group_info {
  int32_t n_problems;
  int32_t n_items[4];      /* work items (workgroups) per tile configuration */
  int32_t item_offset[4];  /* first item of each configuration */
  int64_t bytes;           /* table size */
} crd_wgrad_group_info;
int crd_wgrad_group_build(const crd_wgrad_desc* descs, int32_t n, void* host_table, int64_t capacity,
                          crd_wgrad_group_info* info);
int crd_conv_wgrad_grouped(const void* dev_table, const crd_wgrad_group_info* info, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm family.  Statistics are kept as raw sums over 16-channel slabs ("g16 stats":
 * crd_sum_t [B][C/16][2] = sum, sum of squares, CRD_STAT_FRAC_BITS); a GroupNorm group is `gmul` consecutive slabs, so
 * the reference's GroupNorm(C/16, C) has gmul = 1 and Mlp.norm2 (groups from out_features,
 * simplified_attention.py:24) has gmul = hidden/dim.  Replaces torch group_norm + GELU:
 * simplified_attention.py:23-24,37-40,70,117-118,142,144,162,186; utils.py:213-214,225.
 * ------------------------------------------------------------------------------------------- */
/* stats[b][c/16] += sums over pixels of x (bf16 or fp32); optional per-channel sums chan[b][c][2]. */
int crd_gn_stats(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C,
                 crd_sum_t* stats, crd_sum_t* chan_sums, crd_stream_t stream);

/* y = act((x-mean)*rstd*gamma+beta) * mask[b][c];  act: 0 none, 1 exact GELU.  mask may be NULL.
 * y is a bf16 (y_f32=0) or fp32 (y_f32=1) pixel-major slice.  eps = 1e-5. */
int crd_gn_apply(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t P, int32_t C,
                 const crd_sum_t* stats, int32_t gmul, const float* gamma, const float* beta, int32_t act,
                 const float* mask, void* y, int32_t y_f32, int32_t y_ld, int32_t y_coff, crd_stream_t stream);

/* Backward, phase 1: per (b, c) sums  r[b][c] = (sum g, sum g*xhat), g = dy*mask*act'(u),
 * u = xhat*gamma+beta, followed by the per-group sums rg[b][grp] = (sum_c gamma_c r[b][c][0], sum_c gamma_c r[b][c][1]).
 * r is crd_sum_t[B*C*2 + B*(C/(16*gmul))*2] (CRD_GRAD_FRAC_BITS), zeroed by the caller.  dy is bf16 or fp32 pixel-major.
 * scratch (optional, >= B*1024*2*C floats is always enough): per-workgroup partial sums go there with plain stores and
 * a second small kernel folds them, instead of ~1M contended atomics on large tensors. */
int crd_gn_bwd_reduce(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                      int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                      int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                      crd_sum_t* r, float* scratch, int64_t scratch_capacity, crd_stream_t stream);
/* Backward, phase 2: dgamma[c] += sum_b r[b][c][1], dbeta[c] += sum_b r[b][c][0];
 * dx = (gamma*g - mean_grp(gamma*g) - xhat*mean_grp(gamma*g*xhat)) * rstd, written as bf16
 * (dx_f32=0) or ADDED into an fp32 tensor (dx_f32=1, accumulate).  dx2 (optional): a bf16 copy [B][P][dx2_ld] of the
 * finished dx, scaled per sample by scale2[b] when scale2 is given (the drop-path-scaled gradient the previous block's
 * fc2 backward consumes, simplified_attention.py:144). */
int crd_gn_bwd_apply(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, const void* dy, int32_t dy_f32,
                     int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t P, int32_t C, const crd_sum_t* stats,
                     int32_t gmul, const float* gamma, const float* beta, int32_t act, const float* mask,
                     const crd_sum_t* r, float* dgamma, float* dbeta, void* dx, int32_t dx_f32, int32_t dx_ld,
                     int32_t dx_coff, int32_t dx_accumulate, void* dx2, int32_t dx2_ld, const float* scale2,
                     crd_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Depthwise 3x3 (DWConv, simplified_attention.py:316,318-323), pixel-major bf16 [B][H][W][C].
 * w9 is fp32 [9][C] (tap-major), bias [C] or NULL.  flip=1 mirrors the taps (data gradient).
 * Optional g16 stats of the rounded output.
 * ------------------------------------------------------------------------------------------- */
int crd_dwconv3x3(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* w9, const float* bias,
                  int32_t flip, void* y, crd_sum_t* stats, const crd_sum_t* in_stats, int32_t in_gmul, const float* in_gamma,
                  const float* in_beta, const void* red_x, const crd_sum_t* red_stats, const float* red_gamma, crd_sum_t* red_r,
                  crd_stream_t stream);
/* red_x != NULL: the first phase of the backward of the GroupNorm (gmul = 1, no activation) whose dy this call produces
 * is fused in -- red_x is that GroupNorm's raw input (bf16 [B][H][W][C]), red_stats its g16 sums, red_gamma its weight,
 * red_r the buffer crd_gn_bwd_reduce would fill (crd_sum_t [B*C*2 + B*(C/16)*2], zeroed by the caller); only
 * crd_gn_bwd_apply remains to be called (Mlp.norm1's backward after the depthwise data gradient). */
/* in_stats != NULL (both functions): the input is GroupNorm-ed on load -- xn = bf16((x-mean)*rstd*gamma+beta) with the g16
 * sums in_stats[B][C/16][2] of x and groups of in_gmul slabs, zero padding applied after the normalisation -- i.e.
 * crd_gn_apply (Mlp.norm1, simplified_attention.py:37-38) fused into the consumer; the normalised tensor is never stored. */
/* dw10: crd_sum_t [replicas][10][C] (CRD_GRAD_FRAC_BITS), zeroed by the caller.  Rows 0..8: dw[tap][c] += sum dy*x_shifted; row 9: the bias
 * gradient sum dy.  Workgroups spread their atomics over the `replicas` copies (contended atomics on one copy
 * were the whole cost of this kernel); the true gradient is the sum of the copies (crd_wgrad_unpack does that). */
int crd_dwconv3x3_wgrad(const void* x, const void* dy, int32_t B, int32_t H, int32_t W, int32_t C, crd_sum_t* dw10,
                        int32_t replicas, const crd_sum_t* in_stats, int32_t in_gmul, const float* in_gamma,
                        const float* in_beta, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused Mlp of a Block (Block.forward / Mlp.forward, simplified_attention.py:34-43,141-145):
 *   x2 = x1 + dp * bf16( fc2( GELU( norm2( dwconv3x3( norm1( fc1( Block.norm2(x1) ) ) ) ) ) ) + b_fc2 )
 * for blocks whose Mlp.norm2 groups are exactly 64 hidden channels (hidden / (C / 16) == 64: ff_expansion 4, encoder stages
 * 3 and 4) and whose pixel grid fits in LDS (H * W <= 416): one workgroup per (sample, 64-channel slab of the hidden tensor)
 * runs the whole chain -- both GroupNorms are per sample and per slab, the depthwise conv per channel -- and leaves its fp32
 * contribution to fc2 in fc2_partials [hidden/64][B][H*W][C]; crd_mlp_reduce adds the slabs in order and finishes the
 * residual.  Two launches replace crd_gn_conv(fc1) + crd_dwconv3x3 + crd_gn_apply + crd_conv_igemm(fc2).
 * x1: fp32 residual stream [B][H*W][C] with its g16 sums x1_stats (Block.norm2's statistics); weights in the packed bf16
 * forms of crd_weight_pack (w_fc1 [hidden][C], w_fc2 [C][hidden]) and the depthwise fp32 [9][hidden] form.  Optional bf16
 * outputs (NULL = not stored; the backward pass and the weight gradients read them): xn = Block.norm2(x1) [B][H*W][C],
 * h1 = fc1 output, h2 = depthwise output, h3 = GELU(norm2(h2)), all [B][H*W][hidden].  h1_stats / h2_stats: their g16
 * sums [B][hidden/16][2], written with plain stores.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* x1; const crd_sum_t* x1_stats; const float* norm_gamma; const float* norm_beta;
  const void* w_fc1; const float* b_fc1; const float* norm1_gamma; const float* norm1_beta;
  const float* w9; const float* b_dw; const float* norm2_gamma; const float* norm2_beta;
  const void* w_fc2;
  void* xn; void* h1; void* h2; void* h3;
  crd_sum_t* h1_stats; crd_sum_t* h2_stats;
  float* fc2_partials;
  int32_t B, H, W, C, hidden;
} crd_mlp_desc;
/* number of 64-channel slabs (= first dimension of fc2_partials) if crd_mlp_fwd covers this shape, else 0 */
int crd_mlp_fused_supported(int32_t H, int32_t W, int32_t C, int32_t hidden);
int crd_mlp_fwd(const crd_mlp_desc* d, crd_stream_t stream);
/* x2 = x1 + dp[b] * bf16(sum_s fc2_partials[s] + b_fc2) (slabs added in order: reproducible; dp may be NULL = 1); optional
 * g16 sums `stats` [B][C/16][2] and per-channel sums `chan_sums` [B][C][2] of x2 (+=, the next block's norm1 statistics) */
int crd_mlp_reduce(const float* fc2_partials, int32_t slabs, const float* x1, const float* b_fc2, const float* dp, int32_t B,
                   int32_t N, int32_t C, float* x2, crd_sum_t* stats, crd_sum_t* chan_sums, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Max-pool attention (Attention_MaxPool.forward, simplified_attention.py:90-109).
 * q: bf16 [B][N][C], k: bf16 [B][M][C], C = heads*d.
 *   S[b][n] = sum_h max_m bf16(bf16(q_h.k_h)*scale)      idx[b][n][h] = argmax m
 * Since v = mean_n(x) is shared by all heads (:103) the output of `proj` is rank one:
 *   proj(out)[b][n][c] = u[b][c]*S[b][n] + bp[c],  u[b] = Wp xbar[b],  xbar = mean_n GroupNorm(x).
 * ------------------------------------------------------------------------------------------- */
int crd_attn_scores(const void* q, const void* k, int32_t B, int32_t N, int32_t M, int32_t heads, int32_t d,
                    float scale, float* S, int16_t* idx, crd_stream_t stream);
/* Config 5 experiment (round 6; BASELINE.json configs[4] "fp8 MFMA attention"): crd_attn_scores on e4m3 operands
 * (v_mfma_scale_f32_32x32x64_f8f6f4, one K = 64 MFMA per 32-key tile).  q8 [B][N][heads][64], k8 [B][M][heads][64]: e4m3(x / scale) per
 * tensor, the head dimension zero-padded to 64 bytes (crd_quant_fp8 per head into a zeroed buffer); qk_scale = q_scale * k_scale;
 * s = max_m bf16(bf16(qk_scale * (q8 . k8)) * scale), outputs as crd_attn_scores.  Measured against the bf16 kernel and NOT used by the
 * plan (DESIGN.md "Round 6": the kernel is bound by the arg-max bookkeeping behind the MFMA, not by the MFMA or its operand bytes). */
int crd_attn_scores_fp8(const void* q8, const void* k8, int32_t B, int32_t N, int32_t M, int32_t heads, float qk_scale, float scale,
                        float* ssum, int16_t* idx, crd_stream_t stream);
/* crd_attn_scores and crd_attn_xbar_proj in ONE launch (the value path is one extra workgroup per sample; it only needs
 * norm1's sums, so it rides along with the scores): arguments as in the two calls, C = heads * d. */
int crd_attn_fwd(const void* q, const void* k, int32_t B, int32_t N, int32_t M, int32_t heads, int32_t d, float scale,
                 float* S, int16_t* idx, const crd_sum_t* chan_sums, const crd_sum_t* stats, const float* gamma, const float* beta,
                 const void* w_fwd, void* xbar, float* u, crd_stream_t stream);
/* xbar[b][c] (bf16) from per-channel sums chan[b][c][2] and g16 stats of x (gmul = 1). */
int crd_attn_xbar(const crd_sum_t* chan_sums, const crd_sum_t* stats, const float* gamma, const float* beta, int32_t B,
                  int32_t N, int32_t C, void* xbar, crd_stream_t stream);
/* crd_attn_xbar followed by u[b][co] = sum_ci Wp[co][ci] * xbar[b][ci] (the `proj` Conv1d applied to the rank-one value,
 * simplified_attention.py:103-107) in one launch; w_fwd is Wp in its packed bf16 forward form [C][C].  u: fp32 [B][C]. */
int crd_attn_xbar_proj(const crd_sum_t* chan_sums, const crd_sum_t* stats, const float* gamma, const float* beta, const void* w_fwd,
                       int32_t B, int32_t N, int32_t C, void* xbar, float* u, crd_stream_t stream);
/* Backward of that path: tb = bf16(t) (bf16 [B][C], the dy operand of proj's weight gradient) and
 * es[b][ci] = inv_n * sum_co Wp[co][ci] * tb[b][co] (fp32 [B][C]); w_dgrad is Wp in its packed bf16 data-gradient form
 * [C][Cpad]. */
int crd_attn_vec_bwd(const crd_sum_t* t, const void* w_dgrad, int32_t B, int32_t C, int32_t Cpad, float inv_n, void* tb, float* es,
                     crd_stream_t stream);
/* x1 = x + dp[b]*bf16(u[b][c]*S[b][n] + bp[c])   (fp32 residual stream; Block.forward :143; dp may be NULL) */
int crd_attn_out_residual(const float* x, const float* u, const float* S, const float* bp, const float* dp,
                          int32_t B, int32_t N, int32_t C, float* x1, crd_stream_t stream);
/* crd_attn_out_residual that also accumulates the g16 GroupNorm sums of x1 (what crd_gn_stats(x1) would add to `stats`,
 * crd_sum_t [B][C/16][2], zeroed by the caller): Block.norm2 reads x1 next (simplified_attention.py:143-144). */
int crd_attn_out_residual_stats(const float* x, const float* u, const float* S, const float* bp, const float* dp,
                                int32_t B, int32_t N, int32_t C, float* x1, crd_sum_t* stats, crd_stream_t stream);
/* with dy = dp[b]*dx1:  t[b][c] += sum_n dy*S ; dbp_rows[b][c] += sum_n dy ; dS[b][n] = sum_c dy*u[b][c].
 * t and dbp_rows are crd_sum_t [B][C] (CRD_GRAD_FRAC_BITS), zeroed by the caller; the bias gradient is the sum of the B rows of dbp_rows: per-sample rows keep the
 * chain of contended atomics at the workgroups of one sample; crd_wgrad_unpack (replicas = B) folds them. */
int crd_attn_out_bwd(const float* dx1, const float* u, const float* S, const float* dp, int32_t B, int32_t N,
                     int32_t C, crd_sum_t* t, crd_sum_t* dbp_rows, float* dS, crd_stream_t stream);
/* The same with the apply phase of the preceding GroupNorm's backward fused in (Block.norm2, simplified_attention.py:144: gmul = 1,
 * no activation, fp32 input x = the residual stream, dxn = bf16 d(norm2(x)) from fc1's data gradient, r = the sums
 * crd_gn_bwd_reduce / the fused data-gradient epilogue left): first dx1 += crd_gn_bwd_apply(x, dxn) (written back: dx1 is the
 * residual gradient the block's backward continues with), then crd_attn_out_bwd on the result; dgamma / dbeta (both or neither)
 * += the GroupNorm's parameter gradients.  One launch and one pass over dx1 less per block. */
int crd_attn_out_bwd_gn(float* dx1, const float* u, const float* S, const float* dp, int32_t B, int32_t N, int32_t C,
                        crd_sum_t* t, crd_sum_t* dbp_rows, float* dS, const float* x, const void* dxn, const crd_sum_t* stats,
                        const float* gamma, const crd_sum_t* r, float* dgamma, float* dbeta, crd_stream_t stream);
/* dq[b][n][c] = scale*dS[b][n]*k[b][idx][c] (bf16) ; dk[b][m][c] += scale*dS*q (caller zeroes).
 * Every workgroup accumulates its share of dk in LDS (when [M][C] fp32 fits: always at the reference's sizes).  With
 * dk_partials != NULL (float [P][B][M][C], P = crd_attn_scores_bwd_partials(B,N,M,heads,d) > 0; contents don't-care)
 * the workgroups store their accumulators there with plain stores and dk is not touched: dk = sum over P, which
 * crd_sum_partials_bf16 folds together with the bf16 conversion the next layer needs.  With dk_partials == NULL they add
 * into dk (crd_sum_t [B][M][C], CRD_GRAD_FRAC_BITS) with atomics (2.7 M of them per launch at stage 1: ~16 us at the
 * ~170 G/s the L2s sustain).  Inside a workgroup the pixels routed to a key are found through a per-key bitmask and added
 * in ascending pixel order: the partials are reproducible bit for bit. */
int crd_attn_scores_bwd_partials(int32_t B, int32_t N, int32_t M, int32_t heads, int32_t d);
int crd_attn_scores_bwd(const void* q, const void* k, const float* dS, const int16_t* idx, int32_t B, int32_t N,
                        int32_t M, int32_t heads, int32_t d, float scale, void* dq, crd_sum_t* dk, float* dk_partials,
                        crd_stream_t stream);
/* crd_attn_scores_bwd and crd_attn_vec_bwd in ONE launch (both consume crd_attn_out_bwd's outputs; one extra workgroup per
 * sample runs the vector path): arguments as in the two calls, C = heads * d. */
int crd_attn_bwd(const void* q, const void* k, const float* dS, const int16_t* idx, int32_t B, int32_t N, int32_t M,
                 int32_t heads, int32_t d, float scale, void* dq, crd_sum_t* dk, float* dk_partials, const crd_sum_t* t,
                 const void* w_dgrad, int32_t Cpad, float inv_n, void* tb, float* es, crd_stream_t stream);
/* dst[i] = bf16(sum_{r < replicas} part[r*replica_stride + i]), i < n (n, replica_stride multiples of 8) */
int crd_sum_partials_bf16(const float* part, int32_t replicas, int64_t replica_stride, void* dst, int64_t n,
                          crd_stream_t stream);
/* dst[i] = bf16(src[i] * 2^-CRD_GRAD_FRAC_BITS), i < n (n a multiple of 8): the dk accumulator of crd_attn_scores_bwd's
 * global path as the bf16 operand the next layer needs */
int crd_gsum_to_bf16(const crd_sum_t* src, void* dst, int64_t n, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Bicubic x2 up-sampling, A=-0.75, align_corners=False, clamped borders (nn.Upsample, utils.py:241,
 * 251).  bf16 slice -> bf16 slice; the backward is the exact transpose.
 * ------------------------------------------------------------------------------------------- */
int crd_bicubic2x(const void* x, int32_t x_ld, int32_t x_coff, int32_t B, int32_t H, int32_t W, int32_t C, void* y,
                  int32_t y_ld, int32_t y_coff, crd_stream_t stream);
int crd_bicubic2x_bwd(const void* dy, int32_t dy_ld, int32_t dy_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                      void* dx, int32_t dx_ld, int32_t dx_coff, int32_t accumulate, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Layout / small elementwise helpers at the module boundary.
 * ------------------------------------------------------------------------------------------- */
/* NCHW fp32 [B][C][H][W] -> pixel-major bf16 slice; channels C..Cpad-1 of the slice are zeroed. */
int crd_nchw_to_pm(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, void* y, int32_t y_ld, int32_t y_coff,
                   int32_t Cpad, crd_stream_t stream);
/* pixel-major (bf16 or fp32) -> NCHW fp32 */
int crd_pm_to_nchw(const void* x, int32_t x_f32, int32_t x_ld, int32_t x_coff, int32_t B, int32_t C, int32_t H,
                   int32_t W, float* y, crd_stream_t stream);
/* Seg_Block (utils.py:95-100): y[row*y_ld + y_coff] = argmax_c logits[row][c] / num_classes (first max wins);
 * y is bf16 (y_f32=0, a channel of a pixel-major buffer) or fp32 (y_f32=1, the module output) */
int crd_seg_argmax(const void* logits, int32_t ld, int32_t B, int32_t P, int32_t C, int32_t num_classes, void* y,
                   int32_t y_f32, int32_t y_ld, int32_t y_coff, crd_stream_t stream);
/* dst[i] = scale * src[i]  (fp32) */
int crd_scale_f32(const float* src, float* dst, int64_t n, float scale, crd_stream_t stream);
/* dst[row][d_coff+c] (+)= src[row][s_coff+c]   (bf16 slices) */
int crd_slice_copy(const void* src, int32_t s_ld, int32_t s_coff, void* dst, int32_t d_ld, int32_t d_coff,
                   int64_t rows, int32_t C, int32_t accumulate, crd_stream_t stream);
/* dst[row][d_coff+c] = bf16(scale[row / rows_per_sample] * src[row*s_ld + c] + add[row][add_coff+c]), c < C
 * (scale and add may be NULL; add is a bf16 slice) */
int crd_f32_to_bf16_rows(const float* src, int32_t s_ld, void* dst, int32_t d_ld, int32_t d_coff, int64_t rows,
                         int32_t C, const float* scale, int64_t rows_per_sample, const void* add, int32_t add_ld,
                         int32_t add_coff, crd_stream_t stream);
/* Depth_Activation.conv_2 (utils.py:283,288): 3x3 conv 32 -> 1 (+bias) as a stencil-reduce.  a: bf16 [B][H][W][32]
 * (the sigmoid output), w: fp32 reference layout [1][32][3][3], depth: fp32 [B][H][W]; `copy` (optional) receives
 * the bf16 depth as one channel of a pixel-major buffer (the reference's torch.cat([stage, depth]), CamRaDepth.py:120,146). */
int crd_head_conv2_fwd(const void* a, const float* w, const float* bias, int32_t B, int32_t H, int32_t W, float* depth,
                       void* copy, int32_t copy_ld, int32_t copy_coff, crd_stream_t stream);
/* Backward of the above fused with the sigmoid backward: dy = gd (+ add, a bf16 channel);
 * dz = a(1-a) * conv2^T(dy) (bf16 [B][H][W][32]).  Parameter gradients: dw_rows is crd_sum_t [replicas][289] (CRD_GRAD_FRAC_BITS), zeroed by the
 * caller; every row holds dw[1][32][3][3] (288 values, reference order) followed by dbias, workgroups spread their
 * fp32 atomics over the rows and the gradient is the sum of the rows (crd_wgrad_unpack, replicas = rows). */
int crd_head_conv2_bwd(const float* gd, const void* add, int32_t add_ld, int32_t add_coff, const void* a, const float* w,
                       int32_t B, int32_t H, int32_t W, void* dz, crd_sum_t* dw_rows, int32_t replicas, crd_stream_t stream);
/* The two halves of crd_head_conv2_bwd as calls of their own (the weight-gradient half is not on the backward pass's
 * dependency chain: the graph step replays it on its late stream). */
int crd_head_conv2_bwd_data(const float* gd, const void* add, int32_t add_ld, int32_t add_coff, const void* a, const float* w,
                            int32_t B, int32_t H, int32_t W, void* dz, crd_stream_t stream);
int crd_head_conv2_wgrad(const float* gd, const void* add, int32_t add_ld, int32_t add_coff, const void* a, int32_t B, int32_t H,
                         int32_t W, crd_sum_t* dw_rows, int32_t replicas, crd_stream_t stream);
/* da <- da * a * (1-a)   (bf16, n elements, n % 8 == 0): backward of the sigmoid in Depth_Activation */
int crd_sigmoid_bwd(const void* a, void* da, int64_t n, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Weight packing: fp32 reference layout [Cout][Cin_ref][taps] -> bf16 [Cout][taps][Cin_pad]
 * and the two transposed forms the data-gradient kernels read; table-driven, one launch.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* src;    /* [Cout][Cin_ref][taps] fp32 (reference layout) */
  void* dst_fwd;       /* bf16 [Cout][taps][Cin_pad]      or NULL */
  void* dst_dgrad;     /* bf16 [Cin_pad][taps][Cout_pad]  or NULL   (gather_mode 1) */
  void* dst_scatter;   /* bf16 [taps][Cin_pad][Cout_pad]  or NULL   (out_mode 1)    */
  const int32_t* cmap; /* [Cin_pad]: reference input channel of each packed channel, -1 = zero; NULL = identity */
  int32_t Cout, Cin_ref, taps, Cin_pad, Cout_pad;
  int32_t dst_f32;     /* 1: dst_fwd is written as fp32 (plain copies of vectors); 2: fp32 holding the bf16-ROUNDED value (depthwise
                          weights [9][C]: what autocast feeds the convolution -- the reference casts conv weights to the low
                          precision, src/main/runner.py:191) */
  /* Round 6: K-CONCATENATED data-gradient weights (the write-once data gradients of a ShortResBlock, src/utils/utils.py:127-135 under
   * autograd: dx[n] of the concat buffer = sum over the layers that read channel n).  dgrad_ld > 0: dst_dgrad is a matrix
   * [rows][taps][dgrad_ld] shared by several layers; this entry writes its Cout_pad columns at column offset dgrad_coff for the input
   * channels [dgrad_row0, dgrad_row0 + dgrad_rows) as rows 0 .. dgrad_rows - 1.  dgrad_ld == 0: the plain form above. */
  int32_t dgrad_ld, dgrad_coff, dgrad_row0, dgrad_rows;
} crd_pack_entry;
int crd_weight_pack(const crd_pack_entry* table_dev, int32_t n, int64_t max_elems, crd_stream_t stream);
/* grad_ref[co][ci_ref][tap] (+)= dw_packed[co][tap][ci_pad] */
typedef struct {
  const void* src;     /* [Cout][taps][Cin_pad]: fp32 (src_sum = 0) or crd_sum_t with CRD_GRAD_FRAC_BITS (src_sum = 1) */
  float* dst;          /* fp32 [Cout][Cin_ref][taps] */
  const int32_t* cmap;
  int32_t Cout, Cin_ref, taps, Cin_pad;
  int32_t replicas;        /* > 1: the source is the sum of `replicas` copies, replica_stride ELEMENTS apart (summed in
                              index order: fixed, so reproducible) */
  int32_t src_sum;
  int64_t replica_stride;
} crd_unpack_entry;
int crd_wgrad_unpack(const crd_unpack_entry* table_dev, int32_t n, int64_t max_elems, int32_t accumulate,
                     crd_stream_t stream);

/* Train-mode masks of timm DropPath (simplified_attention.py:123,143-144) and nn.Dropout2d
 * (CamRaDepth.py:96): out[r][c] = Bernoulli(keep[r]) / keep[r].  *counter (device) is advanced by
 * the call so that a replayed HIP graph draws new masks. */
int crd_dropout_masks(float* out, const float* keep, int32_t rows, int32_t cols, uint64_t seed, uint64_t* counter,
                      crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Batch assembly (NuscenesDataset.__getitem__, src/data/dataloader.py:202-333) -- SURVEY 8f N1.
 * ------------------------------------------------------------------------------------------- */
/* out: fp32 [B][7 or 6][H][W].  Channels 0-2: (img/255 - mean)/std of the uint8 [B][H][W][3] image in the order it was
 * read (cv2 BGR; the reference applies the RGB ImageNet constants as is, :226-233); 3: clip(radar[...,0], 0, max_depth) /
 * max_depth (:304-306); 4-5: radar[...,1..2] (:309-310); 6: rad_vel (:315-318; NULL -> 6 channels).  radar: fp32
 * [B][H][W][3], rad_vel: fp32 [B][H][W]. */
int crd_assemble_input(const void* img_u8, const float* radar, const float* rad_vel, int32_t B, int32_t H, int32_t W,
                       float max_depth, float* out, crd_stream_t stream);
/* Ground-truth pyramid (:236-257): full = inverse-normalised depth (clip to [0, max], g > 0 -> (max - g)/max); half,
 * quarter, eighth = successive zero-ignoring 3x3 / stride 2 / pad 1 min-pools (:213-222); lower levels may be NULL. */
int crd_gt_pyramid(const float* depth, int32_t B, int32_t H, int32_t W, float max_depth, float* full, float* half,
                   float* quarter, float* eighth, crd_stream_t stream);
/* cv2.resize(image, (DW, DH), interpolation=cv2.INTER_NEAREST) of dataloader.py:227 on interleaved uint8 pixels
 * [B][SH][SW][C] -> [B][DH][DW][C]: source column = min(floor(dx * (1 / (DW / SW))), SW - 1) in double, rows alike. */
int crd_resize_nearest_u8(const void* src, int32_t B, int32_t SH, int32_t SW, int32_t C, void* dst, int32_t DH, int32_t DW,
                          crd_stream_t stream);
/* The segmentation targets of dataloader.py:262-267: skimage.transform.resize(mseg[:rows], (DH, DW), order=0,
 * preserve_range=True, anti_aliasing=False) (scikit-image 0.19.3 -> scipy.ndimage.zoom(order=0, grid_mode=True)): source
 * index = floor(((o + 0.5) * (S / D) - 0.5) + 0.5) in double.  uint8 [B][SH][SW] label maps in, of which the first
 * `rows` rows are used; int64 [B][DH][DW] labels out (the dtype runner.py:189-190 casts to). */
int crd_resize_labels_nearest(const void* src_u8, int32_t B, int32_t SH, int32_t SW, int32_t rows, int64_t* dst, int32_t DH,
                              int32_t DW, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Losses (src/utils/loss_funcs.py:14-46,77-91; combination src/main/runner.py:197-218).
 * ------------------------------------------------------------------------------------------- */
/* acc[0] += sum smooth_l1(pred-target), acc[1] += #(target>0), acc[2] += sum (target-pred)^2 ; crd_sum_t with
 * CRD_STAT_FRAC_BITS (so that a SUM all-reduce over ranks stays exact); caller zeroes acc */
int crd_masked_l1_fwd(const float* pred, const float* target, int64_t n, crd_sum_t* acc, crd_stream_t stream);
/* Trainer.test metrics (runner.py:443-465) for `frames` fp32 maps of n pixels each, without a host sync per frame:
 * p = clip(pred,0,1)*max_depth, g = gt*max_depth with g > max_distance dropped; on g > 0:
 * acc[f][0] += sum |p-g|, acc[f][1] += sum (p-g)^2, acc[f][2] += sum |p-g|/g, acc[f][3] += count   (crd_sum_t,
 * CRD_STAT_FRAC_BITS; caller zeroes acc;
 * MAE = acc0/acc3, RMSE = sqrt(acc1/acc3), REL = acc2/acc3; frames with acc3 == 0 are skipped by the reference) */
int crd_test_metrics(const float* pred, const float* gt, int32_t frames, int64_t n, float max_depth, float max_distance,
                     crd_sum_t* acc, crd_stream_t stream);
/* Per-frame confusion matrices for the Jaccard index of Trainer.test (runner.py:432-436, torchmetrics 0.10.2
 * JaccardIndex(num_classes=C, ignore_index=255)): prediction = arg-max over the C fp32 logits [frames][C][HW],
 * confmat[f][target][pred] += 1 (int64 [frames][C][C], caller zeroes); labels outside [0, C) are skipped and counted
 * in out_of_range[f] (torchmetrics raises ValueError on them, which the reference catches: that frame's IoU is NaN). */
int crd_seg_confusion(const float* logits, const int64_t* labels, int32_t frames, int32_t C, int64_t HW, int64_t* confmat,
                      int64_t* out_of_range, crd_stream_t stream);
/* dpred = gmul * gout[0] * clamp(pred-target,-1,1) / acc[1] on target>0, else 0   (gout may be NULL = 1) */
int crd_masked_l1_bwd(const float* pred, const float* target, int64_t n, const crd_sum_t* acc, const float* gout,
                      float gmul, float* dpred, crd_stream_t stream);
/* NCHW fp32 logits [B][C][HW], int64 labels [B][HW], ignore_index 255:
 * acc[0] += sum -log softmax[label], acc[1] += #valid */
int crd_ce_fwd(const float* logits, const int64_t* labels, int32_t B, int32_t C, int64_t HW, crd_sum_t* acc,
               crd_stream_t stream);
/* focal on the scalar mean CE (loss_funcs.py:27-29): dlogits = gmul*gout[0]*dF/dce*(softmax-onehot)/acc[1] */
int crd_ce_focal_bwd(const float* logits, const int64_t* labels, int32_t B, int32_t C, int64_t HW, const crd_sum_t* acc,
                     const float* gout, float gmul, float* dlogits, crd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * diffGradNorm.step (src/models/diffGradNorm.py:41-113) over flat fp32 buffers.
 * Tensor t occupies [seg_off[2t], seg_off[2t+1]) of every flat buffer (int64 pairs; gaps allowed).  Workgroup w processes chunk
 * blk2chunk[w] (4096 elements) of tensor blk2seg[w]; the workgroups of a tensor are consecutive and in chunk order.
 * exp_grad_norm / factor are float[n_tensors]; norm_sq is scratch, float[n_blocks]: workgroup w stores its part of
 * ||g||^2 in norm_sq[w] and one wave per tensor adds its workgroups' parts in a fixed order (reproducible; the e > n
 * branch of diffGradNorm.py:84 turns rounding noise into a discrete jump).  active[t]=0 skips tensor t
 * (`p.grad is None`, :54-55).  `step` is the 1-based step count used for the bias corrections.
 * hp_dev (optional, device float[5] = beta1, beta2, eps, weight_decay, lr*sqrt(1-beta2^t)/(1-beta1^t+1e-8))
 * overrides the scalar arguments so that a captured HIP graph can follow a per-iteration schedule.
 * ------------------------------------------------------------------------------------------- */
int crd_diffgradnorm_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* prev_grad,
                          float* exp_grad_norm, float* norm_sq, float* factor, const int64_t* seg_off,
                          const int32_t* blk2seg, const int32_t* blk2chunk, int32_t n_tensors, int32_t n_blocks,
                          const uint8_t* active, float lr, float beta1, float beta2, float eps, float weight_decay,
                          int32_t step, const float* hp_dev, crd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
