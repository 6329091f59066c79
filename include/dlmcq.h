/*
 * dlmcq.h - C ABI of the MI355X (gfx950) fake-quantize path.
 *
 * This is the drop-in boundary for DLMC-QUANT's fake-quantize hot path.  The reference has no
 * FFI of its own: the path is a chain of ATen elementwise/reduction ops issued from Python
 * (reference files cited per entry point below, relative to the reference checkout).  Each entry
 * point here replaces one such chain with ONE hand-written HIP kernel (or a two-launch
 * reduce/finalize pair); the Python layer in `dlmc-quant_amd/dlmc/` binds them with ctypes and
 * re-exports the reference's own names (`quantize`, `emulate_quantize`, `get_qparams_tensor`,
 * `QConv2d`, ... see INTEGRATION.md).
 *
 * Contract:
 *   - plain pointers and sizes only; no torch / C++ types cross the boundary;
 *   - every buffer (x, y, codes, scale, offset, scratch) is caller-allocated DEVICE memory; the
 *     library owns nothing and keeps no global mutable state (all functions are re-entrant);
 *   - all work is enqueued asynchronously on the caller's `stream` (a hipStream_t passed as
 *     void*), on the caller's current device; nothing here synchronises the device or allocates,
 *     so every entry point is hipGraph-capturable;
 *   - return value: 0 = success; < 0 = a DLMCQ_E* argument error (nothing was launched);
 *     > 0 = the hipError_t of the failed launch.  No exception or abort crosses the boundary.
 *
 * Tensor addressing.  Every elementwise entry point sees its tensor as (outer, channels, inner),
 * contiguous, with one (scale, offset) pair per channel:
 *     per-tensor            : outer = 1, channels = 1, inner = numel
 *     KCRS weights, axis 0  : outer = 1, channels = K, inner = C*R*S      (scale [K,1,1,1])
 *     NCHW activations, ax 1: outer = N, channels = C, inner = H*W        (scale [1,C,1,1])
 *     (N,C) activations     : outer = N, channels = C, inner = 1
 * No transpose copy is ever made (the reference materialises one: ops.py:112-118).
 */
#ifndef DLMCQ_H
#define DLMCQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLMCQ_VERSION 100 /* 0.1.0 */

typedef void* dlmcq_stream_t; /* hipStream_t */

/* ---- error codes (negative; positive values are hipError_t) ---- */
#define DLMCQ_OK 0
#define DLMCQ_EINVAL (-1)       /* null pointer, negative size, lo > hi, unknown enum value */
#define DLMCQ_ERANGE (-2)       /* a size exceeds what the kernels index (see each function) */
#define DLMCQ_ESCRATCH (-3)     /* scratch buffer too small or missing */
#define DLMCQ_EALIGN (-4)       /* a pointer violates its stated alignment */

/* ---- fake-quant forms: the five clamp-round-dequant expressions of the reference ----
 * r(v) = rint(v) (round half to even); R(v) = (r(v) - v) + v, the forward value of the
 * reference's `round_pass` (utils.py:29-32): equal to r(v) but -0 -> +0 and +-inf -> NaN.
 * clamp() propagates NaN like torch.clamp.  All arithmetic is IEEE fp32, no FMA contraction. */
#define DLMCQ_FORM_EMULATE 0   /* utils.py:1-11 emulate_quantize:
                                  q = clamp(r((x-o)/(s+1e-7)), lo, hi);       y = q*s + o       */
#define DLMCQ_FORM_QBASE 1     /* modules/base.py:96-102,131-133 (QBase.forward):
                                  s^ = (s - s*g) + s*g  (grad_scale, utils.py:24-27)
                                  q = R(clamp((x-o)/s^, lo, hi));              y = q*s^ + o      */
#define DLMCQ_FORM_ZEROPOINT 2 /* FSPTQuant/base.py:108-109 (FSPTQ activations):
                                  q = clamp(R(x/s) + zp, lo, hi);              y = (q - zp)*s    */
#define DLMCQ_FORM_SYMMETRIC 3 /* FSPTQuant/base.py:149-152 (FSPTQ weights; offset unused):
                                  q = clamp(R(x/s), lo, hi);                   y = q*s           */
#define DLMCQ_FORM_ROOTQ_ACT 4 /* RootQ/base.py:106-111 + RootQ/function.py:15-20 (offset unused):
                                  u = s*(hi-lo); t = x + relu(0-x); t = t - relu(t-u)
                                  q = R(t/s);                                  y = q*s           */
#define DLMCQ_FORM_COUNT 5
/* OR-able into the `q_form` argument of the int8 convolution entry points (the consumer's quantiser evaluated in the
 * epilogue) when its range is an unsigned byte (0 <= q_lo, q_hi <= 255): the codes are stored as int8 `code - 128`
 * (each byte ^ 0x80) instead of uint8 `code`.  A consumer then takes them as signed codes (x_is_unsigned = 0) with the
 * zero point `zp - 128`: the same integers reach the matrix cores (they multiply signed bytes, so uint8 codes are
 * re-centred on every operand read otherwise), hence the same results.  Accepted by dlmcq_conv2d_i8_nhwc_fused / _asym /
 * _dual, by dlmcq_quantize_pad_nhwc4 (its `form`: the padded image buffer then holds `code - 128`, border included) and for
 * the SECOND quantiser of the chain entry points (the first one's codes are read in place by the second GEMM);
 * every other entry point returns DLMCQ_EINVAL for it. */
#define DLMCQ_EMIT_SHIFT128 0x100
/* OR-able into the LAST quantiser form argument of the chain entry points (`q2_form` of dlmcq_conv2d_i8_nhwc_chain, `q3_form` of
 * dlmcq_conv2d_i8_nhwc_dual_chain): the second 1x1 layer's weight codes are handed over CHUNK-MAJOR, int8 [K / 64][K2][64]
 * (element [n][k2][j] = KRSC element [k2][64 n + j]), so that the 64-column chunk the kernel stages per step is one contiguous
 * K2 x 64 byte block (whole cache lines per LDS-DMA instruction instead of 64-byte pieces of K-byte rows).  Same results. */
#define DLMCQ_W2_CHUNK_MAJOR 0x200
/* DLMCQ_FP32_IN_CHUNK_MAJOR / DLMCQ_FP32_OUT_CHUNK_MAJOR (OR-able into the LAST quantiser form - `q2_form` / `q3_form` - of the two chain
 * entry points): the fp32 shortcut the call reads / the fp32 block output it writes, [M][K], is CHUNK-MAJOR: [K / 64][M][64], every
 * 64-channel chunk a plane of M rows x 256 bytes (K a multiple of 64).  The kernels walk a tile of rows chunk by chunk; with row-major
 * tensors the chip's workgroups then touch 256-byte pieces K * 4 bytes apart, spread over hundreds of megabytes, and HBM serves that
 * pattern at 4.6 - 5.8 TB/s where the same loads and stores reach 5.9 - 6.0 on planes in which neighbouring workgroups' pieces are
 * neighbours (tools/probes/stream_pattern_probe.hip, round 5; the chain launches themselves: -11 ... -18 % at 14^2 and 56^2).  A private
 * layout between two of these calls: same values, another place in the buffer.  The two tensors of a call may differ in layout, except
 * for the 128 -> K -> 128 instantiation (DLMCQ_EINVAL).  A call with one fp32 tensor takes either bit for it.
 * The same two bits in `q_form` of dlmcq_conv2d_i8_nhwc_fused (its `residual` / `out`) and dlmcq_conv2d_i8_nhwc_dual (`out`): honoured where
 * the block-end kernel takes the call (DLMCQ_ROUTE_PWR - ask with DLMCQ_ROUTE_ONLY, without the bits), one layout per call; anywhere else the
 * call is refused with DLMCQ_EINVAL rather than run on a kernel that would read or write the tensor row-major. */
#define DLMCQ_FP32_IN_CHUNK_MAJOR 0x2000
#define DLMCQ_FP32_OUT_CHUNK_MAJOR 0x4000
/* Two control bits, OR-able into the `q_form` argument of dlmcq_conv2d_i8_nhwc_fused / _asym / _dual and dlmcq_conv2d_dw_i8_nhwc.
 * DLMCQ_FORCE_TILED: the call runs on the generic kernel of its family (conv_i8_mfma_kernel; conv_dw3*_i8_kernel for the
 * depthwise entry point) even where the library's dispatch would hand it to a specialised one (halo-tile 3x3, weight-resident
 * pointwise, block-end pointwise, matrix-core depthwise).  Same results bit for bit; it exists so that a specialised kernel and
 * the kernel it replaces can be compared on ONE tensor at any size (tests/, tools/) and for same-box A/B timing.
 * DLMCQ_ROUTE_ONLY: nothing is launched; the call validates its arguments exactly as the real call does and returns WHICH kernel
 * the dispatch picks for them (DLMCQ_ROUTE_*, all > 0; errors stay negative) - the dispatch code itself answers, so a caller's
 * bookkeeping (bench.py's per-kernel-family rooflines) cannot drift from what runs. */
#define DLMCQ_FORCE_TILED 0x400
#define DLMCQ_ROUTE_ONLY 0x800
/* DLMCQ_PIPELINED (dlmcq_conv2d_i8_nhwc_fused; opt-in): a 3x3 / stride 1 layer of 128, 256 or 512 input channels with enough tiles runs on the
 * persistent halo kernel that is software-pipelined across tiles (csrc/conv3x3_pipe_i8.hip: tile t's quantising epilogue under tile t + 1's
 * K loop) instead of the plain halo-tile kernel.  Same bytes; measured 12 - 20 % SLOWER than the plain kernel (round 5, LABNOTES 15), hence
 * not the default - kept built and tested for same-box A/B timing. */
#define DLMCQ_PIPELINED 0x1000
#define DLMCQ_ROUTE_TILED 1   /* conv_i8_mfma_kernel (csrc/conv_i8.hip) */
#define DLMCQ_ROUTE_HALO3X3 2 /* conv3x3_halo_i8_kernel (csrc/conv3x3_i8.hip) */
#define DLMCQ_ROUTE_PW 3      /* conv_pw_i8_kernel (csrc/conv_pw_i8.hip) */
#define DLMCQ_ROUTE_PWR 4     /* conv_pwr_i8_kernel (csrc/conv_pwr_i8.hip) */
#define DLMCQ_ROUTE_DW 5      /* conv_dw_i8_kernel / conv_dw3_i8_kernel / conv_dw3p2_i8_kernel (csrc/conv_dw_i8.hip) */
#define DLMCQ_ROUTE_DWM 6     /* conv_dwm_i8_kernel (csrc/conv_dwm_i8.hip) */
#define DLMCQ_ROUTE_HALO3X3_PIPE 7 /* conv3x3_pipe_i8_kernel (csrc/conv3x3_pipe_i8.hip) */

/* ---- what is written to `y` ---- */
#define DLMCQ_Y_DEQUANT 0 /* the fake-quantised value y */
#define DLMCQ_Y_CODES 1   /* the integer code q as fp32 (reference `quantize`, utils.py:1-2) */

/* ---- integer code emission (no reference counterpart: it keeps codes in fp32) ---- */
#define DLMCQ_CODES_NONE 0
#define DLMCQ_CODES_I8 1  /* one byte per element: int8 when lo < 0, uint8 otherwise */
#define DLMCQ_CODES_P4 2  /* two codes per byte, element 2i in the low nibble, 2i+1 in the high
                             nibble; two's-complement nibbles when lo < 0.  Needs -8<=lo, hi<=15 */

int dlmcq_version(void);
const char* dlmcq_strerror(int code);

/*
 * Fake-quantise x -> y (and/or integer codes) in one pass: 4 B read + 4 B written per element
 * (+1 B int8 codes, +0.5 B packed int4).  Replaces the 6-9 separate ATen passes of
 * utils.py:9-11 / modules/base.py:102,133 / FSPTQuant/base.py:108-109,149-152 / RootQ/base.py:108-111.
 *   x, y     : fp32 [outer*channels*inner]; y may alias x; y may be NULL when only codes are wanted
 *   codes    : NULL or the buffer selected by codes_kind (4-byte aligned)
 *   scale    : fp32 [channels]        offset: fp32 [channels] or NULL (= 0)
 *   ste_g    : DLMCQ_FORM_QBASE only - the `g` of grad_scale; 0 gives s^ = s
 * Any pointer alignment is accepted (16-byte aligned x/y take the 128-bit path).
 * DLMCQ_ERANGE if channels*inner >= 2^31 with channels > 1, or the launch would exceed 2^31 blocks.
 */
int dlmcq_fake_quant_f32(const float* x, float* y, void* codes, const float* scale,
                         const float* offset, int64_t outer, int64_t channels, int64_t inner,
                         int32_t lo, int32_t hi, int32_t form, int32_t y_kind, int32_t codes_kind,
                         float ste_g, dlmcq_stream_t stream);

/*
 * Dequantise stored integer codes: the second half of every form above (reference `dequantize`,
 * utils.py:5-6, for forms EMULATE/QBASE).  codes_kind I8 or P4; `is_signed` selects int8/uint8 or
 * the nibble sign extension.  For DLMCQ_FORM_QBASE pass the same ste_g as at quantisation.
 */
int dlmcq_dequant_codes_f32(const void* codes, float* y, const float* scale, const float* offset,
                            int64_t outer, int64_t channels, int64_t inner, int32_t form,
                            int32_t codes_kind, int32_t is_signed, float ste_g,
                            dlmcq_stream_t stream);

/* Reference `dequantize` on fp32 codes (utils.py:5-6): y = q*s + o. */
int dlmcq_dequant_f32(const float* q, float* y, const float* scale, const float* offset,
                      int64_t outer, int64_t channels, int64_t inner, dlmcq_stream_t stream);

/* ---- observer (ops.py:20-34 per tensor, ops.py:112-140 per channel) ---- */
#define DLMCQ_MINMAX_ABSMAX 0 /* out_max[c] = max|x|;             out_min untouched (may be NULL) */
#define DLMCQ_MINMAX_MINMAX 1 /* out_max[c] = max x, out_min[c] = min x                           */
#define DLMCQ_MINMAX_NEGMIN 2 /* out_max[c] = max x, out_min[c] = -min x: [max | -min] is then one
                                 flat vector for a single all_reduce(MAX) across ranks (C2)       */

/* Bytes of device scratch the observer needs for this shape (partials of the first stage). */
size_t dlmcq_minmax_scratch_bytes(int64_t outer, int64_t channels, int64_t inner);

/*
 * One read of x (4 B per element), NaN-propagating like torch.max/min.  Two launches: a partial
 * reduction (wave shuffles -> LDS -> one partial per block) and a finalize over the partials.
 * Deterministic (no atomics); results do not depend on the launch geometry.
 */
int dlmcq_minmax_f32(const float* x, float* out_max, float* out_min, int64_t outer,
                     int64_t channels, int64_t inner, int32_t mode, void* scratch,
                     size_t scratch_bytes, dlmcq_stream_t stream);

/* The second stage of dlmcq_minmax_f32 alone, per tensor: `count` partials in three planes `plane_stride` floats apart ([max | min |
 * bits of max |x|]: what a stage-1 kernel or dlmcq_conv2d_i8_nhwc_fused_observed wrote) -> out_max[0], out_min[0] as `mode` says. */
int dlmcq_minmax_finalize_f32(const float* partials, int64_t count, int64_t plane_stride, float* out_max, float* out_min,
                              int32_t mode, dlmcq_stream_t stream);

/*
 * The arithmetic tail of quantize_minmax_{tensor,channel} on device (no host sync):
 *   signed  : scale = vmax / (2^(b-1)-1)              offset = 0         (vmax = max|x|)
 *   unsigned: scale = (vmax - vmin) / (2^b-1)         offset = vmin      (allow_offset != 0)
 *             scale = (vmax - 0) / (2^b-1)            offset = 0         (allow_offset == 0)
 * then scale += scale_eps when scale_eps != 0 (FSPTQuant/base.py:129 adds 1e-6).
 * `min_is_negated` != 0 when vmin holds -min (DLMCQ_MINMAX_NEGMIN, after the all-reduce).
 */
int dlmcq_qparams_from_minmax(const float* vmax, const float* vmin, float* scale, float* offset,
                              int64_t channels, int32_t n_bits, int32_t is_signed,
                              int32_t allow_offset, int32_t min_is_negated, float scale_eps,
                              dlmcq_stream_t stream);

/*
 * RootQ activation initialisation (RootQ/base.py:80): scale[c] = (vmax[c] - vmin[c]) / span with a true
 * IEEE division (span = hi - lo of the format).  `min_is_negated` as above.
 */
int dlmcq_span_scale_f32(const float* vmax, const float* vmin, float* scale, int64_t channels, float span,
                         int32_t min_is_negated, dlmcq_stream_t stream);

/*
 * LSQ initialisation (modules/base.py:84-85 input, :118-121 weight): scale[0] = 2 * mean|x| / sqrt(Qp), the QAT flow's default
 * first-call initialiser (example/quantization/LSQ_config.yaml, `type: "LSQ"`).  One read of x (4 B per element); per-lane
 * double partials, a fixed-order tree over one double per workgroup (no atomics: deterministic), then the reference's own
 * fp32 chain on the device: mean = fl32(sum) / n, 2 * mean, true IEEE division by `sqrt_qmax` (= the fp32 value of
 * math.sqrt(Qp) the caller computed).  The only difference from the reference's CPU result is the summation order of the mean
 * (<= a few ulp; tests state 2e-6 relative).  `scratch`: dlmcq_lsq_init_scratch_bytes() bytes, 8-byte aligned.
 */
size_t dlmcq_lsq_init_scratch_bytes(int64_t n);
int dlmcq_lsq_init_f32(const float* x, float* scale, int64_t n, float sqrt_qmax, void* scratch, size_t scratch_bytes,
                       dlmcq_stream_t stream);

/* dlmcq_minmax_f32 + dlmcq_qparams_from_minmax fused into the same two launches. */
int dlmcq_observe_qparams_f32(const float* x, float* scale, float* offset, int64_t outer,
                              int64_t channels, int64_t inner, int32_t n_bits, int32_t is_signed,
                              int32_t allow_offset, float scale_eps, void* scratch,
                              size_t scratch_bytes, dlmcq_stream_t stream);

/* ---- sub-byte pack / unpack (BASELINE config 5; layout as DLMCQ_CODES_P4) ---- */
int dlmcq_pack_int4(const int8_t* codes, uint8_t* packed, int64_t n, dlmcq_stream_t stream);
int dlmcq_unpack_int4(const uint8_t* packed, int8_t* codes, int64_t n, int32_t is_signed,
                      dlmcq_stream_t stream);

/*
 * Backward of DLMCQ_FORM_QBASE as autograd executes it through modules/base.py:96-102 (closed
 * form: modules/function.py:37-49).  With v = (x-o)/s^, inside = [lo <= v <= hi]:
 *   gx       = inside ? (gy*s^)/s^ : +0                      (bit-exact with autograd)
 *   gscale[c]= g * sum(gy*q - inside*(gy*s^)*(v/s^))          (fp32 tree sum, deterministic)
 * gx may alias gy; gx or gscale may be NULL.  scratch: dlmcq_fq_bwd_scratch_bytes().
 */
size_t dlmcq_fq_bwd_scratch_bytes(int64_t outer, int64_t channels, int64_t inner);
int dlmcq_fake_quant_bwd_f32(const float* x, const float* gy, float* gx, float* gscale,
                             const float* scale, const float* offset, int64_t outer,
                             int64_t channels, int64_t inner, int32_t lo, int32_t hi, float ste_g,
                             void* scratch, size_t scratch_bytes, dlmcq_stream_t stream);
/*
 * The same for the FSPTQ forms, as autograd executes FSPTQuant/base.py:108-109 (DLMCQ_FORM_ZEROPOINT, offset = zero
 * point: u = x/s, a = R(u) + zp, inside = [lo <= a <= hi], gscale = sum gy*((clamp(a) - zp) - inside*u)) and :149-152
 * (DLMCQ_FORM_SYMMETRIC, per-channel weight scale, no offset), and for the RootQ activation form
 * (DLMCQ_FORM_ROOTQ_ACT, RootQ/base.py:106-111: clipped elements pass nothing to x; the scale also collects
 * gy*(hi - lo) from every element clipped above); DLMCQ_FORM_QBASE is the call above.  ste_g is used by QBASE only.
 */
int dlmcq_fake_quant_bwd_form_f32(const float* x, const float* gy, float* gx, float* gscale,
                                  const float* scale, const float* offset, int64_t outer,
                                  int64_t channels, int64_t inner, int32_t lo, int32_t hi, int32_t form,
                                  float ste_g, void* scratch, size_t scratch_bytes,
                                  dlmcq_stream_t stream);

/*
 * RootQ weight forward (RootQ/base.py:146-155 + RootQ/function.py:15-32,58-67), per tensor.
 * `bounds` is a device array {upper, lower}.  The forward value does not depend on alpha (it only
 * shapes the gradient), so alpha is not an input.  y = ((sgn+1)/2 + interval)*delta + lower.
 */
int dlmcq_rootq_weight_f32(const float* w, float* y, const float* bounds, int64_t n, int32_t lo,
                           int32_t hi, dlmcq_stream_t stream);

/*
 * Backward of that transform as autograd runs the reference's op chain (RootQ/base.py:146-155 + function.py:15-32,58-67;
 * the sign and the floor are straight-through): gw[n] (nullable) and g_bounds_alpha = {g_upper, g_lower, g_alpha}.
 * alpha: device scalar.  scratch: dlmcq_rootq_bwd_scratch_bytes(n).  Two launches instead of the ~35 elementwise ones
 * the op chain costs per layer per step.
 */
size_t dlmcq_rootq_bwd_scratch_bytes(int64_t n);
int dlmcq_rootq_weight_bwd_f32(const float* w, const float* gy, float* gw, float* g_bounds_alpha,
                               const float* bounds, const float* alpha, int64_t n, int32_t lo, int32_t hi,
                               void* scratch, size_t scratch_bytes, dlmcq_stream_t stream);

/* ---- weight transforms that precede the path in the few-shot PTQ flow (FSPTQuant.py:65-67) ---- */

/*
 * BatchNorm folding, in place (dlmc/utils/merge_bn.py:85-101): with sd = sqrt(var + var_eps)
 *   weight[k, :] <- (weight[k, :] * gamma[k]) / sd[k];   bias[k] <- (gamma[k]*(bias[k] - mean[k])) / sd[k] + beta[k]
 * The reference adds 1e-7 to the running variance here (not the layer's eps): pass var_eps = 1e-7f.
 * weight is [out_channels, inner]; bias must exist ([out_channels], zeros when the conv had none).
 */
int dlmcq_fold_bn_f32(float* weight, float* bias, const float* gamma, const float* beta,
                      const float* mean, const float* var, int64_t out_channels, int64_t inner,
                      float var_eps, dlmcq_stream_t stream);

/*
 * RepVGG re-parameterisation (model/classification/repvgg.py:92-147): the 3x3 branch, the 1x1 branch and
 * the optional identity branch, each followed by BatchNorm, fused into one 3x3 kernel + bias.
 * bn3 / bn1 / bnid are HOST arrays of 4 device pointers {gamma, beta, running_mean, running_var}; bnid is
 * NULL when the block has no identity branch.  k3 is [K, cin_per_group, 3, 3], k1 is [K, cin_per_group, 1, 1].
 */
int dlmcq_repvgg_fuse_f32(const float* k3, const float* k1, float* out_kernel, float* out_bias,
                          const float* const* bn3, const float* const* bn1, const float* const* bnid,
                          float eps3, float eps1, float epsid, int64_t out_channels,
                          int64_t cin_per_group, dlmcq_stream_t stream);

/* ---- calibration-time estimators (ops.py:71-83, :198-215) ---- */

/*
 * One iteration of the l2norm scale refinement, fused: q = clamp(r((x-o)/(s+1e-7)), lo, hi),
 * new_scale[c] = SUM x*q / SUM (q*q + 1e-7), one read of x.  Order-dependent fp32/fp64 sums: equal to the
 * reference to summation tolerance.  scratch: dlmcq_l2norm_scratch_bytes().
 */
size_t dlmcq_l2norm_scratch_bytes(int64_t outer, int64_t channels, int64_t inner);
int dlmcq_l2norm_step_f32(const float* x, const float* scale, const float* offset, float* new_scale,
                          int64_t outer, int64_t channels, int64_t inner, int32_t lo, int32_t hi,
                          void* scratch, size_t scratch_bytes, dlmcq_stream_t stream);

/* The same refinement run to convergence WITHOUT host round trips: `iterations` iterations are enqueued; each one does
 * nothing once state[0] != 0 (converged: |new - s| / s <= eps per tensor, ||new - s|| / ||s|| <= eps per channel,
 * ops.py:77-81 / :205-210), so the caller reads the flag once per batch instead of syncing on `diff` every step.
 * scale (in/out) [channels]; state = {done, iterations run, unused} (3 floats, zero-initialised by the caller). */
int dlmcq_l2norm_iterate_f32(const float* x, float* scale, const float* offset, float* state, int64_t outer,
                             int64_t channels, int64_t inner, int32_t lo, int32_t hi, int32_t iterations, float eps,
                             void* scratch, size_t scratch_bytes, dlmcq_stream_t stream);

/* One output-aware refinement step (ops.py:96-108 per tensor, :268-284 per output channel), fused: in ONE read of the
 * layer output `out` and of `out_q` = the layer applied to the quantised weight ([outer, channels, inner] both),
 *   s_new = SUM out*out_q / SUM (out_q*out_q + 1e-7),   mse = SUM (out - out_q)^2 / mse_div   (l2_loss, loss.py:22-24)
 * then on device: the reference's best-scale bookkeeping (per tensor: scale <- s_new, best <- scale if mse improved;
 * per channel: best <- OLD scale if mse improved, scale <- s_new) and the convergence flag as above.
 * scale (in/out) [1 or channels]; best_scale [2 x that] (second half: staging); state = {done, iterations, best mse}. */
size_t dlmcq_l2out_scratch_bytes(int64_t outer, int64_t channels, int64_t inner);
int dlmcq_l2out_update_f32(const float* out, const float* out_q, float* scale, float* best_scale, float* state,
                           int64_t outer, int64_t channels, int64_t inner, int32_t per_channel, float mse_div, float eps,
                           void* scratch, size_t scratch_bytes, dlmcq_stream_t stream);

/* quantize_l2loss_tensor (ops.py:36-68, unsigned): the 80-step shrink search in ONE read of x - 80 running squared errors
 * per thread, candidates (scale_i, round(-min_i / scale_i)) derived on device from (vmax, vmin) - and the reference's
 * selection (first loss below 1000, then strictly better ones).  loss_div = numel / shape[1] (l2_loss sums axis 1 and
 * averages the rest).  vmin may be NULL (allow_offset = False).  scale / offset: one float each. */
size_t dlmcq_l2loss_scratch_bytes(int64_t n);
int dlmcq_l2loss_tensor_f32(const float* x, const float* vmax, const float* vmin, float* scale, float* offset, int64_t n,
                            int32_t n_bits, float loss_div, void* scratch, size_t scratch_bytes, dlmcq_stream_t stream);

/* quantize_l2loss_channel (ops.py:169-196) for [rows, inner] rows: one workgroup per row (cached in LDS up to 8192
 * elements), the 80 steps in order with the reference's aliasing quirk (`min_val` is `offset`: an accepted step replaces
 * the minimum by its zero point).  scale / offset [rows]: in = quantize_minmax_channel's result, out = the search's. */
int dlmcq_l2loss_rows_f32(const float* x, float* scale, float* offset, int64_t rows, int64_t inner, int32_t n_bits,
                          dlmcq_stream_t stream);

/*
 * AdaRound weight path of the few-shot PTQ wrapper (FSPTQuant/base.py:69-79,136-141,151-152), per output channel:
 *   y = clamp(floor(w/s_k) + r, lo, hi) * s_k,  r = clamp(sigmoid(alpha)*1.2 - 0.1, 0, 1) when `training`, else [alpha >= 0]
 * and its backward w.r.t. alpha and s_k (w receives no gradient through floor):
 *   g_alpha = gy*s_k*[lo<=q<=hi]*1.2*sig*(1-sig)*[0<=1.2*sig-0.1<=1],   g_scale[k] = sum gy*clamp(q, lo, hi)
 * w / alpha / y / gy / g_alpha are [out_channels, inner]; g_alpha or g_scale may be NULL.
 */
int dlmcq_adaround_weight_f32(const float* w, const float* alpha, const float* scale, float* y,
                              int64_t out_channels, int64_t inner, int32_t lo, int32_t hi,
                              int32_t training, dlmcq_stream_t stream);
int dlmcq_adaround_weight_bwd_f32(const float* w, const float* alpha, const float* scale,
                                  const float* gy, float* g_alpha, float* g_scale,
                                  int64_t out_channels, int64_t inner, int32_t lo, int32_t hi,
                                  dlmcq_stream_t stream);

/* ---- fused int8-dequant x GEMM convolution / linear on the matrix cores ---- */

/*
 * Weights fp32 KCRS -> int8 codes in KRSC order (the reduction order of the implicit GEMM) with the
 * SYMMETRIC form (q = clamp(R(w / scale[k]), lo, hi), FSPTQuant/base.py:149-152), plus wsum[k] = sum of the
 * codes of output channel k (needed for the activation zero-point / uint8 shift correction).
 */
int dlmcq_quantize_weight_krsc_i8(const float* w, int8_t* wq, int32_t* wsum, const float* scale,
                                  int64_t K, int64_t C, int64_t R, int64_t S, int32_t lo, int32_t hi,
                                  dlmcq_stream_t stream);

/*
 * out[n,p,q,k] = in_scale * w_scale[k] * SUM_{r,s,c} (x[n,h,w,c] - zp) * wq[k,r,s,c]  + bias[k]
 * i.e. F.conv2d(x', w', bias) of modules/conv.py:18-19 on the fake-quantised operands, with exact int32
 * accumulation on v_mfma_i32_32x32x32_i8 (groups = 1).  Zero padding is padding of x' = 0, i.e. of x = zp.
 *   x        integer codes, NHWC (uint8 when x_is_unsigned, else int8), 16-byte aligned, C % 64 == 0
 *   w, wsum  from dlmcq_quantize_weight_krsc_i8
 *   out      fp32 NHWC [N, P, Q, K];  bias [K] or NULL
 *   in_scale, in_zero_point: device scalars (zero point must hold an integer value; NULL = 0); w_scale [K]
 * A linear layer is the case H = W = R = S = 1.
 */
int dlmcq_conv2d_i8_nhwc_f32(const void* x, const int8_t* w, float* out, const float* bias,
                             const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                             const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                             int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                             int32_t x_is_unsigned, dlmcq_stream_t stream);

/*
 * The same contraction with the work that FOLLOWS a quantised layer in a frozen (inference) network folded
 * into the epilogue, so the fp32 output, the ReLU pass, the residual add and the consumer's quantise pass stop
 * being separate trips through HBM.  In order, per output element v = in_scale*w_scale[k]*SUM + bias[k]:
 *   residual != NULL : v = v + residual[n,p,q,k]          (fp32 NHWC, the output's shape: the block's shortcut)
 *   relu != 0        : v = v < 0 ? 0 : v                  (torch.relu; NaN stays NaN)
 *   out != NULL      : out[n,p,q,k] = v
 *   codes != NULL    : codes[n,p,q,k] = the consumer's activation code of v: forms EMULATE / QBASE /
 *                      ZEROPOINT / SYMMETRIC of dlmcq_fake_quant_f32 with (q_scale, q_zero_point, q_lo, q_hi,
 *                      q_ste_g) - the same arithmetic in the same order, hence the bytes that
 *                      dlmcq_fake_quant_f32(..., DLMCQ_CODES_I8) would write for the stored fp32 tensor.
 *                      uint8 when q_lo >= 0, else int8; -128 <= q_lo <= q_hi <= 255, q_hi - q_lo <= 255.
 * At least one of out / codes must be given.  Replaces, for a frozen network, the chain
 * F.conv2d -> (+ identity) -> ReLU -> next FSPTQBase.forward activation branch (FSPTQuant/base.py:108-109).
 */
int dlmcq_conv2d_i8_nhwc_fused(const void* x, const int8_t* w, float* out, const float* bias,
                               const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                               const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                               int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                               int32_t x_is_unsigned, const float* residual, int32_t relu, void* codes,
                               const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                               int32_t q_form, float q_ste_g, dlmcq_stream_t stream);

/* dlmcq_conv2d_i8_nhwc_fused (out != NULL) that also OBSERVES its fp32 output - the calibrating first batch (round 5): the min/max pass
 * of the consumer's activation observer (modules/base.py:82-94 / FSPTQuant/base.py:99-103 -> ops.py:20-34: one more read of the tensor) comes
 * out of the epilogue of the launch that writes the tensor.  Every workgroup of the tiled kernel stores the (max, min, unsigned max of the
 * bits of |v| - the NaN detector) of the values it wrote as entry i of three planes of `*partials_count` floats each in `partials`
 * (observer.hip's partial layout; capacity >= 3 * dlmcq_conv2d_i8_observed_partials(N P Q, K) floats, else DLMCQ_ESCRATCH);
 * dlmcq_minmax_finalize_f32 reduces them to what dlmcq_minmax_f32 gives for the tensor - max and min are exact, so bit for bit.
 * `*partials_count` (written on the host before the call returns) is 0 when the dispatch hands the call to a kernel without the
 * observing epilogue (the block-end pointwise kernel): observe the tensor with dlmcq_minmax_f32 then. */
size_t dlmcq_conv2d_i8_observed_partials(int64_t M, int64_t K);
int dlmcq_conv2d_i8_nhwc_fused_observed(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                        const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                                        int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride,
                                        int32_t pad, int32_t dilation, int32_t x_is_unsigned, const float* residual,
                                        int32_t relu, void* codes, const float* q_scale, const float* q_zero_point,
                                        int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g, float* partials,
                                        int64_t partials_capacity, int64_t* partials_count, dlmcq_stream_t stream);

/* dlmcq_conv2d_i8_nhwc_fused for ASYMMETRIC per-output-channel weights (ops.py:129-136, unsigned `minmax_channel`:
 * w' = qw * s_w[k] + o_w[k], the offset being the channel minimum; BASELINE configs[4], W4A8): the convolution gains the
 * term  o_w[k] * SUM x'  over the receptive field, which the kernel evaluates from a per-pixel sum of the activation
 * codes (v_dot4_i32_i8 on the operand fragments it multiplies anyway):
 *     out = s_in * ( s_w[k] * (SUM q'*qw + (shift - zp) * SUM qw)  +  o_w[k] * SUM (q - zp) ) + bias[k].
 * `w` holds int8 codes; codes above 127 (8-bit unsigned weights) are stored minus 128 with w_offset += 128 * s_w[k]. */
int dlmcq_conv2d_i8_nhwc_asym(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                              const float* in_scale, const float* in_zero_point, const float* w_scale, const float* w_offset,
                              int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride,
                              int32_t pad, int32_t dilation, int32_t x_is_unsigned, const float* residual, int32_t relu,
                              void* codes, const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                              int32_t q_form, float q_ste_g, dlmcq_stream_t stream);

/* Depthwise 3x3 convolution (groups = channels; the MobileOne / MobileNet unit, modules/conv.py:13-19 with `groups`) on
 * activation codes, HBM-bound (1 byte in, 1 byte out): plain vector arithmetic, no matrix cores.  x: NHWC codes (uint8 if
 * x_is_unsigned), C % 4 == 0; w: int8 codes [R*S][C] (tap-major); per-channel w_scale and optional w_offset (asymmetric
 * weights as above), bias; zero padding (x' = 0).  Epilogue as dlmcq_conv2d_i8_nhwc_fused: ReLU, fp32 NHWC out and / or the
 * consumer's codes.   out = s_in * ( s_w[c] * SUM (q - zp) * qw  +  o_w[c] * SUM (q - zp) ) + bias[c]   (exact integer sums). */
int dlmcq_conv2d_dw_i8_nhwc(const void* x, const int8_t* w, float* out, const float* bias, const float* in_scale,
                            const float* in_zero_point, const float* w_scale, const float* w_offset, int64_t N, int64_t H,
                            int64_t W, int64_t C, int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t x_is_unsigned,
                            int32_t relu, void* codes, const float* q_scale, const float* q_zero_point, int32_t q_lo,
                            int32_t q_hi, int32_t q_form, float q_ste_g, dlmcq_stream_t stream);

/*
 * A MobileOne / MobileNet unit as ONE launch (round 4): depthwise 3x3 / stride 1 / pad 1 (+ ReLU + quantiser) followed by the
 * pointwise 1x1 convolution that reads nothing but its codes (+ ReLU + the consumer's quantiser): dlmcq_conv2d_dw_i8_nhwc followed
 * by dlmcq_conv2d_i8_nhwc_asym / _fused with codes-only output, same integers, same fp32 chains, same quantisers - bit-identical -
 * but the wide code tensor between the two layers stays in LDS (csrc/conv_dwpw_i8.hip).
 *   dlmcq_dwpw_pack_table   the depthwise layer's constants (w: int8 codes [9][C] tap-major, per-channel w_scale, optional w_offset
 *                           and bias, the layer's input scale / zero point) -> `table`: C * 32 bytes, 16-byte aligned, C % 64 == 0.
 *                           Made once per (layer, input scale): what conv_dw3_i8_kernel builds per workgroup.
 *   dlmcq_conv2d_dwpw_i8_nhwc   x: NHWC codes [N][H][W][C] (uint8 if x_is_unsigned), W <= 61; (q_*): the depthwise output's
 *                           quantiser = the pointwise layer's input quantiser, range [0, 255]; pw_in_scale: the scale the pointwise
 *                           layer dequantises with (= q_scale, or its grad_scale value for QBase); w: int8 [K][C], K in {128, 192,
 *                           512}, with wsum / w_scale / optional w_offset / bias [K]; codes: [N][H][W][K] under (q2_*).
 */
int dlmcq_dwpw_pack_table(const int8_t* w, const float* bias, const float* in_scale, const float* in_zero_point,
                          const float* w_scale, const float* w_offset, int64_t C, int32_t x_is_unsigned, void* table,
                          dlmcq_stream_t stream);
int dlmcq_conv2d_dwpw_i8_nhwc(const void* x, const void* dw_table, int32_t dw_asym, int32_t dw_bias, int32_t dw_relu,
                              const float* in_zero_point, int64_t N, int64_t H, int64_t W, int64_t C, int32_t x_is_unsigned,
                              const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form,
                              float q_ste_g, const int8_t* w, const float* bias, const int32_t* wsum, const float* pw_in_scale,
                              const float* w_scale, const float* w_offset, int64_t K, int32_t relu, void* codes,
                              const float* q2_scale, const float* q2_zero_point, int32_t q2_lo, int32_t q2_hi, int32_t q2_form,
                              float q2_ste_g, dlmcq_stream_t stream);

/*
 * Two convolutions into ONE output: out = conv(x, w) + conv(x2, w2), each dequantised with its own scales and bias
 * and summed in fp32 (one addition, as `out += identity` does it in a residual block whose shortcut is a
 * convolution), then the epilogue of dlmcq_conv2d_i8_nhwc_fused.  Both pairs must give the same [N, P, Q, K] output;
 * the second pair has its own input size, channel count, filter size, stride, padding and dilation (the 1x1/2
 * downsample next to the block's last 1x1).  Neither addend is written to memory.
 */
int dlmcq_conv2d_i8_nhwc_dual(const void* x, const int8_t* w, float* out, const float* bias,
                              const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                              const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                              int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                              int32_t x_is_unsigned, const void* x2, const int8_t* w2, const float* bias2,
                              const int32_t* wsum2, const float* in_scale2, const float* in_zero_point2,
                              const float* w_scale2, int64_t H2, int64_t W2, int64_t C2, int64_t R2,
                              int64_t S2, int32_t stride2, int32_t pad2, int32_t dilation2,
                              int32_t x2_is_unsigned, int32_t relu, void* codes, const float* q_scale,
                              const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form,
                              float q_ste_g, dlmcq_stream_t stream);

/*
 * A residual block's LAST 1x1 convolution and the NEXT block's FIRST 1x1 convolution as one kernel
 * (csrc/conv_chain_i8.hip):
 *     v      = conv1x1(x, w) + bias + residual;  ReLU (if relu);  out = v (optional);  codes = Q(v) (optional)
 *     codes2 = Q2( ReLU?( conv1x1( Q(v), w2 ) + bias2 ) )
 * i.e. dlmcq_conv2d_i8_nhwc_fused on [M, C] -> [M, K] (stride 1, no padding) followed by the same call on its `codes`
 * with [K2, K] weights, `relu2` and the second consumer's quantiser Q2 - bit-identical to those two calls - but the
 * [M, K] code tensor stays in LDS: it is not read back, and not written unless `codes` is given.  M = N*H*W pixels;
 * x: [M][C] codes; w: [K][C]; residual (required) / out: fp32 [M][K]; w2: [K2][K]; codes2: [M][K2].
 * Q must be an unsigned 8-bit quantiser (q_lo = 0, q_hi = 255: the second reduction reads uint8 codes).
 * Supported: (C, K2) in {(64,64), (64,128), (128,128), (128,256), (256,256)}, K % 64 == 0, M*K*4 < 2^31 - 64 Ki
 * (anything else: DLMCQ_EINVAL / DLMCQ_ERANGE; callers fall back to the two separate calls).
 * rows_per_tile: pixels per workgroup, 1..64; <= 0: 64.  x, w, w2, residual, out, codes and codes2 must be 16-byte aligned
 * (DLMCQ_EALIGN otherwise).
 */
int dlmcq_conv2d_i8_nhwc_chain(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                               const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t M,
                               int64_t C, int64_t K, int32_t x_is_unsigned, const float* residual, int32_t relu,
                               void* codes, const float* q_scale, const float* q_zero_point, int32_t q_lo,
                               int32_t q_hi, int32_t q_form, float q_ste_g, const int8_t* w2, const float* bias2,
                               const int32_t* wsum2, const float* w_scale2, int64_t K2, int32_t relu2, void* codes2,
                               const float* q2_scale, const float* q2_zero_point, int32_t q2_lo, int32_t q2_hi,
                               int32_t q2_form, float q2_ste_g, int32_t rows_per_tile, dlmcq_stream_t stream);

/*
 * dlmcq_conv2d_i8_nhwc_chain for a block whose shortcut is itself a 1x1 convolution (the first block of a stage): the
 * first layer is dlmcq_conv2d_i8_nhwc_dual restricted to two 1x1 convolutions,
 *     v = conv1x1(x, w) + bias  +  conv1x1(x2 sampled at stride2, w2) + bias2;  ReLU;  out = v (optional);  codes = Q(v) (optional)
 *     codes3 = Q3( ReLU?( conv1x1( Q(v), w3 ) + bias3 ) )
 * bit-identical to dlmcq_conv2d_i8_nhwc_dual followed by dlmcq_conv2d_i8_nhwc_fused on its codes.  x: [N][H][W][C] codes
 * (stride 1); x2: [N][H2][W2][C2] codes with (H2 - 1) / stride2 + 1 == H (likewise W); w: [K][C]; w2: [K][C2]; w3: [K3][K].
 * Supported: (C, C2, K3) in {(64, 64, 64), (128, 256, 128)}; K % 64 == 0; N*H*W*K*4 < 2^31 - 64 Ki.
 */
int dlmcq_conv2d_i8_nhwc_dual_chain(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                    const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                                    int64_t H, int64_t W, int64_t C, int64_t K, int32_t x_is_unsigned, const void* x2,
                                    const int8_t* w2, const float* bias2, const int32_t* wsum2, const float* in_scale2,
                                    const float* in_zero_point2, const float* w_scale2, int64_t H2, int64_t W2, int64_t C2,
                                    int32_t stride2, int32_t x2_is_unsigned, int32_t relu, void* codes, const float* q_scale,
                                    const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g,
                                    const int8_t* w3, const float* bias3, const int32_t* wsum3, const float* w_scale3,
                                    int64_t K3, int32_t relu3, void* codes3, const float* q3_scale, const float* q3_zero_point,
                                    int32_t q3_lo, int32_t q3_hi, int32_t q3_form, float q3_ste_g, int32_t rows_per_tile,
                                    dlmcq_stream_t stream);

/* ---- the 3-channel first layer and its max-pool, in the integer-code domain (csrc/conv_stem_i8.hip) ---- */

/*
 * Quantise an image batch (C <= 4 channels, any memory format: element strides given) to activation codes in a
 * zero-point-PADDED NHWC buffer with 4 bytes per pixel: out[N][H+2*pad][W+2*pad][4], border = the code of
 * x' = 0 (zero padding of the fake-quantised image, modules/conv.py:18-19), bytes c >= C unspecified.
 * Forms EMULATE / QBASE / ZEROPOINT / SYMMETRIC as in dlmcq_fake_quant_f32 (same arithmetic, same codes).
 * The consumer over-reads up to 32 bytes past the last pixel: allocate N*(H+2*pad)*(W+2*pad)*4 + 32 bytes.
 */
int dlmcq_quantize_pad_nhwc4(const float* x, void* out, const float* scale, const float* zero_point, int64_t N,
                             int64_t C, int64_t H, int64_t W, int64_t stride_n, int64_t stride_c, int64_t stride_h,
                             int64_t stride_w, int32_t pad, int32_t lo, int32_t hi, int32_t form, float ste_g,
                             dlmcq_stream_t stream);

/* Weights fp32 KCRS (C <= 4, R <= 7, S <= 8) -> int8 [K][R][8][4] zero-filled, SYMMETRIC form, + wsum[K]. */
int dlmcq_quantize_weight_stem_i8(const float* w, int8_t* wq, int32_t* wsum, const float* scale, int64_t K,
                                  int64_t C, int64_t R, int64_t S, int32_t lo, int32_t hi, dlmcq_stream_t stream);

/*
 * The convolution of modules/conv.py:18-19 for such a layer: out[n,p,q,k] = in_scale*w_scale[k]*SUM (x - zp)*wq
 * + bias[k] over the padded codes (P = (Hp - R)/stride + 1, Q = (Wp - S)/stride + 1, dilation 1), then the
 * epilogue of dlmcq_conv2d_i8_nhwc_fused (ReLU, fp32 NHWC output and / or the consumer's codes).  K % 4 == 0.
 */
int dlmcq_conv2d_i8_stem_fused(const void* xpad, const int8_t* w, float* out, const float* bias,
                               const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                               const float* w_scale, int64_t N, int64_t Hp, int64_t Wp, int64_t K, int64_t R,
                               int64_t S, int32_t stride, int32_t x_is_unsigned, int32_t relu, void* codes,
                               const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                               int32_t q_form, float q_ste_g, dlmcq_stream_t stream);

/* dlmcq_conv2d_i8_stem_fused for ASYMMETRIC per-output-channel weights (as dlmcq_conv2d_i8_nhwc_asym: w' = qw * s_w[k] + o_w[k]):
 * the term o_w[k] * SUM x' comes from a per-pixel sum of the operand codes over the R x S taps and the C real channels
 * (C <= 4: the 4th byte of a pixel and the taps beyond S are excluded by a 0/1 byte mask). */
int dlmcq_conv2d_i8_stem_asym(const void* xpad, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                              const float* in_scale, const float* in_zero_point, const float* w_scale, const float* w_offset,
                              int64_t C, int64_t N, int64_t Hp, int64_t Wp, int64_t K, int64_t R, int64_t S, int32_t stride,
                              int32_t x_is_unsigned, int32_t relu, void* codes, const float* q_scale, const float* q_zero_point,
                              int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g, dlmcq_stream_t stream);

/*
 * dlmcq_conv2d_i8_stem_fused followed by ReLU (if asked) and nn.MaxPool2d(3, 2, 1) in ONE kernel, K <= 64: out / codes
 * are the POOLED tensor [N, (P+1)/2.., (Q+1)/2.., K].  The pool runs on the fp32 values (the reference's order: ReLU,
 * max-pool, fake-quant; NaN propagates as in torch), so only the pooled quarter of the outputs is quantised and
 * nothing un-pooled is ever written.
 */
int dlmcq_conv2d_i8_stem_pool_fused(const void* xpad, const int8_t* w, float* out, const float* bias,
                                    const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                    const float* w_scale, int64_t N, int64_t Hp, int64_t Wp, int64_t K, int64_t R,
                                    int64_t S, int32_t stride, int32_t x_is_unsigned, int32_t relu, void* codes,
                                    const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                                    int32_t q_form, float q_ste_g, dlmcq_stream_t stream);

/*
 * nn.MaxPool2d (square window, dilation 1, floor mode) on NHWC activation codes.  The quantisers of this library
 * are monotone, so  code(maxpool(v)) == maxpool(code(v)):  pooling the 1-byte codes replaces pooling the fp32
 * tensor and quantising the result.  C % 4 == 0; padding never wins (torch pads with -inf).
 */
int dlmcq_maxpool_codes_nhwc(const void* x, void* y, int64_t N, int64_t H, int64_t W, int64_t C, int32_t kernel,
                             int32_t stride, int32_t pad, int32_t x_is_unsigned, dlmcq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DLMCQ_H */
