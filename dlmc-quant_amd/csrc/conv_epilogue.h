// Epilogue shared by the int8 convolution kernels (conv_i8.hip, conv_stem_i8.hip).  Not part of the ABI.
#pragma once

#include "dlmcq_internal.h"

namespace dlmcq {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// What an int8 kernel does with a finished output element besides storing it (all optional): add a residual
// tensor of the output's shape, apply ReLU, and emit the NEXT layer's activation codes - the consumer's own
// fake-quant (its scale / zero point / range / form) evaluated on the value still in a register, so the fp32
// tensor, the ReLU pass and the consumer's quantise pass never touch HBM.  Same arithmetic, same order, as the
// separate kernels (fq_one), hence bit-identical codes.
struct ConvEpi {
  const float* w_off;      // asymmetric per-channel weights w' = qw * s_w[k] + w_off[k] (ops.py:129-136): the extra term
                           // w_off[k] * SUM x' of the convolution, from a per-pixel sum of the activation codes (conv_i8.hip)
  const float* residual;
  uint8_t* codes;
  const float* q_scale;
  const float* q_zp;
  float q_lo, q_hi, q_g;
  int q_form;
  int relu;
  uint32_t q_xor;          // 0x80808080 when the codes are stored as int8 `code - 128` (DLMCQ_EMIT_SHIFT128), else 0
  uint32_t ctl;            // host side only: DLMCQ_FORCE_TILED | DLMCQ_ROUTE_ONLY | DLMCQ_PIPELINED as passed in `q_form`
  // observer partials of the fp32 OUTPUT (dlmcq_conv2d_i8_nhwc_fused_observed; the tiled kernel only): every workgroup writes the
  // (max, min, max of |x|'s bits) of the values it stored at entry blockIdx.x of three planes of `mm_np` floats each - observer.hip's
  // partial layout, reduced by dlmcq_minmax_finalize_f32.  The calibrating first batch: the consumer's min/max pass (one read of the
  // tensor: modules/base.py:82-94 -> ops.py:20-34) comes out of the launch that produces the tensor.
  float* mm;
  int mm_np;
};

// `q_form` argument of an entry point -> (form, shifted-emission flag); false = invalid
static inline bool epi_set_form(ConvEpi& ep, int32_t q_form, int32_t q_lo, int32_t q_hi) {
  const bool shifted = (q_form & DLMCQ_EMIT_SHIFT128) != 0;
  constexpr int32_t CTL = DLMCQ_FORCE_TILED | DLMCQ_ROUTE_ONLY | DLMCQ_PIPELINED | DLMCQ_FP32_IN_CHUNK_MAJOR | DLMCQ_FP32_OUT_CHUNK_MAJOR;
  ep.ctl = (uint32_t)q_form & CTL;
  ep.q_form = q_form & ~(DLMCQ_EMIT_SHIFT128 | CTL);
  ep.q_xor = shifted ? 0x80808080u : 0u;
  return ep.q_form >= DLMCQ_FORM_EMULATE && ep.q_form <= DLMCQ_FORM_SYMMETRIC && (!shifted || (q_lo >= 0 && q_hi <= 255));
}

// the quantiser of every post-ReLU tensor in the frozen plans: unsigned byte range, no zero point (EpiQuant::code4n<N, true>)
__host__ __device__ static inline bool epi_plain_q(const ConvEpi& ep) { return !ep.q_zp && ep.q_lo == 0.0f && ep.q_hi == 255.0f; }      // (the quantiser alone)
__host__ __device__ static inline bool epi_plain(const ConvEpi& ep) { return ep.codes && epi_plain_q(ep); }


// The one rounding chain of every int8 kernel: exact integer sum -> fp32, then ONE fused multiply-add with the layer's
// (s_in * s_w[k]) and bias[k].  (Rounds 1-2a multiplied and added separately; fused is one VALU instruction fewer per element
// in epilogues that are bound by exactly that, and one rounding closer to the real-valued result.)
__device__ __forceinline__ float dequant1(int sum, float mult, float bias) { return __builtin_fmaf((float)sum, mult, bias); }
// ... on a pair: v_pk_fma_f32 (one fused multiply-add per element, the same rounding)
__device__ __forceinline__ f32x2 pk_fma(const f32x2& a, const f32x2& b, const f32x2& c) { return __builtin_elementwise_fma(a, b, c); }

// torch.relu on four values (relu_nan: NaN and -0 pass) as 4 compares into 4 different SGPR pairs followed by 4 selects: the
// compiler's rendering serialises on vcc with wait states after every compare
__device__ __forceinline__ f32x4 relu4_nan(const f32x4& y) {
  float o0, o1, o2, o3;
  uint64_t m0, m1, m2, m3;
  asm("v_cmp_ngt_f32_e64 %[m0], 0, %[y0]\n\t"
      "v_cmp_ngt_f32_e64 %[m1], 0, %[y1]\n\t"
      "v_cmp_ngt_f32_e64 %[m2], 0, %[y2]\n\t"
      "v_cmp_ngt_f32_e64 %[m3], 0, %[y3]\n\t"
      "v_cndmask_b32_e64 %[o0], 0, %[y0], %[m0]\n\t"
      "v_cndmask_b32_e64 %[o1], 0, %[y1], %[m1]\n\t"
      "v_cndmask_b32_e64 %[o2], 0, %[y2], %[m2]\n\t"
      "v_cndmask_b32_e64 %[o3], 0, %[y3], %[m3]"
      : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2),
        [m3] "=&s"(m3)
      : [y0] "v"(y.x), [y1] "v"(y.y), [y2] "v"(y.z), [y3] "v"(y.w));
  return f32x4{o0, o1, o2, o3};
}

struct EpiQuant {   // the consumer's constants, resolved once per thread
  // |t - rint(t)| must stay below this for the fast path to stand (see code4_fast): 0.5 minus the bound on |t - (d + zadd)| at
  // the largest code that does not saturate.  With A = max(|lo|, |hi|) + 0.5: |q| <= A + |zadd|, |t| <= A + 2^-13, so the bound is
  // 2^-24 (2 (A + |zadd|) + A) - for an unsigned 8-bit quantiser with zero point 0 (every post-ReLU tensor) 0.37 * 2^-13, at the
  // extremes the ABI allows (|zadd| = 256, A = 255.5) 1.75 * 2^-14 < 2^-13.
  float tie_thr;
  float dv, rdv, of, zadd, lo, hi;
  float lo_fast;    // lower clamp of the fast path: `lo`, or max(lo, code of 0) when the ReLU is folded into the quantiser
  int form;
  uint32_t xemit;   // DLMCQ_EMIT_SHIFT128: every emitted byte ^ 0x80
  bool sgn, fold;
  // fold_relu: the caller quantises relu(v) but passes v: every form is monotone, so code(relu(v)) = max(code(v), code(0))
  // for a finite v - one clamp bound instead of a compare + select per element (NaN takes the exact path, which rectifies)
  __device__ __forceinline__ EpiQuant(const ConvEpi& ep, bool fold_relu = false)
      : tie_thr(0.5f - 0x1p-13f), dv(1.0f), rdv(1.0f), of(0.0f), zadd(0.0f), lo(ep.q_lo), hi(ep.q_hi), lo_fast(ep.q_lo), form(ep.q_form),
        xemit(ep.q_xor), sgn(ep.q_lo < 0.0f), fold(fold_relu) {
    if (!ep.codes) return;
    const float s = ep.q_scale[0];
    const float z = ep.q_zp ? ep.q_zp[0] : 0.0f;
    dv = form == DLMCQ_FORM_EMULATE ? s + 1e-7f : (form == DLMCQ_FORM_QBASE ? ste_scale(s, ep.q_g) : s);
    // EMULATE / QBASE divide (v - offset); ZEROPOINT adds the zero point after rounding; v - 0 is v, r + 0 is r
    of = (form == DLMCQ_FORM_EMULATE || form == DLMCQ_FORM_QBASE) ? z : 0.0f;
    zadd = form == DLMCQ_FORM_ZEROPOINT ? z : 0.0f;
    // the fast path below is proven for a well-scaled divisor and a byte-sized zero point; anything else (and NaN)
    // makes rdv NaN, which routes every element to the exact division
    const bool tame = __builtin_fabsf(dv) >= 0x1p-100f && __builtin_fabsf(dv) <= 0x1p100f && __builtin_fabsf(zadd) <= 256.0f &&
                      zadd == __builtin_rintf(zadd);
    rdv = tame ? 1.0f / dv : __builtin_nanf("");
    {
      const float A = (__builtin_fabsf(lo) > __builtin_fabsf(hi) ? __builtin_fabsf(lo) : __builtin_fabsf(hi)) + 0.5f;
      tie_thr = 0.5f - (3.0f * A + 2.0f * __builtin_fabsf(zadd) + 4.0f) * 0x1p-24f;     // (+ 4: slack for the roundings of this line)
    }
    if (fold) {
      const float q0 = exact_q(0.0f);
      lo_fast = q0 > lo ? q0 : lo;     // (a NaN code of 0 - NaN scale - keeps lo; rdv is NaN then and nothing takes the fast path)
    }
  }
  // the reference arithmetic itself, form by form (fq_one): what the fast path below must reproduce, and what decides
  // the elements it cannot vouch for.  (Not interchangeable with the reduced formula for +-inf: the STE identity
  // R(v) = (rint(v) - v) + v turns an infinite quotient into NaN -> code 0, where EMULATE's plain rint saturates.)
  __device__ __forceinline__ float exact_q(float v) const {
    if (form == DLMCQ_FORM_EMULATE) return clamp_nan(__builtin_rintf((v - of) / dv), lo, hi);
    if (form == DLMCQ_FORM_QBASE) return ste_round(clamp_nan((v - of) / dv, lo, hi));
    if (form == DLMCQ_FORM_ZEROPOINT) return clamp_nan(ste_round(v / dv) + zadd, lo, hi);
    return clamp_nan(ste_round(v / dv), lo, hi);
  }
  __device__ __forceinline__ uint32_t exact(float v) const { return (uint32_t)(code_of(exact_q(v)) & 0xff) ^ (xemit & 0xffu); }
  // The rare path of code4: `wfast` holds the fast path's bytes, of which at least one failed its tie test in this lane.
  // Only the failing ELEMENTS are redone (the test is repeated per element; an element that passes keeps its fast byte,
  // which the argument below vouches for element by element), and the exact division exists once, in a loop over the set
  // bits of the lane's failure mask: a wave that enters here typically has ONE failing element in one lane, so it pays one
  // division instead of the four per lane of rounds 1-2a (which were ~5 of the ~23 vector instructions an output element
  // cost on average).
  __device__ __forceinline__ uint32_t exact4(const f32x4& v, uint32_t wfast) const {
    const float t0 = __builtin_fmaf(v.x - of, rdv, zadd), t1 = __builtin_fmaf(v.y - of, rdv, zadd);
    const float t2 = __builtin_fmaf(v.z - of, rdv, zadd), t3 = __builtin_fmaf(v.w - of, rdv, zadd);
    // Second look, by magnitude: |t - (d + zadd)| <= 2^-23 |q| + 2^-24 |t| with |q| <= |t| + 256.5, i.e. < 2^-24 (3 |t| + 514), which
    // is below 2^-22 (|t| + 256) for every finite t - no saturation argument needed.  The fast path's fixed margin (2^-13) is
    // that bound at |t| = 768; an element of ordinary size sits ~16x closer to a tie before it really needs the division,
    // so most entries into this branch end here, with the fast bytes confirmed.
    auto near_tie = [](float t) {
      return !(__builtin_fabsf(t - __builtin_rintf(t)) < 0.5f - (__builtin_fabsf(t) + 256.0f) * 0x1p-22f);
    };
    uint32_t m = (near_tie(t0) ? 1u : 0u) | (near_tie(t1) ? 2u : 0u) | (near_tie(t2) ? 4u : 0u) | (near_tie(t3) ? 8u : 0u);
    uint32_t w = wfast;
#pragma unroll 1
    while (m) {
      const int e = __builtin_ctz(m);
      m &= m - 1u;
      float u = e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
      if (fold) u = relu_nan(u);
      const int sh = 8 * e;
      w = (w & ~(0xffu << sh)) | (exact(u) << sh);
    }
    return w;
  }
  // For a FINITE quotient d = fl(u / dv), u = v - of, all four forms reduce to  code = clamp(rint(d) + zadd, lo, hi)
  // (rint(clamp(d)) = clamp(rint(d)) for integral bounds; the STE identity (r - d) + d returns r exactly).
  // A correctly rounded division costs ~25 VALU operations per element - more than everything else in the epilogue -
  // so d is replaced by t = fl(u * fl(1/dv) + zadd) (one fma).  With q = u/dv: |u fl(1/dv) - q| <= 2^-24 |q|, the fma rounds by
  // <= 2^-24 |t|, and |d - q| <= 2^-24 |q|, so |t - (d + zadd)| <= 2^-23 |q| + 2^-24 |t|.  A code that does not saturate has
  // |t| <= A = max(|lo|, |hi|) + 0.5 and |q| <= A + |zadd|: the difference is below 2^-24 (3 A + 2 |zadd|) =: M (`tie_thr` = 0.5 - M;
  // at most 1.75 * 2^-14), so unless t lies within M of a rounding tie (x.5), rint(t) = rint(d) + zadd.  Beyond |t| = A the
  // difference may exceed M, but it stays below 0.2 up to |t| = 2^20 and both values are then past the same clamp bound by
  // more than 0.3: both saturate to it (and further out a fortiori).  Ties that close, infinities and NaNs (about one element
  // in 10 000 for an unsigned byte with zero point 0) go to exact4: a second, magnitude-aware look, then the exact division.
  // Bit-identical codes at ~6 operations per element: the
  // block-end layers are bound by the VALU instructions of their epilogue (tools/chain_trace.py), so the arithmetic
  // is written on pairs (v_pk_add_f32 / v_pk_mul_f32: two elements per instruction, same roundings), and a non-finite
  // t needs no separate test: t - rint(t) is NaN then, and NaN is "not below the threshold".
  // the fast path alone, branch-free: the packed codes and whether any of the four needs the exact division instead.
  // Callers with several independent quads evaluate them all, OR the flags and branch ONCE: a branch per quad chains
  // the quads one behind the other (each ~25 dependent instructions long), which is what bounded the block-end layers.
  // The zero point rides on the multiply: t = fma(u, 1/dv, zadd) (zadd is integral, checked above), q = rint(t).  t differs
  // from d + zadd by less than M in the range that does not saturate (above), so the tie test is applied to t itself: 6 instructions per element (fma, rint, sub, compare, clamp, pack)
  // when the form has no offset to subtract first (OFZ).
  // PLAIN (the caller has checked epi_plain on the host): unsigned byte range [0, 255], no zero point - of = zadd = 0, lo_fast = 0
  // with or without the folded ReLU, no negative codes.  The clamp is then the saturation of v_cvt_pk_u8_f32 itself (negative -> 0,
  // beyond 255 -> 255; a NaN has failed the tie test): 5 instructions per element (mul, rint, sub, compare, pack), same bytes.
  template <bool OFZ, bool PLAIN = false>
  __device__ __forceinline__ uint32_t code4_fast(const f32x4& v, bool& unsure) const {
    if constexpr (PLAIN) {
      const float t0 = v.x * rdv, t1 = v.y * rdv, t2 = v.z * rdv, t3 = v.w * rdv;      // (= fma(v, rdv, 0) up to the sign of a zero, which nothing below sees)
      const float r0 = __builtin_rintf(t0), r1 = __builtin_rintf(t1), r2 = __builtin_rintf(t2), r3 = __builtin_rintf(t3);
      const float thr = tie_thr;
      unsure = !(__builtin_fabsf(t0 - r0) < thr) | !(__builtin_fabsf(t1 - r1) < thr) | !(__builtin_fabsf(t2 - r2) < thr) |
               !(__builtin_fabsf(t3 - r3) < thr);
      uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(r0, 0, 0u);
      w = __builtin_amdgcn_cvt_pk_u8_f32(r1, 1, w);
      w = __builtin_amdgcn_cvt_pk_u8_f32(r2, 2, w);
      return __builtin_amdgcn_cvt_pk_u8_f32(r3, 3, w) ^ xemit;
    }
    const float u0 = OFZ ? v.x : v.x - of, u1 = OFZ ? v.y : v.y - of, u2 = OFZ ? v.z : v.z - of, u3 = OFZ ? v.w : v.w - of;
    const float t0 = __builtin_fmaf(u0, rdv, zadd), t1 = __builtin_fmaf(u1, rdv, zadd), t2 = __builtin_fmaf(u2, rdv, zadd),
                t3 = __builtin_fmaf(u3, rdv, zadd);
    const float r0 = __builtin_rintf(t0), r1 = __builtin_rintf(t1), r2 = __builtin_rintf(t2), r3 = __builtin_rintf(t3);
    const float thr = tie_thr;
    unsure = !(__builtin_fabsf(t0 - r0) < thr) | !(__builtin_fabsf(t1 - r1) < thr) | !(__builtin_fabsf(t2 - r2) < thr) |
             !(__builtin_fabsf(t3 - r3) < thr);
    float q0 = __builtin_amdgcn_fmed3f(r0, lo_fast, hi), q1 = __builtin_amdgcn_fmed3f(r1, lo_fast, hi);
    float q2 = __builtin_amdgcn_fmed3f(r2, lo_fast, hi), q3 = __builtin_amdgcn_fmed3f(r3, lo_fast, hi);
    if (sgn) {   // two's-complement byte of a negative code
      q0 = q0 < 0.0f ? q0 + 256.0f : q0;
      q1 = q1 < 0.0f ? q1 + 256.0f : q1;
      q2 = q2 < 0.0f ? q2 + 256.0f : q2;
      q3 = q3 < 0.0f ? q3 + 256.0f : q3;
    }
    uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(q0, 0, 0u);
    w = __builtin_amdgcn_cvt_pk_u8_f32(q1, 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(q2, 2, w);
    return __builtin_amdgcn_cvt_pk_u8_f32(q3, 3, w) ^ xemit;
  }
  __device__ __forceinline__ uint32_t code4(const f32x4& v) const {
    bool unsure;
    const uint32_t w = code4_fast<false>(v, unsure);
    return unsure ? exact4(v, w) : w;
  }
  // N independent quads: one branch for all of them
  // `u` (optional) receives the per-quad flags, for callers that have more to redo on the exact path (a NaN's ReLU)
  // N independent quads of a PLAIN quantiser (epi_plain), written on PAIRS (v_pk_mul_f32 / v_pk_add_f32: one instruction for two
  // elements, the same roundings) where the instruction set has a packed form.  The tie flags stay wave masks (ballots of the
  // compares themselves: one scalar OR each, one uniform branch for all quads); a quad with a flagged lane anywhere in the wave is
  // redone by exact4 in every lane - which re-tests each element itself and returns the fast byte or the exact one, the same byte
  // either way (the fast path's proof holds per element).
  template <int N>
  __device__ __forceinline__ void code4n_plain(const f32x4 (&v)[N], uint32_t (&w)[N]) const {
    uint64_t bal[N], any = 0;
    const f32x2 rdv2 = f32x2{rdv, rdv};
    const float thr = tie_thr;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const f32x2 ta = f32x2{v[i].x, v[i].y} * rdv2, tb = f32x2{v[i].z, v[i].w} * rdv2;    // (= fma(v, rdv, 0) up to the sign of a zero, which nothing below sees)
      const f32x2 ra = f32x2{__builtin_rintf(ta.x), __builtin_rintf(ta.y)}, rb = f32x2{__builtin_rintf(tb.x), __builtin_rintf(tb.y)};
      const f32x2 da = ta - ra, db = tb - rb;
      bal[i] = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(da.x) < thr)) | __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(da.y) < thr)) |
               __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(db.x) < thr)) | __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(db.y) < thr));
      any |= bal[i];
      uint32_t c = __builtin_amdgcn_cvt_pk_u8_f32(ra.x, 0, 0u);       // saturates: negative -> 0 (the folded ReLU, the lower bound), beyond 255 -> 255
      c = __builtin_amdgcn_cvt_pk_u8_f32(ra.y, 1, c);
      c = __builtin_amdgcn_cvt_pk_u8_f32(rb.x, 2, c);
      w[i] = __builtin_amdgcn_cvt_pk_u8_f32(rb.y, 3, c) ^ xemit;
    }
    if (__builtin_expect(any != 0, false)) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (bal[i]) w[i] = exact4(v[i], w[i]);
    }
  }
  // One quad of a PLAIN quantiser in two parts, for a caller that puts other work (matrix instructions) between the branch-free fast path and
  // the rare redo: `w = code4_plain_fast(v, bal); ...; if (bal) w = exact4(v, w);` - code4n_plain<1> cut at its branch (csrc/conv3x3_pipe_i8.hip)
  __device__ __forceinline__ uint32_t code4_plain_fast(const f32x4& v, uint64_t& bal) const {
    const f32x2 rdv2 = f32x2{rdv, rdv};
    const float thr = tie_thr;
    const f32x2 ta = f32x2{v.x, v.y} * rdv2, tb = f32x2{v.z, v.w} * rdv2;
    const f32x2 ra = f32x2{__builtin_rintf(ta.x), __builtin_rintf(ta.y)}, rb = f32x2{__builtin_rintf(tb.x), __builtin_rintf(tb.y)};
    const f32x2 da = ta - ra, db = tb - rb;
    bal = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(da.x) < thr)) | __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(da.y) < thr)) |
          __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(db.x) < thr)) | __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(db.y) < thr));
    uint32_t c = __builtin_amdgcn_cvt_pk_u8_f32(ra.x, 0, 0u);
    c = __builtin_amdgcn_cvt_pk_u8_f32(ra.y, 1, c);
    c = __builtin_amdgcn_cvt_pk_u8_f32(rb.x, 2, c);
    return __builtin_amdgcn_cvt_pk_u8_f32(rb.y, 3, c) ^ xemit;
  }
  template <int N, bool PLAIN = false>
  __device__ __forceinline__ bool code4n(const f32x4 (&v)[N], uint32_t (&w)[N], bool (&u)[N]) const {
    bool any = false;
    if constexpr (PLAIN) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        w[i] = code4_fast<true, true>(v[i], u[i]);
        any |= u[i];
      }
    } else if (of == 0.0f) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        w[i] = code4_fast<true>(v[i], u[i]);
        any |= u[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        w[i] = code4_fast<false>(v[i], u[i]);
        any |= u[i];
      }
    }
    if (any) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (u[i]) w[i] = exact4(v[i], w[i]);
    }
    return any;
  }
};

}  // namespace dlmcq
