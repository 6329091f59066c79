// Epilogue shared by the int8 convolution kernels (conv_i8.hip, conv_stem_i8.hip).  Not part of the ABI.
#pragma once

#include "dlmcq_internal.h"

namespace dlmcq {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

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
};

struct EpiQuant {   // the consumer's constants, resolved once per thread
  float dv, rdv, of, zadd, lo, hi;
  int form;
  bool sgn;
  __device__ __forceinline__ EpiQuant(const ConvEpi& ep)
      : dv(1.0f), rdv(1.0f), of(0.0f), zadd(0.0f), lo(ep.q_lo), hi(ep.q_hi), form(ep.q_form), sgn(ep.q_lo < 0.0f) {
    if (!ep.codes) return;
    const float s = ep.q_scale[0];
    const float z = ep.q_zp ? ep.q_zp[0] : 0.0f;
    dv = form == DLMCQ_FORM_EMULATE ? s + 1e-7f : (form == DLMCQ_FORM_QBASE ? ste_scale(s, ep.q_g) : s);
    // EMULATE / QBASE divide (v - offset); ZEROPOINT adds the zero point after rounding; v - 0 is v, r + 0 is r
    of = (form == DLMCQ_FORM_EMULATE || form == DLMCQ_FORM_QBASE) ? z : 0.0f;
    zadd = form == DLMCQ_FORM_ZEROPOINT ? z : 0.0f;
    // the fast path below is proven for a well-scaled divisor and a byte-sized zero point; anything else (and NaN)
    // makes rdv NaN, which routes every element to the exact division
    const bool tame = __builtin_fabsf(dv) >= 0x1p-100f && __builtin_fabsf(dv) <= 0x1p100f && __builtin_fabsf(zadd) <= 256.0f;
    rdv = tame ? 1.0f / dv : __builtin_nanf("");
  }
  // the reference arithmetic itself, form by form (fq_one): what the fast path below must reproduce, and what decides
  // the elements it cannot vouch for.  (Not interchangeable with the reduced formula for +-inf: the STE identity
  // R(v) = (rint(v) - v) + v turns an infinite quotient into NaN -> code 0, where EMULATE's plain rint saturates.)
  __device__ __forceinline__ uint32_t exact(float v) const {
    float q;
    if (form == DLMCQ_FORM_EMULATE) q = clamp_nan(__builtin_rintf((v - of) / dv), lo, hi);
    else if (form == DLMCQ_FORM_QBASE) q = ste_round(clamp_nan((v - of) / dv, lo, hi));
    else if (form == DLMCQ_FORM_ZEROPOINT) q = clamp_nan(ste_round(v / dv) + zadd, lo, hi);
    else q = clamp_nan(ste_round(v / dv), lo, hi);
    return (uint32_t)(code_of(q) & 0xff);
  }
  // For a FINITE quotient d = fl(u / dv), u = v - of, all four forms reduce to  code = clamp(rint(d) + zadd, lo, hi)
  // (rint(clamp(d)) = clamp(rint(d)) for integral bounds; the STE identity (r - d) + d returns r exactly).
  // A correctly rounded division costs ~25 VALU operations per element - more than everything else in the epilogue -
  // so d is replaced by t = fl(u * fl(1/dv)), which differs from d by less than 2^-22 |t|.  For |t| <= 512 that is
  // below 2^-13: unless t lies within 2^-12 of a rounding tie (x.5), rint(t) = rint(d); for |t| > 512 both saturate
  // to the same bound (|zadd| <= 256, |lo|, |hi| <= 255).  Ties that close, infinities and NaNs (about one element in
  // 2000) take the exact division, four elements at a time.  Bit-identical codes at ~10 operations per element.
  __device__ __forceinline__ uint32_t code4(const f32x4& v) const {
    const float t0 = (v.x - of) * rdv, t1 = (v.y - of) * rdv, t2 = (v.z - of) * rdv, t3 = (v.w - of) * rdv;
    const float r0 = __builtin_rintf(t0), r1 = __builtin_rintf(t1), r2 = __builtin_rintf(t2), r3 = __builtin_rintf(t3);
    // worst distance from an integer; a non-finite input poisons it through (sum * 0), max() alone would drop a NaN
    float worst = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(t0 - r0), __builtin_fabsf(t1 - r1)),
                                  __builtin_fmaxf(__builtin_fabsf(t2 - r2), __builtin_fabsf(t3 - r3)));
    worst = worst + ((t0 + t1) + (t2 + t3)) * 0.0f;
    if (!(worst < 0.5f - 0x1p-12f)) return exact(v.x) | (exact(v.y) << 8) | (exact(v.z) << 16) | (exact(v.w) << 24);
    float q0 = __builtin_amdgcn_fmed3f(r0 + zadd, lo, hi), q1 = __builtin_amdgcn_fmed3f(r1 + zadd, lo, hi);
    float q2 = __builtin_amdgcn_fmed3f(r2 + zadd, lo, hi), q3 = __builtin_amdgcn_fmed3f(r3 + zadd, lo, hi);
    if (sgn) {   // two's-complement byte of a negative code
      q0 = q0 < 0.0f ? q0 + 256.0f : q0;
      q1 = q1 < 0.0f ? q1 + 256.0f : q1;
      q2 = q2 < 0.0f ? q2 + 256.0f : q2;
      q3 = q3 < 0.0f ? q3 + 256.0f : q3;
    }
    uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(q0, 0, 0u);
    w = __builtin_amdgcn_cvt_pk_u8_f32(q1, 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(q2, 2, w);
    return __builtin_amdgcn_cvt_pk_u8_f32(q3, 3, w);
  }
};

}  // namespace dlmcq
