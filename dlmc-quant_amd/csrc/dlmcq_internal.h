// Shared device/host helpers for the gfx950 fake-quantize kernels.  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dlmcq.h"

#define DLMCQ_WAVE 64          // CDNA wavefront
#define DLMCQ_BLOCK 256        // 4 waves: one per SIMD of a CU
#define DLMCQ_CUS 256          // MI355X

namespace dlmcq {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- exact division of n < 2^31 by a runtime-constant d < 2^31 (Granlund-Montgomery) ----------
// l = ceil(log2 d), M = floor(2^(31+l)/d) + 1 < 2^32, q = (n*M) >> (31+l).  The error term
// e = M*d - 2^(31+l) lies in (0, d], so n*e < 2^(31+l) for every n < 2^31: exact.
struct FastDiv {
  uint32_t mul;
  uint32_t shift;  // 31 + l
  uint32_t d;
};

static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = 31 + l;
  f.mul = (uint32_t)(((1ull << f.shift) / d) + 1ull);
  return f;
}

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (uint32_t)(((uint64_t)n * f.mul) >> f.shift);
}

// ---- torch-semantics scalar helpers (NaN-propagating; IEEE; never contracted) ----------------
__device__ __forceinline__ float clamp_nan(float v, float lo, float hi) {
  // torch.clamp: NaN stays NaN (both compares false)
  return v < lo ? lo : (v > hi ? hi : v);
}

__device__ __forceinline__ float relu_nan(float v) {
  // torch.relu = clamp_min(0): NaN stays NaN, -0 -> ... max(-0, 0): ATen returns the input when
  // it is not smaller than 0, so -0.0 stays -0.0
  return v < 0.0f ? 0.0f : v;
}

__device__ __forceinline__ float ste_round(float v) {
  // forward value of the reference's round_pass: (round(v) - v) + v
  float r = __builtin_rintf(v);
  return (r - v) + v;
}

__device__ __forceinline__ float ste_scale(float s, float g) {
  // forward value of grad_scale: (s - s*g) + s*g
  float sg = s * g;
  return (s - sg) + sg;
}

__device__ __forceinline__ int code_of(float q) {
  // q is integral (or NaN -> 0) and already clamped to [lo, hi] by the caller's form
  return (q != q) ? 0 : (int)q;
}

// Per-channel constants of one form, derived from the stored (scale, offset) exactly as the
// reference derives them, once per float4 (or once per thread when the whole tensor shares them).
template <int FORM>
struct ChanConst {
  float dv;   // divisor
  float ml;   // multiplier of the dequant
  float of;   // offset / zero point
  float up;   // ROOTQ_ACT: upper clip
  __device__ __forceinline__ ChanConst(float s, float o, float g, float lo, float hi) {
    of = o;
    up = 0.0f;
    if (FORM == DLMCQ_FORM_EMULATE) {
      dv = s + 1e-7f;
      ml = s;
    } else if (FORM == DLMCQ_FORM_QBASE) {
      dv = ste_scale(s, g);
      ml = dv;
    } else {
      dv = s;
      ml = s;
      if (FORM == DLMCQ_FORM_ROOTQ_ACT) up = s * (hi - lo);
    }
  }
};

// One element: returns the code q (fp32, integral or NaN) and the fake-quantised value y.
template <int FORM>
__device__ __forceinline__ void fq_one(float x, const ChanConst<FORM>& c, float lo, float hi, float& q,
                                       float& y) {
  if (FORM == DLMCQ_FORM_EMULATE) {
    q = clamp_nan(__builtin_rintf((x - c.of) / c.dv), lo, hi);
    y = q * c.ml + c.of;
  } else if (FORM == DLMCQ_FORM_QBASE) {
    q = ste_round(clamp_nan((x - c.of) / c.dv, lo, hi));
    y = q * c.ml + c.of;
  } else if (FORM == DLMCQ_FORM_ZEROPOINT) {
    q = clamp_nan(ste_round(x / c.dv) + c.of, lo, hi);
    y = (q - c.of) * c.ml;
  } else if (FORM == DLMCQ_FORM_SYMMETRIC) {
    q = clamp_nan(ste_round(x / c.dv), lo, hi);
    y = q * c.ml;
  } else {  // DLMCQ_FORM_ROOTQ_ACT
    float t = x + relu_nan(0.0f - x);
    t = t - relu_nan(t - c.up);
    q = ste_round(t / c.dv);
    y = q * c.ml;
  }
}

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DLMCQ_OK : (int)e;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline bool aligned4(const void* p) { return (((uintptr_t)p) & 3u) == 0; }

}  // namespace dlmcq
