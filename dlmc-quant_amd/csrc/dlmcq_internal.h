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

static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DLMCQ_OK : (int)e;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline bool aligned4(const void* p) { return (((uintptr_t)p) & 3u) == 0; }

}  // namespace dlmcq
