// RootQ weight forward (RootQ/base.py:146-155 + RootQ/function.py:15-32,58-67), per tensor, one pass.
//
// The reference runs ~15 elementwise passes including a `pow`; the forward VALUE only needs the sign
// of phi = pow(k|d| + 1e-5, alpha) * d/(|d| + 1e-5), and pow(...) > 0 always (base >= 1e-5, alpha in
// [1e-4, 1]), so sgn(phi) = sgn(d) for every |d| >= 2^-126 - alpha only shapes the gradient.  (For
// |d| < ~1e-40 the reference's product can underflow to 0; weights never get there.)
#include "dlmcq_internal.h"

namespace dlmcq {

__device__ __forceinline__ float rootq_w_one(float w, float up, float lw, float delta) {
  float t = w + relu_nan(lw - w);           // clip lower (additive, as the reference)
  t = t - relu_nan(t - up);                 // clip upper
  const float v = (t - lw) / delta;
  const float fl = __builtin_floorf(v);
  const float iv = (fl - v) + v;            // floor_pass forward value
  const float mi = (iv + 0.5f) * delta + lw;
  const float d = t - mi;
  const float s = (float)((0.0f < d) - (d < 0.0f));   // torch.sgn; NaN -> 0 (result is NaN anyway)
  return ((s + 1.0f) / 2.0f + iv) * delta + lw;
}

template <int U>
__global__ __launch_bounds__(DLMCQ_WAVE) void rootq_weight_kernel(const float* w, float* y,
                                                                  const float* __restrict__ bounds, int64_t n,
                                                                  float range, int vec) {
  const float up = bounds[0], lw = bounds[1];
  const float delta = (up - lw) / range;
  const int64_t n4 = vec ? (n >> 2) : 0;
  const int64_t nchunks = (n4 + DLMCQ_WAVE * U - 1) / (DLMCQ_WAVE * U);
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int64_t i0 = chunk * (DLMCQ_WAVE * U) + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * DLMCQ_WAVE;
      if (i < n4) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w) + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * DLMCQ_WAVE;
      if (i < n4) {
        f32x4 o;
        o.x = rootq_w_one(v[u].x, up, lw, delta);
        o.y = rootq_w_one(v[u].y, up, lw, delta);
        o.z = rootq_w_one(v[u].z, up, lw, delta);
        o.w = rootq_w_one(v[u].w, up, lw, delta);
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(y) + i);
      }
    }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * DLMCQ_WAVE + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * DLMCQ_WAVE)
    y[i] = rootq_w_one(w[i], up, lw, delta);
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_rootq_weight_f32(const float* w, float* y, const float* bounds, int64_t n, int32_t lo, int32_t hi,
                                      dlmcq_stream_t stream) {
  if (n < 0 || lo >= hi) return DLMCQ_EINVAL;
  if (n == 0) return DLMCQ_OK;
  if (!w || !y || !bounds) return DLMCQ_EINVAL;
  constexpr int U = 1;   // one-wave workgroups, one float4 per lane (the fake-quant kernels' measured optimum)
  const int vec = aligned16(w) && aligned16(y);
  int64_t b = ((n >> 2) + DLMCQ_WAVE * U - 1) / (DLMCQ_WAVE * U);
  if (b < 1) b = 1;
  if (b > (1 << 24)) b = 1 << 24;
  hipLaunchKernelGGL((rootq_weight_kernel<U>), dim3((int)b), dim3(DLMCQ_WAVE), 0, reinterpret_cast<hipStream_t>(stream),
                     w, y, bounds, n, (float)(hi - lo), vec);
  return launch_status();
}
