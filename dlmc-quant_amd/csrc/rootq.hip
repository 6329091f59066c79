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

// ---------------------------------------------------------------------------------------------------------
// Backward of the RootQ weight transform as autograd runs RootQ/base.py:146-155 + RootQ/function.py:15-32,58-67:
//   wc = clip(w, U, L);  d = (U - L)/(hi - lo);  v = (wc - L)/d;  I = floor_pass(v);  mi = (I + .5) d + L  (detached)
//   phi = pow(k|e| + 1e-5, a) * e/(|e| + 1e-5),  e = wc - mi,  k = 2/d,  a = alpha clipped additively to [1e-4, 1]
//   y = ((sgn_ste(phi) + 1)/2 + I) d + L
// The sign and the floor are straight-through, so the gradient reaches w through BOTH phi and v, and the three scalars
// (upper, lower, alpha) through every element.  One pass: gw per element, four block-reduced sums (d-path, L-path,
// U-path, alpha), folded in fp64 by the finalize.  The reference does this with ~35 elementwise launches per layer per
// step on a tensor of at most a few MB - launch-bound; here it is two launches.
namespace dlmcq {

constexpr int RQ_SUMS = 4;   // [g_delta, g_lower (direct + clip + v), g_upper (clip), g_alpha']

__global__ __launch_bounds__(DLMCQ_BLOCK) void rootq_weight_bwd_kernel(const float* __restrict__ w, const float* __restrict__ gy,
                                                                      float* __restrict__ gw, const float* __restrict__ bounds,
                                                                      const float* __restrict__ alpha_p, int64_t n, float range,
                                                                      float* __restrict__ partials) {
  const float up = bounds[0], lw = bounds[1];
  const float delta = (up - lw) / range;
  const float k = 2.0f / delta;
  float a = alpha_p[0];
  a = a + relu_nan(1e-4f - a);
  a = a - relu_nan(a - 1.0f);
  float s_d = 0.0f, s_l = 0.0f, s_u = 0.0f, s_a = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * DLMCQ_BLOCK) {
    const float x = w[i], g = gy[i];
    const bool m_lo = (lw - x) > 0.0f;
    const float t1 = x + relu_nan(lw - x);
    const bool m_up = (t1 - up) > 0.0f;
    const float wc = t1 - relu_nan(t1 - up);
    const float v = (wc - lw) / delta;
    const float fl = __builtin_floorf(v);
    const float iv = (fl - v) + v;
    const float mi = (iv + 0.5f) * delta + lw;
    const float e = wc - mi;
    const float ae = __builtin_fabsf(e);
    const float sg = (float)((0.0f < e) - (e < 0.0f));
    const float den = ae + 1e-5f;
    const float sign = e / den;
    const float B = k * ae + 1e-5f;
    const float P = powf(B, a);
    const float phi = P * sign;
    const float sphi = (float)((0.0f < phi) - (phi < 0.0f));
    // y = ((sphi + 1)/2 + iv) * delta + lw
    const float g_phi = g * delta * 0.5f;                 // sgn is straight-through
    const float g_v = g * delta;                          // floor is straight-through
    s_d += g * ((sphi + 1.0f) * 0.5f + iv) - g_v * (v / delta);
    s_l += g - g_v / delta;
    // phi = P * sign
    const float g_P = g_phi * sign, g_sign = g_phi * P;
    const float g_B = g_P * a * powf(B, a - 1.0f);
    s_a += g_P * P * logf(B);
    const float g_k = g_B * ae;
    s_d += g_k * (-2.0f / (delta * delta));
    const float g_e = g_B * k * sg + g_sign / den - (g_sign * e / (den * den)) * sg;
    const float g_wc = g_v / delta + g_e;
    // the clip
    const float g_t1 = m_up ? 0.0f : g_wc;
    if (m_up) s_u += g_wc;
    if (m_lo) s_l += g_t1;
    if (gw) gw[i] = m_lo ? 0.0f : g_t1;
  }
  __shared__ float red[RQ_SUMS][DLMCQ_BLOCK / DLMCQ_WAVE];
  float vals[RQ_SUMS] = {s_d, s_l, s_u, s_a};
#pragma unroll
  for (int q = 0; q < RQ_SUMS; ++q) {
    float t = vals[q];
#pragma unroll
    for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, DLMCQ_WAVE);
    if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) red[q][threadIdx.x / DLMCQ_WAVE] = t;
  }
  __syncthreads();
  if (threadIdx.x < RQ_SUMS) {
    float t = 0.0f;
    for (int wv = 0; wv < DLMCQ_BLOCK / DLMCQ_WAVE; ++wv) t += red[threadIdx.x][wv];
    partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = t;
  }
}

// out = {g_upper, g_lower, g_alpha}: delta = (U - L)/range feeds both bounds; alpha passes where it was not clipped
__global__ __launch_bounds__(DLMCQ_BLOCK) void rootq_weight_bwd_finalize_kernel(const float* __restrict__ partials, int nblk,
                                                                               const float* __restrict__ alpha_p, float range,
                                                                               float* __restrict__ out) {
  __shared__ double red[RQ_SUMS][DLMCQ_BLOCK];
  for (int q = 0; q < RQ_SUMS; ++q) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += DLMCQ_BLOCK) s += (double)partials[(int64_t)q * nblk + i];
    red[q][threadIdx.x] = s;
  }
  __syncthreads();
  for (int off = DLMCQ_BLOCK / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off)
      for (int q = 0; q < RQ_SUMS; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double g_d = red[0][0], g_l = red[1][0], g_u = red[2][0], g_a = red[3][0];
    out[0] = (float)(g_u + g_d / range);
    out[1] = (float)(g_l - g_d / range);
    const float a0 = alpha_p[0];
    const float a1 = a0 + relu_nan(1e-4f - a0);
    const bool pass = !((a1 - 1.0f) > 0.0f) && !((1e-4f - a0) > 0.0f);
    out[2] = pass ? (float)g_a : 0.0f;
  }
}

constexpr int RQ_BWD_BLOCKS = 1024;

}  // namespace dlmcq

extern "C" size_t dlmcq_rootq_bwd_scratch_bytes(int64_t n) {
  (void)n;
  return (size_t)dlmcq::RQ_SUMS * dlmcq::RQ_BWD_BLOCKS * sizeof(float);
}

extern "C" int dlmcq_rootq_weight_bwd_f32(const float* w, const float* gy, float* gw, float* g_bounds_alpha, const float* bounds,
                                          const float* alpha, int64_t n, int32_t lo, int32_t hi, void* scratch,
                                          size_t scratch_bytes, dlmcq_stream_t stream) {
  if (n < 0 || lo >= hi) return DLMCQ_EINVAL;
  if (!bounds || !alpha || !g_bounds_alpha) return DLMCQ_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n == 0) {
    (void)hipMemsetAsync(g_bounds_alpha, 0, 3 * sizeof(float), st);
    return launch_status();
  }
  if (!w || !gy) return DLMCQ_EINVAL;
  if (!scratch || scratch_bytes < dlmcq_rootq_bwd_scratch_bytes(n)) return DLMCQ_ESCRATCH;
  int64_t b = (n + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
  if (b > RQ_BWD_BLOCKS) b = RQ_BWD_BLOCKS;
  float* part = static_cast<float*>(scratch);
  hipLaunchKernelGGL(rootq_weight_bwd_kernel, dim3((int)b), dim3(DLMCQ_BLOCK), 0, st, w, gy, gw, bounds, alpha, n, (float)(hi - lo), part);
  int rc = launch_status();
  if (rc != DLMCQ_OK) return rc;
  hipLaunchKernelGGL(rootq_weight_bwd_finalize_kernel, dim3(1), dim3(DLMCQ_BLOCK), 0, st, part, (int)b, alpha, (float)(hi - lo),
                     g_bounds_alpha);
  return launch_status();
}
