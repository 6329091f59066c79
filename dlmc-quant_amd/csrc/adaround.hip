// AdaRound weight path of the few-shot PTQ wrapper (FSPTQuant/base.py:69-79,136-141,151-152), fused.
//   forward   q = floor(w / s_k) + h(alpha)            (training: h = clamp(sigmoid(alpha)*1.2 - 0.1, 0, 1))
//                                 + [alpha >= 0]        (eval)
//             y = clamp(q, lo, hi) * s_k
//   backward  (only alpha and the per-channel scale learn; floor() passes no gradient to w)
//             g_alpha = gy * s_k * [lo <= q <= hi] * 1.2*sig*(1-sig) * [0 < 1.2*sig - 0.1 < 1]
//             g_s[k]  = sum over the channel of gy * clamp(q, lo, hi)
// The reference runs ~9 elementwise launches forward and ~15 backward per layer per reconstruction step; weights
// are small, so the win is launches, not bytes.  The eval form is bit-exact; the training form matches the
// reference to the accuracy of expf (the CPU reference uses a different exp implementation).
#include "dlmcq_internal.h"

namespace dlmcq {

__device__ __forceinline__ float sigmoidf_(float a) { return 1.0f / (1.0f + __expf(-a)); }

__global__ __launch_bounds__(DLMCQ_BLOCK) void adaround_fwd_kernel(const float* __restrict__ w,
                                                                  const float* __restrict__ alpha,
                                                                  const float* __restrict__ scale, float* __restrict__ y,
                                                                  int64_t inner, float lo, float hi, int training) {
  const int64_t k = blockIdx.x;
  const float s = scale[k];
  for (int64_t i = threadIdx.x; i < inner; i += DLMCQ_BLOCK) {
    const int64_t e = k * inner + i;
    const float a = alpha[e];
    float q = __builtin_floorf(w[e] / s);
    if (training) {
      const float t = 1.0f / (1.0f + expf(-a));            // accurate expf: this is a weights-sized kernel
      q = q + clamp_nan(t * (1.1f - (-0.1f)) + (-0.1f), 0.0f, 1.0f);
    } else {
      q = q + (a >= 0.0f ? 1.0f : 0.0f);
    }
    y[e] = clamp_nan(q, lo, hi) * s;
  }
}

__global__ __launch_bounds__(DLMCQ_BLOCK) void adaround_bwd_kernel(const float* __restrict__ w,
                                                                  const float* __restrict__ alpha,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ gy, float* __restrict__ g_alpha,
                                                                  float* __restrict__ g_scale, int64_t inner, float lo,
                                                                  float hi) {
  __shared__ float part[DLMCQ_BLOCK / DLMCQ_WAVE];
  const int64_t k = blockIdx.x;
  const float s = scale[k];
  float acc = 0.0f;
  for (int64_t i = threadIdx.x; i < inner; i += DLMCQ_BLOCK) {
    const int64_t e = k * inner + i;
    const float sig = 1.0f / (1.0f + expf(-alpha[e]));
    const float hraw = sig * 1.2f + (-0.1f);
    const float q = __builtin_floorf(w[e] / s) + clamp_nan(hraw, 0.0f, 1.0f);
    const float g = gy[e];
    const bool in_q = (q >= lo) && (q <= hi);               // clamp passes gradient on the closed interval
    const bool in_h = (hraw >= 0.0f) && (hraw <= 1.0f);
    if (g_alpha) g_alpha[e] = (in_q && in_h) ? g * s * (1.2f * sig * (1.0f - sig)) : 0.0f;
    acc += g * clamp_nan(q, lo, hi);
  }
#pragma unroll
  for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DLMCQ_WAVE);
  if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) part[threadIdx.x / DLMCQ_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && g_scale) g_scale[k] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_adaround_weight_f32(const float* w, const float* alpha, const float* scale, float* y,
                                         int64_t out_channels, int64_t inner, int32_t lo, int32_t hi, int32_t training,
                                         dlmcq_stream_t stream) {
  if (out_channels < 0 || inner < 0 || lo > hi) return DLMCQ_EINVAL;
  if (out_channels == 0 || inner == 0) return DLMCQ_OK;
  if (!w || !alpha || !scale || !y) return DLMCQ_EINVAL;
  if (out_channels >= (1ll << 31)) return DLMCQ_ERANGE;
  hipLaunchKernelGGL(adaround_fwd_kernel, dim3((uint32_t)out_channels), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), w, alpha, scale, y, inner, (float)lo, (float)hi, training);
  return launch_status();
}

extern "C" int dlmcq_adaround_weight_bwd_f32(const float* w, const float* alpha, const float* scale, const float* gy,
                                             float* g_alpha, float* g_scale, int64_t out_channels, int64_t inner,
                                             int32_t lo, int32_t hi, dlmcq_stream_t stream) {
  if (out_channels < 0 || inner < 0 || lo > hi) return DLMCQ_EINVAL;
  if (out_channels == 0 || inner == 0) return DLMCQ_OK;
  if (!w || !alpha || !scale || !gy || (!g_alpha && !g_scale)) return DLMCQ_EINVAL;
  if (out_channels >= (1ll << 31)) return DLMCQ_ERANGE;
  hipLaunchKernelGGL(adaround_bwd_kernel, dim3((uint32_t)out_channels), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), w, alpha, scale, gy, g_alpha, g_scale, inner, (float)lo,
                     (float)hi);
  return launch_status();
}
