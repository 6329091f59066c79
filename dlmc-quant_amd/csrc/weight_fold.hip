// Weight transforms that precede the fake-quantize path in the reference's few-shot PTQ flow
// (example/quantization/FSPTQuant.py:65-67): RepVGG re-parameterisation and BatchNorm folding.
// Weights-sized, one launch each, all arithmetic on device in the reference's operation order
// (IEEE sqrt/div, no FMA contraction) so the folded weights are bit-identical to the CPU reference.
#include "dlmcq_internal.h"

namespace dlmcq {

// dlmc/utils/merge_bn.py:85-101:  var' = var + 1e-7 (NOT bn.eps);  w <- (w * gamma) / sqrt(var');
// b <- (gamma * (b - mean)) / sqrt(var') + beta.
__global__ __launch_bounds__(DLMCQ_BLOCK) void fold_bn_kernel(float* w, float* bias, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ var, int64_t inner,
                                                             float var_eps) {
  const int64_t k = blockIdx.x;
  const float g = gamma[k];
  const float sd = __builtin_sqrtf(var[k] + var_eps);
  float* row = w + k * inner;
  for (int64_t i = threadIdx.x; i < inner; i += DLMCQ_BLOCK) row[i] = (row[i] * g) / sd;
  if (threadIdx.x == 0) bias[k] = (g * (bias[k] - mean[k])) / sd + beta[k];
}

struct BnRef {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float eps;
};

// model/classification/repvgg.py:92-130 (get_equivalent_kernel_bias): per branch std = sqrt(var + eps),
// t = gamma/std, kernel*t, bias = beta - (mean*gamma)/std; sum = (3x3 + pad(1x1)) + identity.
__global__ __launch_bounds__(DLMCQ_BLOCK) void repvgg_fuse_kernel(const float* __restrict__ k3,
                                                                 const float* __restrict__ k1, float* out_k,
                                                                 float* out_b, BnRef b3, BnRef b1, BnRef bid,
                                                                 int has_id, int64_t cin_g) {
  const int64_t k = blockIdx.x;
  const float t3 = b3.gamma[k] / __builtin_sqrtf(b3.var[k] + b3.eps);
  const float t1 = b1.gamma[k] / __builtin_sqrtf(b1.var[k] + b1.eps);
  const float tid = has_id ? bid.gamma[k] / __builtin_sqrtf(bid.var[k] + bid.eps) : 0.0f;
  const int64_t n = cin_g * 9;
  for (int64_t i = threadIdx.x; i < n; i += DLMCQ_BLOCK) {
    const int64_t c = i / 9, tap = i - c * 9;
    const float a = k3[k * n + i] * t3;
    const float b = tap == 4 ? k1[k * cin_g + c] * t1 : 0.0f;           // F.pad(kernel1x1 * t1, [1,1,1,1])
    float r = a + b;
    if (has_id) r = r + ((tap == 4 && c == k % cin_g) ? 1.0f : 0.0f) * tid;  // id_tensor * t
    else r = r + 0.0f;                                                   // "+ 0" of the absent branch
    out_k[k * n + i] = r;
  }
  if (threadIdx.x == 0) {
    const float s3 = __builtin_sqrtf(b3.var[k] + b3.eps), s1 = __builtin_sqrtf(b1.var[k] + b1.eps);
    float bias = (b3.beta[k] - (b3.mean[k] * b3.gamma[k]) / s3) + (b1.beta[k] - (b1.mean[k] * b1.gamma[k]) / s1);
    if (has_id) bias = bias + (bid.beta[k] - (bid.mean[k] * bid.gamma[k]) / __builtin_sqrtf(bid.var[k] + bid.eps));
    else bias = bias + 0.0f;
    out_b[k] = bias;
  }
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_fold_bn_f32(float* weight, float* bias, const float* gamma, const float* beta, const float* mean,
                                 const float* var, int64_t out_channels, int64_t inner, float var_eps,
                                 dlmcq_stream_t stream) {
  if (out_channels < 0 || inner < 0) return DLMCQ_EINVAL;
  if (out_channels == 0) return DLMCQ_OK;
  if (!weight || !bias || !gamma || !beta || !mean || !var) return DLMCQ_EINVAL;
  if (out_channels >= (1ll << 31)) return DLMCQ_ERANGE;
  hipLaunchKernelGGL(fold_bn_kernel, dim3((uint32_t)out_channels), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), weight, bias, gamma, beta, mean, var, inner, var_eps);
  return launch_status();
}

extern "C" int dlmcq_repvgg_fuse_f32(const float* k3, const float* k1, float* out_kernel, float* out_bias,
                                     const float* const* bn3, const float* const* bn1, const float* const* bnid,
                                     float eps3, float eps1, float epsid, int64_t out_channels, int64_t cin_per_group,
                                     dlmcq_stream_t stream) {
  if (out_channels < 0 || cin_per_group < 1) return DLMCQ_EINVAL;
  if (out_channels == 0) return DLMCQ_OK;
  if (!k3 || !k1 || !out_kernel || !out_bias || !bn3 || !bn1) return DLMCQ_EINVAL;
  for (int j = 0; j < 4; ++j)
    if (!bn3[j] || !bn1[j] || (bnid && !bnid[j])) return DLMCQ_EINVAL;
  if (out_channels >= (1ll << 31)) return DLMCQ_ERANGE;
  BnRef b3{bn3[0], bn3[1], bn3[2], bn3[3], eps3}, b1{bn1[0], bn1[1], bn1[2], bn1[3], eps1};
  BnRef bi{nullptr, nullptr, nullptr, nullptr, epsid};
  if (bnid) bi = BnRef{bnid[0], bnid[1], bnid[2], bnid[3], epsid};
  hipLaunchKernelGGL(repvgg_fuse_kernel, dim3((uint32_t)out_channels), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), k3, k1, out_kernel, out_bias, b3, b1, bi, bnid ? 1 : 0,
                     cin_per_group);
  return launch_status();
}
