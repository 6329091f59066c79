// One fused step of the iterative l2norm scale refinement (ops.py:71-83 per tensor, :198-215 per channel):
//     q = clamp(r((x - o)/(s + 1e-7)), lo, hi)          (utils.py:1-2 `quantize`)
//     s_new[c] = SUM x*q / SUM (q*q + 1e-7)
// The reference spends 1 quantize (4 passes) + 2 multiplies + 2 reductions per iteration; here one read of x
// (4 B per element), partial sums per workgroup in fp32, final fold in fp64, true division.  Calibration-time
// only.  Sums are order-dependent, so this matches the reference to fp32 summation tolerance, not bit for bit.
#include "dlmcq_internal.h"

namespace dlmcq {

struct L2Acc {
  float a, b;
};

__device__ __forceinline__ void l2_add(L2Acc& acc, float x, float s, float o, float lo, float hi) {
  const float q = clamp_nan(__builtin_rintf((x - o) / (s + 1e-7f)), lo, hi);
  acc.a += x * q;
  acc.b += q * q + 1e-7f;
}

__device__ __forceinline__ L2Acc l2_block_reduce(L2Acc v) {
  __shared__ L2Acc part[DLMCQ_BLOCK / DLMCQ_WAVE];
#pragma unroll
  for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) {
    v.a += __shfl_xor(v.a, off, DLMCQ_WAVE);
    v.b += __shfl_xor(v.b, off, DLMCQ_WAVE);
  }
  if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) part[threadIdx.x / DLMCQ_WAVE] = v;
  __syncthreads();
  L2Acc r = part[0];
#pragma unroll
  for (int k = 1; k < DLMCQ_BLOCK / DLMCQ_WAVE; ++k) {
    r.a += part[k].a;
    r.b += part[k].b;
  }
  return r;
}

// Rows decomposition shared with the observer: block (c, seg) walks rows (n, c) for its share of `outer`;
// per tensor the caller passes channels = 1 and the kernel splits `inner` instead.
__global__ __launch_bounds__(DLMCQ_BLOCK) void l2_step_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ offset, int64_t outer,
                                                             int64_t channels, int64_t inner, int64_t npseg,
                                                             int64_t ipseg, float lo, float hi,
                                                             float* __restrict__ partials) {
  const int64_t c = blockIdx.x, sg = blockIdx.y, nseg = gridDim.y;
  const float s = scale[c], o = offset ? offset[c] : 0.0f;
  L2Acc acc{0.0f, 0.0f};
  int64_t n_lo = 0, n_hi = outer, i_lo = 0, i_hi = inner;
  if (channels == 1 && outer == 1) {          // per tensor: segments cut `inner`
    i_lo = sg * ipseg;
    i_hi = (i_lo + ipseg < inner) ? i_lo + ipseg : inner;
  } else {
    n_lo = sg * npseg;
    n_hi = (n_lo + npseg < outer) ? n_lo + npseg : outer;
  }
  for (int64_t n = n_lo; n < n_hi; ++n) {
    const float* __restrict__ row = x + (n * channels + c) * inner;
    for (int64_t i = i_lo + threadIdx.x; i < i_hi; i += DLMCQ_BLOCK) l2_add(acc, row[i], s, o, lo, hi);
  }
  const L2Acc r = l2_block_reduce(acc);
  if (threadIdx.x == 0) {
    partials[(sg * channels + c) * 2] = r.a;
    partials[(sg * channels + c) * 2 + 1] = r.b;
  }
  (void)nseg;
}

__global__ __launch_bounds__(DLMCQ_BLOCK) void l2_finalize_kernel(const float* __restrict__ partials, int64_t nseg,
                                                                 int64_t channels, float* __restrict__ new_scale) {
  const int64_t c = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x;
  if (c >= channels) return;
  double a = 0.0, b = 0.0;
  for (int64_t k = 0; k < nseg; ++k) {
    a += (double)partials[(k * channels + c) * 2];
    b += (double)partials[(k * channels + c) * 2 + 1];
  }
  new_scale[c] = (float)a / (float)b;
}

struct L2Plan {
  int64_t nseg, npseg, ipseg;
};

static L2Plan l2_plan(int64_t outer, int64_t channels, int64_t inner) {
  L2Plan p{1, outer, inner};
  const int64_t target = DLMCQ_CUS * 8;
  if (channels == 1 && outer == 1) {
    int64_t nseg = (inner + 16383) / 16384;       // >= 16K elements per workgroup
    if (nseg > target) nseg = target;
    if (nseg < 1) nseg = 1;
    p.ipseg = ((inner + nseg - 1) / nseg + 3) & ~int64_t(3);
    p.nseg = (inner + p.ipseg - 1) / p.ipseg;
    if (p.nseg < 1) p.nseg = 1;
  } else {
    int64_t nseg = (target + channels - 1) / channels;
    if (nseg > outer) nseg = outer;
    if (nseg < 1) nseg = 1;
    if (nseg > 65535) nseg = 65535;
    p.npseg = (outer + nseg - 1) / nseg;
    p.nseg = (outer + p.npseg - 1) / p.npseg;
  }
  return p;
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" size_t dlmcq_l2norm_scratch_bytes(int64_t outer, int64_t channels, int64_t inner) {
  if (outer < 1 || channels < 1 || inner < 1) return 0;
  const L2Plan p = l2_plan(outer, channels, inner);
  return (size_t)(p.nseg * channels * 2) * sizeof(float);
}

extern "C" int dlmcq_l2norm_step_f32(const float* x, const float* scale, const float* offset, float* new_scale,
                                     int64_t outer, int64_t channels, int64_t inner, int32_t lo, int32_t hi,
                                     void* scratch, size_t scratch_bytes, dlmcq_stream_t stream) {
  if (outer < 1 || channels < 1 || inner < 1 || lo > hi) return DLMCQ_EINVAL;
  if (!x || !scale || !new_scale) return DLMCQ_EINVAL;
  if (channels >= (1ll << 31)) return DLMCQ_ERANGE;
  if (channels == 1 && outer > 1) {  // per tensor over several outer slices: they are contiguous - flatten
    inner *= outer;
    outer = 1;
  }
  const L2Plan p = l2_plan(outer, channels, inner);
  if (!scratch || scratch_bytes < (size_t)(p.nseg * channels * 2) * sizeof(float)) return DLMCQ_ESCRATCH;
  float* part = reinterpret_cast<float*>(scratch);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(l2_step_kernel, dim3((uint32_t)channels, (uint32_t)p.nseg), dim3(DLMCQ_BLOCK), 0, st, x, scale, offset,
                     outer, channels, inner, p.npseg, p.ipseg, (float)lo, (float)hi, part);
  int rc = launch_status();
  if (rc != DLMCQ_OK) return rc;
  const int g = (int)((channels + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK);
  hipLaunchKernelGGL(l2_finalize_kernel, dim3(g), dim3(DLMCQ_BLOCK), 0, st, part, p.nseg, channels, new_scale);
  return launch_status();
}
