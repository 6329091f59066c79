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

// ------------------------------------------------------------------------------------------------------------
// The rest of the calibration-time estimators of ops.py as device code: nothing below syncs
// with the host inside a loop, and no tensor-sized step is left to a chain of library element-wise kernels.

// ---- convergence on the device: the tail of one l2norm iteration (ops.py:77-81, :205-210, :96-108, :268-284) ----
// state[0] = done flag, state[1] = iterations run, state[2] = best mse (output-aware variants), all as floats.
// mode 0: scale <- new, stop when |new - s| / s <= eps (per tensor) or ||new - s|| / ||s|| <= eps (per channel).
// mode 1 (l2norm_output, ops.py:85-109):          scale <- new; if mse < best: best_scale <- scale (the NEW one).
// mode 2 (l2norm_output_channel, ops.py:252-292): if mse < best: best_scale <- scale (the OLD one); scale <- new.
__global__ __launch_bounds__(DLMCQ_BLOCK) void l2_update_kernel(const float* __restrict__ partials, int64_t nseg, int64_t channels,
                                                               int nsum, float* __restrict__ scale, float* __restrict__ best_scale,
                                                               float* __restrict__ state, int mode, float mse_div, float eps) {
  if (state[0] != 0.0f) return;
  __shared__ double red[3][DLMCQ_BLOCK];
  double dn = 0.0, dd = 0.0, se = 0.0;
  for (int64_t c = threadIdx.x; c < channels; c += DLMCQ_BLOCK) {
    double a = 0.0, b = 0.0, e = 0.0;
    for (int64_t k = 0; k < nseg; ++k) {
      const float* p = partials + (k * channels + c) * nsum;
      a += (double)p[0];
      b += (double)p[1];
      if (nsum > 2) e += (double)p[2];
    }
    const float s = scale[c], ns = (float)a / (float)b;
    if (channels == 1) {
      dn = (double)(__builtin_fabsf(ns - s) / s);
      dd = 1.0;
    } else {
      dn += (double)((ns - s) * (ns - s));
      dd += (double)(s * s);
    }
    se += e;
    if (mode == 2) best_scale[channels + c] = s;          // staging: the old scale, taken below if this mse is the best
    scale[c] = ns;
  }
  red[0][threadIdx.x] = dn;
  red[1][threadIdx.x] = dd;
  red[2][threadIdx.x] = se;
  __syncthreads();
  for (int off = DLMCQ_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  const float diff = channels == 1 ? (float)red[0][0] : __builtin_sqrtf((float)red[0][0]) / __builtin_sqrtf((float)red[1][0]);
  const bool better = mode != 0 && (float)(red[2][0] / (double)mse_div) < state[2];
  __syncthreads();
  if (better)
    for (int64_t c = threadIdx.x; c < channels; c += DLMCQ_BLOCK) best_scale[c] = mode == 1 ? scale[c] : best_scale[channels + c];
  if (threadIdx.x == 0) {
    if (better) state[2] = (float)(red[2][0] / (double)mse_div);
    state[1] += 1.0f;
    if (!(diff > eps)) state[0] = 1.0f;
  }
}

__global__ __launch_bounds__(DLMCQ_BLOCK) void l2_step_guarded_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                     const float* __restrict__ offset, int64_t outer,
                                                                     int64_t channels, int64_t inner, int64_t npseg, int64_t ipseg,
                                                                     float lo, float hi, float* __restrict__ partials,
                                                                     const float* __restrict__ state) {
  if (state[0] != 0.0f) return;                 // converged: the remaining launches of the batch do nothing
  const int64_t c = blockIdx.x, sg = blockIdx.y;
  const float s = scale[c], o = offset ? offset[c] : 0.0f;
  L2Acc acc{0.0f, 0.0f};
  int64_t n_lo = 0, n_hi = outer, i_lo = 0, i_hi = inner;
  if (channels == 1 && outer == 1) {
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
}

// ---- output-aware step (ops.py:96-108, :268-284): SUM o*oq, SUM (oq*oq + 1e-7), SUM (o - oq)^2 in ONE read of both ----
__global__ __launch_bounds__(DLMCQ_BLOCK) void l2out_sums_kernel(const float* __restrict__ o, const float* __restrict__ oq,
                                                                int64_t outer, int64_t channels, int64_t inner, int64_t npseg,
                                                                int64_t ipseg, float* __restrict__ partials,
                                                                const float* __restrict__ state) {
  if (state && state[0] != 0.0f) return;
  const int64_t c = blockIdx.x, sg = blockIdx.y;
  float a = 0.0f, b = 0.0f, e = 0.0f;
  int64_t n_lo = 0, n_hi = outer, i_lo = 0, i_hi = inner;
  if (channels == 1 && outer == 1) {
    i_lo = sg * ipseg;
    i_hi = (i_lo + ipseg < inner) ? i_lo + ipseg : inner;
  } else {
    n_lo = sg * npseg;
    n_hi = (n_lo + npseg < outer) ? n_lo + npseg : outer;
  }
  for (int64_t n = n_lo; n < n_hi; ++n) {
    const int64_t base = (n * channels + c) * inner;
    for (int64_t i = i_lo + threadIdx.x; i < i_hi; i += DLMCQ_BLOCK) {
      const float u = o[base + i], v = oq[base + i], d = u - v;
      a += u * v;
      b += v * v + 1e-7f;
      e += d * d;
    }
  }
  __shared__ float part[3][DLMCQ_BLOCK / DLMCQ_WAVE];
#pragma unroll
  for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, DLMCQ_WAVE);
    b += __shfl_xor(b, off, DLMCQ_WAVE);
    e += __shfl_xor(e, off, DLMCQ_WAVE);
  }
  if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) {
    part[0][threadIdx.x / DLMCQ_WAVE] = a;
    part[1][threadIdx.x / DLMCQ_WAVE] = b;
    part[2][threadIdx.x / DLMCQ_WAVE] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = partials + (sg * channels + c) * 3;
    p[0] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
    p[1] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    p[2] = part[2][0] + part[2][1] + part[2][2] + part[2][3];
  }
}

// ---- shrink search per tensor (ops.py:36-68): all 80 candidates in ONE read of x ----
constexpr int L2L_STEPS = 80;

__device__ __forceinline__ float shrink_of(int i) { return (float)(1.0 - 0.01 * (double)i); }   // the reference's Python double, cast by the multiply

__global__ void l2loss_cands_kernel(const float* __restrict__ vmax, const float* __restrict__ vmin, float qmax,
                                    float* __restrict__ cand /* [2][80]: scale, zero point */) {
  const int i = threadIdx.x;
  if (i >= L2L_STEPS) return;
  const float sh = shrink_of(i);
  const float nmax = sh * vmax[0], nmin = sh * (vmin ? vmin[0] : 0.0f);
  const float s = (nmax - nmin) / qmax;
  cand[i] = s;
  cand[L2L_STEPS + i] = __builtin_rintf(-nmin / s);
}

__global__ __launch_bounds__(DLMCQ_BLOCK) void l2loss_search_kernel(const float* __restrict__ x, int64_t n, int64_t per_block,
                                                                   const float* __restrict__ cand, float qmax,
                                                                   float* __restrict__ partials /* [blocks][80] */) {
  __shared__ float cs[2 * L2L_STEPS];
  __shared__ float red[DLMCQ_BLOCK / DLMCQ_WAVE][L2L_STEPS];
  if (threadIdx.x < 2 * L2L_STEPS) cs[threadIdx.x] = cand[threadIdx.x];
  __syncthreads();
  float acc[L2L_STEPS];
#pragma unroll
  for (int i = 0; i < L2L_STEPS; ++i) acc[i] = 0.0f;
  const int64_t lo = (int64_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
  for (int64_t k = lo + threadIdx.x; k < hi; k += DLMCQ_BLOCK) {
    const float v = x[k];
#pragma unroll
    for (int i = 0; i < L2L_STEPS; ++i) {
      const float s = cs[i], z = cs[L2L_STEPS + i];
      float q = __builtin_rintf(v / s) + z;                       // ops.py:59-60
      q = clamp_nan(q, 0.0f, qmax);
      const float d = (q - z) * s - v;
      acc[i] += d * d;
    }
  }
#pragma unroll
  for (int i = 0; i < L2L_STEPS; ++i) {
    float t = acc[i];
#pragma unroll
    for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, DLMCQ_WAVE);
    if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) red[threadIdx.x / DLMCQ_WAVE][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < L2L_STEPS)
    partials[(int64_t)blockIdx.x * L2L_STEPS + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void l2loss_pick_kernel(const float* __restrict__ partials, int64_t nblocks, const float* __restrict__ cand,
                                   const float* __restrict__ vmax, float qmax, float loss_div, float* __restrict__ scale,
                                   float* __restrict__ offset) {
  __shared__ float loss[L2L_STEPS];
  const int i = threadIdx.x;
  if (i < L2L_STEPS) {
    double t = 0.0;
    for (int64_t b = 0; b < nblocks; ++b) t += (double)partials[b * L2L_STEPS + i];
    loss[i] = (float)(t / (double)loss_div);      // l2_loss (trainer/loss/loss.py:22-24): sum over axis 1, mean over the rest
  }
  __syncthreads();
  if (i == 0) {
    float min_loss = 1000.0f, s = vmax[0] / qmax, o = 0.0f;     // ops.py:48-50
    for (int k = 0; k < L2L_STEPS; ++k)
      if (loss[k] < min_loss) {
        min_loss = loss[k];
        s = cand[k];
        o = cand[L2L_STEPS + k];
      }
    scale[0] = s;
    offset[0] = o;
  }
}

// ---- shrink search per channel (ops.py:169-196): one workgroup per channel, its row cached in LDS, the 80 steps in order.
// The reference's aliasing quirk is kept: `min_val` IS `offset`, so once a step is accepted the following candidates shrink
// the accepted ZERO POINT, not the channel minimum - each candidate depends on the outcome of the previous ones. ----
constexpr int L2L_ROW_LDS = 8192;    // floats of a row kept on chip (longer rows are re-read from memory)

__global__ __launch_bounds__(DLMCQ_BLOCK) void l2loss_rows_kernel(const float* __restrict__ x, int64_t inner, float qmax,
                                                                 float* __restrict__ scale, float* __restrict__ offset) {
  __shared__ float row[L2L_ROW_LDS];
  __shared__ float red[DLMCQ_BLOCK / DLMCQ_WAVE];
  const int64_t c = blockIdx.x;
  const float* __restrict__ src = x + c * inner;
  const bool cached = inner <= L2L_ROW_LDS;
  if (cached)
    for (int64_t k = threadIdx.x; k < inner; k += DLMCQ_BLOCK) row[k] = src[k];
  __syncthreads();
  float s_best = scale[c], off = offset[c];         // `off` doubles as min_val (the alias)
  const float max_val = off + s_best * qmax;        // ops.py:172 (computed once, before the loop)
  float min_loss = 1000.0f;
  for (int i = 0; i < L2L_STEPS; ++i) {
    const float sh = shrink_of(i);
    const float nmin = sh * off, nmax = sh * max_val;
    const float s = (nmax - nmin) / qmax;
    const float z = __builtin_rintf(-nmin / s);
    float t = 0.0f;
    for (int64_t k = threadIdx.x; k < inner; k += DLMCQ_BLOCK) {
      const float v = cached ? row[k] : src[k];
      float q = clamp_nan(__builtin_rintf(v / s) + z, 0.0f, qmax);
      const float d = v - (q - z) * s;
      t += d * d;
    }
#pragma unroll
    for (int o2 = DLMCQ_WAVE / 2; o2 > 0; o2 >>= 1) t += __shfl_xor(t, o2, DLMCQ_WAVE);
    __syncthreads();
    if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) red[threadIdx.x / DLMCQ_WAVE] = t;
    __syncthreads();
    const float loss = red[0] + red[1] + red[2] + red[3];     // every thread sees the same value: uniform decisions
    if (min_loss > loss) {
      s_best = s;
      off = z;
      min_loss = loss;
    }
  }
  if (threadIdx.x == 0) {
    scale[c] = s_best;
    offset[c] = off;
  }
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


// -------------------------------------------------------------------------------- entry points of the additions above
extern "C" int dlmcq_l2norm_iterate_f32(const float* x, float* scale, const float* offset, float* state, int64_t outer,
                                        int64_t channels, int64_t inner, int32_t lo, int32_t hi, int32_t iterations, float eps,
                                        void* scratch, size_t scratch_bytes, dlmcq_stream_t stream) {
  if (outer < 1 || channels < 1 || inner < 1 || lo > hi || iterations < 1) return DLMCQ_EINVAL;
  if (!x || !scale || !state) return DLMCQ_EINVAL;
  if (channels >= (1ll << 31)) return DLMCQ_ERANGE;
  if (channels == 1 && outer > 1) {
    inner *= outer;
    outer = 1;
  }
  const L2Plan p = l2_plan(outer, channels, inner);
  if (!scratch || scratch_bytes < (size_t)(p.nseg * channels * 2) * sizeof(float)) return DLMCQ_ESCRATCH;
  float* part = reinterpret_cast<float*>(scratch);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int it = 0; it < iterations; ++it) {
    hipLaunchKernelGGL(l2_step_guarded_kernel, dim3((uint32_t)channels, (uint32_t)p.nseg), dim3(DLMCQ_BLOCK), 0, st, x, scale, offset,
                       outer, channels, inner, p.npseg, p.ipseg, (float)lo, (float)hi, part, state);
    hipLaunchKernelGGL(l2_update_kernel, dim3(1), dim3(DLMCQ_BLOCK), 0, st, part, p.nseg, channels, 2, scale, (float*)nullptr, state, 0,
                       1.0f, eps);
  }
  return launch_status();
}

extern "C" size_t dlmcq_l2out_scratch_bytes(int64_t outer, int64_t channels, int64_t inner) {
  if (outer < 1 || channels < 1 || inner < 1) return 0;
  // enough for either mode of dlmcq_l2out_update_f32: per channel (one row per channel) or per tensor (ONE flat row, whose
  // segment count can exceed the per-channel plan's: batch 4 x 16 channels x 224 x 224 needs 588 floats against 192)
  const L2Plan pc = l2_plan(outer, channels, inner), pt = l2_plan(1, 1, outer * channels * inner);
  const int64_t a = pc.nseg * channels * 3, b = pt.nseg * 3;
  return (size_t)(a > b ? a : b) * sizeof(float);
}

extern "C" int dlmcq_l2out_update_f32(const float* out, const float* out_q, float* scale, float* best_scale, float* state,
                                      int64_t outer, int64_t channels, int64_t inner, int32_t per_channel, float mse_div, float eps,
                                      void* scratch, size_t scratch_bytes, dlmcq_stream_t stream) {
  if (outer < 1 || channels < 1 || inner < 1) return DLMCQ_EINVAL;
  if (!out || !out_q || !scale || !best_scale || !state) return DLMCQ_EINVAL;
  if (channels >= (1ll << 31)) return DLMCQ_ERANGE;
  const int64_t nch = per_channel ? channels : 1;
  int64_t o2 = outer, c2 = channels, i2 = inner;
  if (!per_channel) {            // one scale for the whole tensor: a single flat row
    i2 = outer * channels * inner;
    o2 = c2 = 1;
  }
  const L2Plan p = l2_plan(o2, c2, i2);
  if (!scratch || scratch_bytes < (size_t)(p.nseg * nch * 3) * sizeof(float)) return DLMCQ_ESCRATCH;
  float* part = reinterpret_cast<float*>(scratch);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(l2out_sums_kernel, dim3((uint32_t)c2, (uint32_t)p.nseg), dim3(DLMCQ_BLOCK), 0, st, out, out_q, o2, c2, i2, p.npseg,
                     p.ipseg, part, state);
  hipLaunchKernelGGL(l2_update_kernel, dim3(1), dim3(DLMCQ_BLOCK), 0, st, part, p.nseg, nch, 3, scale, best_scale, state,
                     per_channel ? 2 : 1, mse_div, eps);
  return launch_status();
}

extern "C" size_t dlmcq_l2loss_scratch_bytes(int64_t n) {
  if (n < 1) return 0;
  int64_t blocks = (n + 16383) / 16384;
  if (blocks > DLMCQ_CUS * 8) blocks = DLMCQ_CUS * 8;
  return (size_t)((blocks + 2) * L2L_STEPS) * sizeof(float);
}

extern "C" int dlmcq_l2loss_tensor_f32(const float* x, const float* vmax, const float* vmin, float* scale, float* offset, int64_t n,
                                       int32_t n_bits, float loss_div, void* scratch, size_t scratch_bytes,
                                       dlmcq_stream_t stream) {
  if (n < 1 || n_bits < 1 || n_bits > 16 || !(loss_div > 0.0f)) return DLMCQ_EINVAL;
  if (!x || !vmax || !scale || !offset) return DLMCQ_EINVAL;
  int64_t blocks = (n + 16383) / 16384;
  if (blocks > DLMCQ_CUS * 8) blocks = DLMCQ_CUS * 8;
  const int64_t per_block = (n + blocks - 1) / blocks;
  if (!scratch || scratch_bytes < (size_t)((blocks + 2) * L2L_STEPS) * sizeof(float)) return DLMCQ_ESCRATCH;
  float* cand = reinterpret_cast<float*>(scratch);
  float* part = cand + 2 * L2L_STEPS;
  const float qmax = (float)((1 << n_bits) - 1);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(l2loss_cands_kernel, dim3(1), dim3(128), 0, st, vmax, vmin, qmax, cand);
  hipLaunchKernelGGL(l2loss_search_kernel, dim3((uint32_t)blocks), dim3(DLMCQ_BLOCK), 0, st, x, n, per_block, cand, qmax, part);
  hipLaunchKernelGGL(l2loss_pick_kernel, dim3(1), dim3(128), 0, st, part, blocks, cand, vmax, qmax, loss_div, scale, offset);
  return launch_status();
}

extern "C" int dlmcq_l2loss_rows_f32(const float* x, float* scale, float* offset, int64_t rows, int64_t inner, int32_t n_bits,
                                     dlmcq_stream_t stream) {
  if (rows < 1 || inner < 1 || n_bits < 1 || n_bits > 16) return DLMCQ_EINVAL;
  if (!x || !scale || !offset) return DLMCQ_EINVAL;
  if (rows >= (1ll << 31)) return DLMCQ_ERANGE;
  hipLaunchKernelGGL(l2loss_rows_kernel, dim3((uint32_t)rows), dim3(DLMCQ_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), x, inner,
                     (float)((1 << n_bits) - 1), scale, offset);
  return launch_status();
}
