// Backward of the QBase fake-quant (DLMCQ_FORM_QBASE) for QAT: one pass, HBM-bound.
// 8 B read (x, gy) + 4 B written (gx) per element, plus a deterministic tree sum for the scale
// gradient (block partials in fp32, final fold in fp64; no atomics, so two runs agree bit for bit).
//
// Specification = what autograd executes through modules/base.py:96-102 (closed form in
// modules/function.py:37-49); with v = (x - o)/s^ and inside = [lo <= v <= hi]:
//   gx    = inside ? (gy * s^) / s^ : +0          (two roundings - identical to autograd's mul, div)
//   gs[c] = g * sum( gy*(q - inside*v) ),               q = ste_round(clamp(v))      (= LSQ's closed form)
#include "dlmcq_internal.h"

namespace dlmcq {

struct BwdConst {
  float sh;  // the divisor: s^ (QBASE) or s
  float of;  // QBASE: the offset subtracted before the division;  ZEROPOINT: the zero point added after the rounding;
             // ROOTQ_ACT: the upper clip s*(hi - lo)
  float span;  // ROOTQ_ACT: hi - lo
  int form;
  __device__ __forceinline__ BwdConst(float s, float o, float g, int f, float lo, float hi)
      : sh(f == DLMCQ_FORM_QBASE ? ste_scale(s, g) : s),
        of(f == DLMCQ_FORM_SYMMETRIC ? 0.0f : (f == DLMCQ_FORM_ROOTQ_ACT ? s * (hi - lo) : o)), span(hi - lo), form(f) {}
};

// QBASE:      v = (x - o)/s^,  inside = [lo <= v <= hi],              q = R(clamp(v)),  gs += gy*(q - inside*v)
// ZEROPOINT:  u = x/s, a = R(u) + zp, inside = [lo <= a <= hi], t = clamp(a) - zp,       gs += gy*(t - inside*u)
// SYMMETRIC:  ZEROPOINT with zp = 0     (FSPTQuant/base.py:108-109, 149-152 as autograd runs them: the rounding is a
//             straight-through identity, torch.clamp passes the gradient on the closed interval, x/s gives gx = g/s)
// gx = inside ? (gy*s)/s : +0 in all three - the two roundings autograd performs.
__device__ __forceinline__ void bwd_one(float x, float gy, const BwdConst& c, float lo, float hi, float& gx,
                                        float& contrib) {
  float v, q;
  bool inside;
  if (c.form == DLMCQ_FORM_ROOTQ_ACT) {
    // RootQ/base.py:106-111 + function.py:15-20 as autograd runs them: t1 = x + relu(0 - x), t = t1 - relu(t1 - up),
    // u = t/s, y = R(u)*s.  A clipped element passes no gradient to x (gt - gt = +0); the scale collects gy*R(u) from the
    // product, -gy*u from the division and, through up = s*(hi - lo), gy*(hi - lo) from every element clipped above.
    const float t1 = x + relu_nan(0.0f - x);
    const bool below = (0.0f - x) > 0.0f, above = (t1 - c.of) > 0.0f;
    const float t = t1 - relu_nan(t1 - c.of);
    v = t / c.sh;
    q = ste_round(v);
    const float gv = (below || above) ? 0.0f : gy * c.sh;
    gx = gv / c.sh;
    contrib = gy * (q - v) + (above ? gy * c.span : 0.0f);
    return;
  }
  if (c.form == DLMCQ_FORM_QBASE) {
    v = (x - c.of) / c.sh;
    q = ste_round(clamp_nan(v, lo, hi));
    inside = (v >= lo) && (v <= hi);
  } else {
    v = x / c.sh;
    const float a = ste_round(v) + c.of;
    inside = (a >= lo) && (a <= hi);
    q = clamp_nan(a, lo, hi) - c.of;
  }
  const float gv = inside ? gy * c.sh : 0.0f;
  gx = gv / c.sh;                               // bit-exact with autograd's mul-then-div
  // autograd accumulates gy*q and -gv*(v/s) separately; gv*(v/s) == gy*v up to rounding and the scale
  // gradient is an order-dependent sum anyway, so the third division is not spent: gy*(q - [inside]*v)
  contrib = gy * (q - (inside ? v : 0.0f));
}

__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
  for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, DLMCQ_WAVE);
  return s;
}

__device__ __forceinline__ float block_sum(float s) {
  __shared__ float part[DLMCQ_BLOCK / DLMCQ_WAVE];
  s = wave_sum(s);
  if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) part[threadIdx.x / DLMCQ_WAVE] = s;
  __syncthreads();
  return part[0] + part[1] + part[2] + part[3];
}

// Per tensor.  VEC: all pointers 16-byte aligned.
template <int U, bool VEC>
__global__ __launch_bounds__(DLMCQ_BLOCK) void fq_bwd_tensor_kernel(const float* x, const float* gy, float* gx,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ offset, int64_t n,
                                                                   float lo, float hi, float g, int form,
                                                                   float* __restrict__ partials) {
  const BwdConst c(scale[0], offset ? offset[0] : 0.0f, g, form, lo, hi);
  float acc = 0.0f;
  if (VEC) {
    const int64_t n4 = n >> 2;
    const int64_t nchunks = (n4 + DLMCQ_BLOCK * U - 1) / (DLMCQ_BLOCK * U);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gy);
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
      const int64_t i0 = chunk * (DLMCQ_BLOCK * U) + threadIdx.x;
      f32x4 xv[U], gv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * DLMCQ_BLOCK;
        if (i < n4) {
          xv[u] = __builtin_nontemporal_load(x4 + i);
          gv[u] = __builtin_nontemporal_load(g4 + i);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * DLMCQ_BLOCK;
        if (i < n4) {
          float o0, o1, o2, o3, e0, e1, e2, e3;
          bwd_one(xv[u].x, gv[u].x, c, lo, hi, o0, e0);
          bwd_one(xv[u].y, gv[u].y, c, lo, hi, o1, e1);
          bwd_one(xv[u].z, gv[u].z, c, lo, hi, o2, e2);
          bwd_one(xv[u].w, gv[u].w, c, lo, hi, o3, e3);
          const f32x4 o = {o0, o1, o2, o3};
          acc += (e0 + e1) + (e2 + e3);
          if (gx) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(gx) + i);
        }
      }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
      const int64_t i = (n4 << 2) + threadIdx.x;
      float o, e;
      bwd_one(x[i], gy[i], c, lo, hi, o, e);
      acc += e;
      if (gx) gx[i] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * DLMCQ_BLOCK) {
      float o, e;
      bwd_one(x[i], gy[i], c, lo, hi, o, e);
      acc += e;
      if (gx) gx[i] = o;
    }
  }
  const float s = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// Per channel: block (c, sg) walks rows (n, c) for its share of n (same decomposition as the observer).
template <bool VEC>
__global__ __launch_bounds__(DLMCQ_BLOCK) void fq_bwd_rows_kernel(const float* x, const float* gy, float* gx,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ offset, int64_t outer,
                                                                 int64_t channels, int64_t inner, int64_t npseg,
                                                                 float lo, float hi, float g, int form,
                                                                 float* __restrict__ partials) {
  const int64_t c = blockIdx.x, sg = blockIdx.y;
  const BwdConst k(scale[c], offset ? offset[c] : 0.0f, g, form, lo, hi);
  const int64_t n_lo = sg * npseg;
  const int64_t n_hi = (n_lo + npseg < outer) ? n_lo + npseg : outer;
  float acc = 0.0f;
  for (int64_t n = n_lo; n < n_hi; ++n) {
    const int64_t base = (n * channels + c) * inner;
    if (VEC) {
      const int64_t i4 = inner >> 2;
      for (int64_t i = threadIdx.x; i < i4; i += DLMCQ_BLOCK) {
        const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + base) + i);
        const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gy + base) + i);
        float o0, o1, o2, o3, e0, e1, e2, e3;
        bwd_one(xv.x, gv.x, k, lo, hi, o0, e0);
        bwd_one(xv.y, gv.y, k, lo, hi, o1, e1);
        bwd_one(xv.z, gv.z, k, lo, hi, o2, e2);
        bwd_one(xv.w, gv.w, k, lo, hi, o3, e3);
        const f32x4 o = {o0, o1, o2, o3};
        acc += (e0 + e1) + (e2 + e3);
        if (gx) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(gx + base) + i);
      }
    } else {
      for (int64_t i = threadIdx.x; i < inner; i += DLMCQ_BLOCK) {
        float o, e;
        bwd_one(x[base + i], gy[base + i], k, lo, hi, o, e);
        acc += e;
        if (gx) gx[base + i] = o;
      }
    }
  }
  const float s = block_sum(acc);
  if (threadIdx.x == 0) partials[sg * channels + c] = s;
}

// gscale[c] = g * sum_s partials[s][c], folded in fp64 in a fixed order.
__global__ __launch_bounds__(DLMCQ_BLOCK) void fq_bwd_finalize_kernel(const float* __restrict__ partials, int64_t nseg,
                                                                     int64_t channels, float g,
                                                                     float* __restrict__ gscale) {
  const int64_t c = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x;
  if (c >= channels) return;
  double s = 0.0;
  for (int64_t k = 0; k < nseg; ++k) s += (double)partials[k * channels + c];
  gscale[c] = (float)s * g;
}

// Per tensor: one block folds all workgroup partials (strided fp64 sums, then a tree through LDS).
__global__ __launch_bounds__(DLMCQ_BLOCK) void fq_bwd_finalize_tensor_kernel(const float* __restrict__ partials, int64_t n,
                                                                            float g, float* __restrict__ gscale) {
  __shared__ double red[DLMCQ_BLOCK];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += DLMCQ_BLOCK) s += (double)partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = DLMCQ_BLOCK / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) gscale[0] = (float)red[0] * g;
}

constexpr int BWD_U = 1;                      // one float4 of x and of gy per lane, one chunk per workgroup
constexpr int BWD_TENSOR_BLOCKS = 8192;       // persistent grid: one partial sum per workgroup for the finalize to fold

struct BwdPlan {
  int64_t nseg;   // partial rows
  int64_t npseg;  // per channel only
  int grid_x;
};

static BwdPlan bwd_plan(int64_t outer, int64_t channels, int64_t inner) {
  BwdPlan p{};
  if (channels == 1) {
    int64_t b = ((outer * inner >> 2) + DLMCQ_BLOCK * BWD_U - 1) / (DLMCQ_BLOCK * BWD_U);
    if (b < 1) b = 1;
    if (b > BWD_TENSOR_BLOCKS) b = BWD_TENSOR_BLOCKS;
    p.grid_x = (int)b;
    p.nseg = b;
    return p;
  }
  int64_t nseg = (DLMCQ_CUS * 8 + channels - 1) / channels;
  if (nseg > outer) nseg = outer;
  if (nseg < 1) nseg = 1;
  if (nseg > 65535) nseg = 65535;
  p.npseg = (outer + nseg - 1) / nseg;
  p.nseg = (outer + p.npseg - 1) / p.npseg;
  p.grid_x = (int)channels;
  return p;
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" size_t dlmcq_fq_bwd_scratch_bytes(int64_t outer, int64_t channels, int64_t inner) {
  if (outer < 0 || channels < 1 || inner < 0) return 0;
  const BwdPlan p = bwd_plan(outer, channels, inner);
  return (size_t)(p.nseg * channels) * sizeof(float);
}

extern "C" int dlmcq_fake_quant_bwd_form_f32(const float* x, const float* gy, float* gx, float* gscale, const float* scale,
                                             const float* offset, int64_t outer, int64_t channels, int64_t inner,
                                             int32_t lo, int32_t hi, int32_t form, float ste_g, void* scratch,
                                             size_t scratch_bytes, dlmcq_stream_t stream) {
  if (outer < 0 || channels < 1 || inner < 0 || lo > hi) return DLMCQ_EINVAL;
  if (form != DLMCQ_FORM_QBASE && form != DLMCQ_FORM_ZEROPOINT && form != DLMCQ_FORM_SYMMETRIC && form != DLMCQ_FORM_ROOTQ_ACT)
    return DLMCQ_EINVAL;
  if (form != DLMCQ_FORM_QBASE) ste_g = 1.0f;     // g scales the QBASE scale gradient only
  const int64_t n = outer * channels * inner;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n == 0) {
    if (gscale) (void)hipMemsetAsync(gscale, 0, (size_t)channels * sizeof(float), st);
    return launch_status();
  }
  if (!x || !gy || !scale || (!gx && !gscale)) return DLMCQ_EINVAL;
  if (channels >= (1ll << 31)) return DLMCQ_ERANGE;
  const BwdPlan p = bwd_plan(outer, channels, inner);
  if (!scratch || scratch_bytes < (size_t)(p.nseg * channels) * sizeof(float)) return DLMCQ_ESCRATCH;
  float* part = reinterpret_cast<float*>(scratch);
  const float flo = (float)lo, fhi = (float)hi;
  const bool al = aligned16(x) && aligned16(gy) && (!gx || aligned16(gx));
  if (channels == 1) {
    if (al)
      hipLaunchKernelGGL((fq_bwd_tensor_kernel<BWD_U, true>), dim3(p.grid_x), dim3(DLMCQ_BLOCK), 0, st, x, gy, gx, scale,
                         offset, n, flo, fhi, ste_g, form, part);
    else
      hipLaunchKernelGGL((fq_bwd_tensor_kernel<BWD_U, false>), dim3(p.grid_x), dim3(DLMCQ_BLOCK), 0, st, x, gy, gx,
                         scale, offset, n, flo, fhi, ste_g, form, part);
  } else {
    const dim3 grid(p.grid_x, (uint32_t)p.nseg);
    if (al && inner % 4 == 0)
      hipLaunchKernelGGL((fq_bwd_rows_kernel<true>), grid, dim3(DLMCQ_BLOCK), 0, st, x, gy, gx, scale, offset, outer,
                         channels, inner, p.npseg, flo, fhi, ste_g, form, part);
    else
      hipLaunchKernelGGL((fq_bwd_rows_kernel<false>), grid, dim3(DLMCQ_BLOCK), 0, st, x, gy, gx, scale, offset, outer,
                         channels, inner, p.npseg, flo, fhi, ste_g, form, part);
  }
  int rc = launch_status();
  if (rc != DLMCQ_OK || !gscale) return rc;
  if (channels == 1) {
    hipLaunchKernelGGL(fq_bwd_finalize_tensor_kernel, dim3(1), dim3(DLMCQ_BLOCK), 0, st, part, p.nseg, ste_g, gscale);
    return launch_status();
  }
  const int g = (int)((channels + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK);
  hipLaunchKernelGGL(fq_bwd_finalize_kernel, dim3(g), dim3(DLMCQ_BLOCK), 0, st, part, p.nseg, channels, ste_g, gscale);
  return launch_status();
}

extern "C" int dlmcq_fake_quant_bwd_f32(const float* x, const float* gy, float* gx, float* gscale, const float* scale,
                                        const float* offset, int64_t outer, int64_t channels, int64_t inner,
                                        int32_t lo, int32_t hi, float ste_g, void* scratch, size_t scratch_bytes,
                                        dlmcq_stream_t stream) {
  return dlmcq_fake_quant_bwd_form_f32(x, gy, gx, gscale, scale, offset, outer, channels, inner, lo, hi, DLMCQ_FORM_QBASE, ste_g,
                                       scratch, scratch_bytes, stream);
}
