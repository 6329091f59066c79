// Observer min/max for gfx950 (MI355X): ONE read of the tensor (4 B per element), HBM-bound.
//
// Replaces ops.py:20-34 (`abs()` copy + `max`, or `min` + `max`: 2 passes) and ops.py:112-140
// (transpose COPY + abs + reduce: 3-4 passes) with a two-launch deterministic reduction:
//   stage 1  128-bit loads, U independent loads in flight per lane, a running (max, min) per lane,
//            wave64 butterfly (__shfl_xor) -> LDS across the 4 waves -> one partial per block/team;
//   stage 2  a small finalize over the partials that can also apply the scale/offset arithmetic,
//            so the observer never synchronises with the host.
// No atomics: the result is independent of scheduling and of the launch geometry (max/min are exact).
// NaN propagates like torch.max/min: every lane also keeps the unsigned max of |x|'s bit pattern
// (NaN > inf as an integer), which costs one v_and_or/v_max_u32 pair and avoids compare-select chains.
//
// Per-channel layout (outer, channels, inner) is reduced IN PLACE: a "team" (one wave, or the whole
// 256-thread block for long rows) owns channel c and walks rows (n, c) for its share of n, so the
// NCHW activation is never transposed or copied.
#include "dlmcq_internal.h"

namespace dlmcq {

struct Acc {
  float mx;       // running max of x (or of |x| in ABSMAX mode, kept in `ab` only)
  float mn;       // running min of x
  uint32_t ab;    // running unsigned max of bits(|x|): absmax, and the NaN detector
};

template <int MODE>
__device__ __forceinline__ void acc_init(Acc& a) {
  a.mx = -__builtin_inff();
  a.mn = __builtin_inff();
  a.ab = 0u;
}

template <int MODE>
__device__ __forceinline__ void acc_add(Acc& a, float v) {
  a.ab = max(a.ab, __float_as_uint(v) & 0x7fffffffu);
  if (MODE != DLMCQ_MINMAX_ABSMAX) {
    a.mx = __builtin_fmaxf(a.mx, v);  // v_max_f32: NaN operands are dropped here and caught via `ab`
    a.mn = __builtin_fminf(a.mn, v);
  }
}

template <int MODE>
__device__ __forceinline__ void acc_merge(Acc& a, const Acc& b) {
  a.ab = max(a.ab, b.ab);
  if (MODE != DLMCQ_MINMAX_ABSMAX) {
    a.mx = __builtin_fmaxf(a.mx, b.mx);
    a.mn = __builtin_fminf(a.mn, b.mn);
  }
}

template <int MODE>
__device__ __forceinline__ void acc_add4(Acc& a, const f32x4& v) {
  acc_add<MODE>(a, v.x);
  acc_add<MODE>(a, v.y);
  acc_add<MODE>(a, v.z);
  acc_add<MODE>(a, v.w);
}

template <int MODE>
__device__ __forceinline__ Acc wave_reduce(Acc a) {
#pragma unroll
  for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) {
    Acc b;
    b.ab = __shfl_xor(a.ab, off, DLMCQ_WAVE);
    if (MODE != DLMCQ_MINMAX_ABSMAX) {
      b.mx = __shfl_xor(a.mx, off, DLMCQ_WAVE);
      b.mn = __shfl_xor(a.mn, off, DLMCQ_WAVE);
    }
    acc_merge<MODE>(a, b);
  }
  return a;
}

// Across the 4 waves of a block through LDS; valid in every thread of wave 0 afterwards.
template <int MODE>
__device__ __forceinline__ Acc block_reduce(Acc a) {
  __shared__ Acc part[DLMCQ_BLOCK / DLMCQ_WAVE];
  a = wave_reduce<MODE>(a);
  const int w = threadIdx.x / DLMCQ_WAVE;
  if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) part[w] = a;
  __syncthreads();
  if (w == 0) {
    a = part[0];
#pragma unroll
    for (int k = 1; k < DLMCQ_BLOCK / DLMCQ_WAVE; ++k) acc_merge<MODE>(a, part[k]);
  }
  return a;
}

// Partials are stored as three planes [max | min | absbits], `np` entries each.
__device__ __forceinline__ void store_partial(float* scratch, int64_t np, int64_t i, const Acc& a) {
  scratch[i] = a.mx;
  scratch[np + i] = a.mn;
  reinterpret_cast<uint32_t*>(scratch)[2 * np + i] = a.ab;
}

// ----------------------------------------------------------------------- stage 1, per tensor
// Geometry from tools/tune_fq.hip: one-wave workgroups, 4 loads in flight per lane,
// a persistent grid of 2048 workgroups (= 2048 partials for stage 2) - 6.1 TB/s at 196 MiB, 6.9 at 784 MiB.
constexpr int MMT_BLOCK = DLMCQ_WAVE;

template <int MODE, int U>
__global__ __launch_bounds__(MMT_BLOCK) void minmax_tensor_kernel(const float* __restrict__ x, int64_t n,
                                                                 float* __restrict__ scratch) {
  Acc a;
  acc_init<MODE>(a);
  const int64_t n4 = n >> 2;
  const int64_t nchunks = (n4 + MMT_BLOCK * U - 1) / (MMT_BLOCK * U);
  const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
  for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const int64_t i0 = chunk * (MMT_BLOCK * U) + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * MMT_BLOCK;
      if (i < n4) v[u] = __builtin_nontemporal_load(x4 + i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * MMT_BLOCK;
      if (i < n4) acc_add4<MODE>(a, v[u]);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc_add<MODE>(a, x[(n4 << 2) + threadIdx.x]);
  a = wave_reduce<MODE>(a);
  if (threadIdx.x == 0) store_partial(scratch, gridDim.x, blockIdx.x, a);
}

// Unaligned per-tensor input: dword loads (fallback).
template <int MODE>
__global__ __launch_bounds__(DLMCQ_BLOCK) void minmax_tensor_scalar_kernel(const float* __restrict__ x, int64_t n,
                                                                          float* __restrict__ scratch) {
  Acc a;
  acc_init<MODE>(a);
  for (int64_t i = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * DLMCQ_BLOCK)
    acc_add<MODE>(a, x[i]);
  a = block_reduce<MODE>(a);
  if (threadIdx.x == 0) store_partial(scratch, gridDim.x, blockIdx.x, a);
}

// ---------------------------------------------------------------------- stage 1, per channel
// Team t of segment sg owns channel c = t and the rows (n, c) for n in [sg*npseg, (sg+1)*npseg).
// TEAM = 64 (one wave per channel; a block covers 4 adjacent channels = adjacent memory) for short
// rows, TEAM = 256 for long rows.  VEC: inner % 4 == 0 and x 16-byte aligned -> every row start is
// 16-byte aligned.  Partials: planes [max|min|abs] of [nseg][channels].
template <int MODE, int TEAM, bool VEC>
__global__ __launch_bounds__(DLMCQ_BLOCK) void minmax_rows_kernel(const float* __restrict__ x, int64_t outer,
                                                                 int64_t channels, int64_t inner, int64_t npseg,
                                                                 float* __restrict__ scratch) {
  constexpr int TPB = DLMCQ_BLOCK / TEAM;  // teams per block
  const int team = threadIdx.x / TEAM;
  const int lane = threadIdx.x % TEAM;
  const int64_t c = (int64_t)blockIdx.x * TPB + team;
  const int64_t sg = blockIdx.y;
  const int64_t nseg = gridDim.y;
  Acc a;
  acc_init<MODE>(a);
  if (c < channels) {
    const int64_t n_lo = sg * npseg;
    const int64_t n_hi = (n_lo + npseg < outer) ? n_lo + npseg : outer;
    const int64_t rstride = channels * inner;
    if (VEC) {
      const int64_t i4 = inner >> 2;
      if (i4 <= TEAM) {
        // at most one float4 per lane per row: unroll across rows so 4 loads are in flight
        const bool on = lane < i4;
        for (int64_t n = n_lo; n < n_hi; n += 4) {
          f32x4 v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (on && n + k < n_hi)
              v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (n + k) * rstride + c * inner) + lane);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (on && n + k < n_hi) acc_add4<MODE>(a, v[k]);
        }
      } else {
        constexpr int UF = TEAM == DLMCQ_WAVE ? 8 : 4;     // loads in flight per lane (one-wave teams: deeper)
        for (int64_t n = n_lo; n < n_hi; ++n) {
          const f32x4* __restrict__ r4 = reinterpret_cast<const f32x4*>(x + n * rstride + c * inner);
          for (int64_t i = lane; i < i4; i += UF * TEAM) {
            f32x4 v[UF];
#pragma unroll
            for (int k = 0; k < UF; ++k)
              if (i + k * TEAM < i4) v[k] = __builtin_nontemporal_load(r4 + i + k * TEAM);
#pragma unroll
            for (int k = 0; k < UF; ++k)
              if (i + k * TEAM < i4) acc_add4<MODE>(a, v[k]);
          }
        }
      }
    } else {
      if (inner <= TEAM) {
        const bool on = lane < inner;
        for (int64_t n = n_lo; n < n_hi; n += 4) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (on && n + k < n_hi) v[k] = x[(n + k) * rstride + c * inner + lane];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (on && n + k < n_hi) acc_add<MODE>(a, v[k]);
        }
      } else {
        for (int64_t n = n_lo; n < n_hi; ++n) {
          const float* __restrict__ r = x + n * rstride + c * inner;
          for (int64_t i = lane; i < inner; i += 4 * TEAM) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (i + k * TEAM < inner) v[k] = r[i + k * TEAM];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (i + k * TEAM < inner) acc_add<MODE>(a, v[k]);
          }
        }
      }
    }
  }
  if (TEAM == DLMCQ_WAVE) {
    a = wave_reduce<MODE>(a);
    if (lane == 0 && c < channels) store_partial(scratch, nseg * channels, sg * channels + c, a);
  } else {
    a = block_reduce<MODE>(a);
    if (threadIdx.x == 0 && c < channels) store_partial(scratch, nseg * channels, sg * channels + c, a);
  }
}

// ------------------------------------------------------------------------------------ stage 2
struct Finalize {
  float* out_a;      // out_max or scale
  float* out_b;      // out_min or offset
  int qparams;       // 0: write (max, min) per `mode`; 1: write (scale, offset)
  int mode;          // DLMCQ_MINMAX_*
  float qmax;        // 2^(b-1)-1 (signed) or 2^b-1 (unsigned)
  int is_signed;
  int allow_offset;
  float scale_eps;
};

__device__ __forceinline__ void qparams_tail(float vmax, float vmin, float absmax, const Finalize& f, float& scale,
                                             float& offset) {
  if (f.is_signed) {           // ops.py:22-24 / :125-127
    scale = absmax / f.qmax;
    offset = 0.0f;
  } else {                     // ops.py:26-33 / :129-136
    const float mn = f.allow_offset ? vmin : 0.0f;
    scale = (vmax - mn) / f.qmax;
    offset = mn;
  }
  if (f.scale_eps != 0.0f) scale = scale + f.scale_eps;
}

__device__ __forceinline__ void finalize_write(const Acc& a, int64_t c, const Finalize& f) {
  const float nan = __builtin_nanf("");
  const bool has_nan = a.ab > 0x7f800000u;
  const float absmax = __uint_as_float(a.ab);  // NaN bits if any NaN was seen
  const float vmax = has_nan ? nan : a.mx;
  const float vmin = has_nan ? nan : a.mn;
  if (f.qparams) {
    float s, o;
    qparams_tail(vmax, vmin, absmax, f, s, o);
    f.out_a[c] = s;
    f.out_b[c] = o;
  } else if (f.mode == DLMCQ_MINMAX_ABSMAX) {
    f.out_a[c] = absmax;
  } else {
    f.out_a[c] = vmax;
    f.out_b[c] = f.mode == DLMCQ_MINMAX_NEGMIN ? -vmin : vmin;
  }
}

// Per tensor: one block reduces `np` block partials.
__global__ __launch_bounds__(DLMCQ_BLOCK) void finalize_tensor_kernel(const float* __restrict__ scratch, int64_t np,
                                                                     Finalize f) {
  Acc a;
  acc_init<DLMCQ_MINMAX_MINMAX>(a);
  for (int64_t i = threadIdx.x; i < np; i += DLMCQ_BLOCK) {
    Acc b;
    b.mx = scratch[i];
    b.mn = scratch[np + i];
    b.ab = reinterpret_cast<const uint32_t*>(scratch)[2 * np + i];
    acc_merge<DLMCQ_MINMAX_MINMAX>(a, b);
  }
  a = block_reduce<DLMCQ_MINMAX_MINMAX>(a);
  if (threadIdx.x == 0) finalize_write(a, 0, f);
}

// ... the same over `np` partials whose three planes are `stride` entries apart (dlmcq_minmax_finalize_f32: partials written by the
// epilogue of a convolution launch, one per workgroup)
__global__ __launch_bounds__(DLMCQ_BLOCK) void finalize_planes_kernel(const float* __restrict__ scratch, int64_t np, int64_t stride,
                                                                     Finalize f) {
  Acc a;
  acc_init<DLMCQ_MINMAX_MINMAX>(a);
  for (int64_t i = threadIdx.x; i < np; i += DLMCQ_BLOCK) {
    Acc b;
    b.mx = scratch[i];
    b.mn = scratch[stride + i];
    b.ab = reinterpret_cast<const uint32_t*>(scratch)[2 * stride + i];
    acc_merge<DLMCQ_MINMAX_MINMAX>(a, b);
  }
  a = block_reduce<DLMCQ_MINMAX_MINMAX>(a);
  if (threadIdx.x == 0) finalize_write(a, 0, f);
}

// Per channel: thread c folds its nseg partials (coalesced across c).
__global__ __launch_bounds__(DLMCQ_BLOCK) void finalize_rows_kernel(const float* __restrict__ scratch, int64_t nseg,
                                                                   int64_t channels, Finalize f) {
  const int64_t c = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x;
  if (c >= channels) return;
  const int64_t np = nseg * channels;
  Acc a;
  acc_init<DLMCQ_MINMAX_MINMAX>(a);
  for (int64_t s = 0; s < nseg; ++s) {
    Acc b;
    b.mx = scratch[s * channels + c];
    b.mn = scratch[np + s * channels + c];
    b.ab = reinterpret_cast<const uint32_t*>(scratch)[2 * np + s * channels + c];
    acc_merge<DLMCQ_MINMAX_MINMAX>(a, b);
  }
  finalize_write(a, c, f);
}

// Stand-alone tail (after an all-reduce of [max | -min] across ranks).
__global__ __launch_bounds__(DLMCQ_BLOCK) void qparams_kernel(const float* __restrict__ vmax,
                                                             const float* __restrict__ vmin, int64_t channels,
                                                             int min_is_negated, Finalize f) {
  const int64_t c = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x;
  if (c >= channels) return;
  const float mx = vmax[c];
  float mn = vmin ? vmin[c] : 0.0f;
  if (min_is_negated) mn = -mn;
  float s, o;
  qparams_tail(mx, mn, mx, f, s, o);
  f.out_a[c] = s;
  if (f.out_b) f.out_b[c] = o;
}

// ---------------------------------------------------------------------------------- host side
constexpr int MM_U = 4;
constexpr int MM_TENSOR_BLOCKS = 2048;  // stage-1 grid cap, per tensor (persistent one-wave workgroups)
constexpr int64_t MM_TARGET_TEAMS = DLMCQ_CUS * 32;  // per-channel: aim for this many teams in flight

struct Plan {
  bool per_tensor;
  int grid_x, grid_y;  // stage 1
  int team;            // 64 or 256 (per channel)
  int64_t npseg;       // rows of `outer` per segment (per channel)
  int64_t partials;    // entries per plane
};

static Plan make_plan(int64_t outer, int64_t channels, int64_t inner) {
  Plan p{};
  const int64_t n = outer * channels * inner;
  if (channels == 1) {
    p.per_tensor = true;
    int64_t b = ((n >> 2) + MMT_BLOCK * MM_U - 1) / (MMT_BLOCK * MM_U);
    if (b < 1) b = 1;
    if (b > MM_TENSOR_BLOCKS) b = MM_TENSOR_BLOCKS;
    p.grid_x = (int)b;
    p.grid_y = 1;
    p.partials = b;
    return p;
  }
  p.per_tensor = false;
  p.team = (inner <= 16384) ? DLMCQ_WAVE : DLMCQ_BLOCK;   // one-wave teams stream best (tools/tune_fq.hip); whole blocks only for very long rows
  const int tpb = DLMCQ_BLOCK / p.team;
  p.grid_x = (int)((channels + tpb - 1) / tpb);
  // split `outer` so that enough teams exist to fill the chip, at >= 1 row per segment
  int64_t nseg = (MM_TARGET_TEAMS + channels - 1) / channels;
  if (nseg > outer) nseg = outer;
  if (nseg < 1) nseg = 1;
  if (nseg > 65535) nseg = 65535;
  p.npseg = (outer + nseg - 1) / nseg;
  nseg = (outer + p.npseg - 1) / p.npseg;
  p.grid_y = (int)nseg;
  p.partials = nseg * channels;
  return p;
}

template <int MODE>
static int launch_stage1(const float* x, int64_t outer, int64_t channels, int64_t inner, const Plan& p, float* scratch,
                         hipStream_t st) {
  const int64_t n = outer * channels * inner;
  if (p.per_tensor) {
    if (aligned16(x))
      hipLaunchKernelGGL((minmax_tensor_kernel<MODE, MM_U>), dim3(p.grid_x), dim3(MMT_BLOCK), 0, st, x, n, scratch);
    else
      hipLaunchKernelGGL((minmax_tensor_scalar_kernel<MODE>), dim3(p.grid_x), dim3(DLMCQ_BLOCK), 0, st, x, n, scratch);
    return launch_status();
  }
  const bool vec = aligned16(x) && (inner % 4 == 0);
  const dim3 grid(p.grid_x, p.grid_y);
#define DLMCQ_ROWS(TEAM, VEC)                                                                                    \
  hipLaunchKernelGGL((minmax_rows_kernel<MODE, TEAM, VEC>), grid, dim3(DLMCQ_BLOCK), 0, st, x, outer, channels, \
                     inner, p.npseg, scratch)
  if (p.team == DLMCQ_WAVE) {
    if (vec) DLMCQ_ROWS(DLMCQ_WAVE, true); else DLMCQ_ROWS(DLMCQ_WAVE, false);
  } else {
    if (vec) DLMCQ_ROWS(DLMCQ_BLOCK, true); else DLMCQ_ROWS(DLMCQ_BLOCK, false);
  }
#undef DLMCQ_ROWS
  return launch_status();
}

static int run_observer(const float* x, int64_t outer, int64_t channels, int64_t inner, int mode, const Finalize& f,
                        void* scratch, size_t scratch_bytes, hipStream_t st) {
  const Plan p = make_plan(outer, channels, inner);
  if (!scratch || scratch_bytes < (size_t)p.partials * 3 * sizeof(float)) return DLMCQ_ESCRATCH;
  if (p.grid_x < 1 || channels > (1ll << 31)) return DLMCQ_ERANGE;
  float* sc = reinterpret_cast<float*>(scratch);
  int rc;
  if (mode == DLMCQ_MINMAX_ABSMAX)
    rc = launch_stage1<DLMCQ_MINMAX_ABSMAX>(x, outer, channels, inner, p, sc, st);
  else
    rc = launch_stage1<DLMCQ_MINMAX_MINMAX>(x, outer, channels, inner, p, sc, st);
  if (rc != DLMCQ_OK) return rc;
  if (p.per_tensor) {
    hipLaunchKernelGGL(finalize_tensor_kernel, dim3(1), dim3(DLMCQ_BLOCK), 0, st, sc, p.partials, f);
  } else {
    const int g = (int)((channels + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK);
    hipLaunchKernelGGL(finalize_rows_kernel, dim3(g), dim3(DLMCQ_BLOCK), 0, st, sc, (int64_t)p.grid_y, channels, f);
  }
  return launch_status();
}

static float qmax_of(int n_bits, int is_signed) {
  return is_signed ? (float)((1ll << (n_bits - 1)) - 1) : (float)((1ll << n_bits) - 1);
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" size_t dlmcq_minmax_scratch_bytes(int64_t outer, int64_t channels, int64_t inner) {
  if (outer < 0 || channels < 1 || inner < 0) return 0;
  const Plan p = make_plan(outer, channels, inner);
  return (size_t)p.partials * 3 * sizeof(float);
}

extern "C" int dlmcq_minmax_f32(const float* x, float* out_max, float* out_min, int64_t outer, int64_t channels,
                                int64_t inner, int32_t mode, void* scratch, size_t scratch_bytes,
                                dlmcq_stream_t stream) {
  if (outer < 0 || channels < 1 || inner < 0) return DLMCQ_EINVAL;
  if (mode < DLMCQ_MINMAX_ABSMAX || mode > DLMCQ_MINMAX_NEGMIN) return DLMCQ_EINVAL;
  if (!x || !out_max || (mode != DLMCQ_MINMAX_ABSMAX && !out_min)) return DLMCQ_EINVAL;
  // an empty reduction has no value (torch raises): refuse rather than invent one
  if (outer * channels * inner == 0) return DLMCQ_EINVAL;
  Finalize f{};
  f.out_a = out_max;
  f.out_b = out_min;
  f.qparams = 0;
  f.mode = mode;
  return run_observer(x, outer, channels, inner, mode, f, scratch, scratch_bytes, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dlmcq_observe_qparams_f32(const float* x, float* scale, float* offset, int64_t outer, int64_t channels,
                                         int64_t inner, int32_t n_bits, int32_t is_signed, int32_t allow_offset,
                                         float scale_eps, void* scratch, size_t scratch_bytes,
                                         dlmcq_stream_t stream) {
  if (outer < 0 || channels < 1 || inner < 0) return DLMCQ_EINVAL;
  if (n_bits < 1 || n_bits > 24) return DLMCQ_EINVAL;
  if (!x || !scale || !offset) return DLMCQ_EINVAL;
  if (outer * channels * inner == 0) return DLMCQ_EINVAL;
  Finalize f{};
  f.out_a = scale;
  f.out_b = offset;
  f.qparams = 1;
  f.qmax = qmax_of(n_bits, is_signed);
  f.is_signed = is_signed;
  f.allow_offset = allow_offset;
  f.scale_eps = scale_eps;
  const int mode = is_signed ? DLMCQ_MINMAX_ABSMAX : DLMCQ_MINMAX_MINMAX;
  f.mode = mode;
  return run_observer(x, outer, channels, inner, mode, f, scratch, scratch_bytes, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dlmcq_minmax_finalize_f32(const float* partials, int64_t count, int64_t plane_stride, float* out_max, float* out_min,
                                         int32_t mode, dlmcq_stream_t stream) {
  if (count < 1 || plane_stride < count || !partials || !out_max || mode < DLMCQ_MINMAX_ABSMAX || mode > DLMCQ_MINMAX_NEGMIN ||
      (mode != DLMCQ_MINMAX_ABSMAX && !out_min))
    return DLMCQ_EINVAL;
  Finalize f{};
  f.out_a = out_max;
  f.out_b = out_min;
  f.qparams = 0;
  f.mode = mode;
  hipLaunchKernelGGL(finalize_planes_kernel, dim3(1), dim3(DLMCQ_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), partials, count,
                     plane_stride, f);
  return launch_status();
}

extern "C" int dlmcq_qparams_from_minmax(const float* vmax, const float* vmin, float* scale, float* offset,
                                         int64_t channels, int32_t n_bits, int32_t is_signed, int32_t allow_offset,
                                         int32_t min_is_negated, float scale_eps, dlmcq_stream_t stream) {
  if (channels < 1 || n_bits < 1 || n_bits > 24) return DLMCQ_EINVAL;
  if (!vmax || !scale || !offset || (!is_signed && allow_offset && !vmin)) return DLMCQ_EINVAL;
  Finalize f{};
  f.out_a = scale;
  f.out_b = offset;
  f.qparams = 1;
  f.qmax = qmax_of(n_bits, is_signed);
  f.is_signed = is_signed;
  f.allow_offset = allow_offset;
  f.scale_eps = scale_eps;
  const int g = (int)((channels + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK);
  hipLaunchKernelGGL(qparams_kernel, dim3(g), dim3(DLMCQ_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), vmax, vmin,
                     channels, min_is_negated, f);
  return launch_status();
}

// ---- LSQ initialisation: 2 * mean|x| / sqrt(Qp) (modules/base.py:84-85,118-121) ----
namespace dlmcq {
constexpr int LSQ_WGS = 2048;   // one-wave workgroups, persistent over the tensor

__global__ __launch_bounds__(64) void abs_sum_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 64;
  double acc = 0.0;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = __builtin_nontemporal_load(x4 + i);
    acc += (double)__builtin_fabsf(v.x) + (double)__builtin_fabsf(v.y) + (double)__builtin_fabsf(v.z) + (double)__builtin_fabsf(v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc += (double)__builtin_fabsf(x[(n4 << 2) + threadIdx.x]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(DLMCQ_BLOCK) void lsq_finalize_kernel(const double* __restrict__ part, int np, int64_t n, float sqrt_qmax,
                                                                   float* __restrict__ scale) {
  __shared__ double sh[DLMCQ_BLOCK];
  double a = 0.0;
  for (int i = threadIdx.x; i < np; i += DLMCQ_BLOCK) a += part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = DLMCQ_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float mean = (float)sh[0] / (float)n;     // x.abs().mean(): fp32 sum / numel
    scale[0] = (2.0f * mean) / sqrt_qmax;           // 2 * mean / math.sqrt(Qp): true division (tensor / python scalar on the CPU)
  }
}
}  // namespace dlmcq

extern "C" size_t dlmcq_lsq_init_scratch_bytes(int64_t n) { return n < 0 ? 0 : (size_t)LSQ_WGS * sizeof(double); }

extern "C" int dlmcq_lsq_init_f32(const float* x, float* scale, int64_t n, float sqrt_qmax, void* scratch, size_t scratch_bytes,
                                  dlmcq_stream_t stream) {
  if (n < 1 || !x || !scale || !(sqrt_qmax > 0.0f)) return DLMCQ_EINVAL;   // (the mean of nothing has no value: torch gives NaN, we refuse)
  if (!scratch || scratch_bytes < (size_t)LSQ_WGS * sizeof(double)) return DLMCQ_ESCRATCH;
  if (!aligned16(x) || (((uintptr_t)scratch) & 7u)) return DLMCQ_EALIGN;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t want = ((n >> 2) + 63) / 64;
  const int np = (int)(want < 1 ? 1 : (want > LSQ_WGS ? LSQ_WGS : want));
  hipLaunchKernelGGL(abs_sum_kernel, dim3(np), dim3(64), 0, st, x, n, static_cast<double*>(scratch));
  hipLaunchKernelGGL(lsq_finalize_kernel, dim3(1), dim3(DLMCQ_BLOCK), 0, st, static_cast<const double*>(scratch), np, n, sqrt_qmax, scale);
  return launch_status();
}

extern "C" int dlmcq_span_scale_f32(const float* vmax, const float* vmin, float* scale, int64_t channels, float span,
                                    int32_t min_is_negated, dlmcq_stream_t stream) {
  if (channels < 1 || !(span > 0.0f)) return DLMCQ_EINVAL;
  if (!vmax || !vmin || !scale) return DLMCQ_EINVAL;
  Finalize f{};
  f.out_a = scale;
  f.out_b = nullptr;
  f.qparams = 1;
  f.qmax = span;
  f.is_signed = 0;
  f.allow_offset = 1;
  const int g = (int)((channels + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK);
  hipLaunchKernelGGL(qparams_kernel, dim3(g), dim3(DLMCQ_BLOCK), 0, reinterpret_cast<hipStream_t>(stream), vmax, vmin,
                     channels, min_is_negated, f);
  return launch_status();
}
