// Sub-byte pack / unpack (BASELINE config 5): int8 codes <-> two 4-bit codes per byte.
// Layout = DLMCQ_CODES_P4: element 2i in the low nibble, 2i+1 in the high nibble.
// HBM-bound byte shuffling: 1 B read + 0.5 B written per code (pack), the reverse for unpack;
// each lane moves 16 codes per step (one dwordx4 in, one dwordx2 out).
#include "dlmcq_internal.h"

namespace dlmcq {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack8(uint32_t a, uint32_t b) {
  // a = codes 0..3 (one per byte), b = codes 4..7 -> 8 nibbles, code k in nibble k
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r |= ((a >> (8 * k)) & 0xfu) << (4 * k);
    r |= ((b >> (8 * k)) & 0xfu) << (16 + 4 * k);
  }
  return r;
}

__device__ __forceinline__ uint32_t unpack4(uint32_t nib16, int is_signed) {
  // low 16 bits = 4 nibbles -> 4 bytes
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int v = (nib16 >> (4 * k)) & 0xf;
    if (is_signed) v = (v ^ 8) - 8;
    r |= ((uint32_t)v & 0xffu) << (8 * k);
  }
  return r;
}

__global__ __launch_bounds__(DLMCQ_BLOCK) void pack4_kernel(const int8_t* __restrict__ codes,
                                                           uint8_t* __restrict__ packed, int64_t n, int vec) {
  const int64_t n16 = vec ? (n >> 4) : 0;
  for (int64_t i = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x; i < n16; i += (int64_t)gridDim.x * DLMCQ_BLOCK) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(codes) + i);
    u32x2 o;
    o.x = pack8(v.x, v.y);
    o.y = pack8(v.z, v.w);
    __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(packed) + i);
  }
  // remainder (and everything, when unaligned): one output byte per thread
  const int64_t b0 = n16 << 3, nb = (n + 1) >> 1;
  for (int64_t b = b0 + (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x; b < nb; b += (int64_t)gridDim.x * DLMCQ_BLOCK) {
    const uint32_t lo4 = (uint32_t)codes[2 * b] & 0xfu;
    const uint32_t hi4 = (2 * b + 1 < n) ? ((uint32_t)codes[2 * b + 1] & 0xfu) : 0u;
    packed[b] = (uint8_t)(lo4 | (hi4 << 4));
  }
}

__global__ __launch_bounds__(DLMCQ_BLOCK) void unpack4_kernel(const uint8_t* __restrict__ packed,
                                                             int8_t* __restrict__ codes, int64_t n, int is_signed,
                                                             int vec) {
  const int64_t n16 = vec ? (n >> 4) : 0;
  for (int64_t i = (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x; i < n16; i += (int64_t)gridDim.x * DLMCQ_BLOCK) {
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(packed) + i);
    u32x4 o;
    o.x = unpack4(v.x & 0xffffu, is_signed);
    o.y = unpack4(v.x >> 16, is_signed);
    o.z = unpack4(v.y & 0xffffu, is_signed);
    o.w = unpack4(v.y >> 16, is_signed);
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(codes) + i);
  }
  const int64_t b0 = n16 << 3, nb = (n + 1) >> 1;
  for (int64_t b = b0 + (int64_t)blockIdx.x * DLMCQ_BLOCK + threadIdx.x; b < nb; b += (int64_t)gridDim.x * DLMCQ_BLOCK) {
    const int v = packed[b];
    int l = v & 0xf, h = v >> 4;
    if (is_signed) {
      l = (l ^ 8) - 8;
      h = (h ^ 8) - 8;
    }
    codes[2 * b] = (int8_t)l;
    if (2 * b + 1 < n) codes[2 * b + 1] = (int8_t)h;
  }
}

static int pack_grid(int64_t n) {
  int64_t b = ((n >> 4) + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
  if (b < 1) b = 1;
  if (b > DLMCQ_CUS * 16) b = DLMCQ_CUS * 16;
  return (int)b;
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_pack_int4(const int8_t* codes, uint8_t* packed, int64_t n, dlmcq_stream_t stream) {
  if (n < 0) return DLMCQ_EINVAL;
  if (n == 0) return DLMCQ_OK;
  if (!codes || !packed) return DLMCQ_EINVAL;
  const int vec = aligned16(codes) && ((((uintptr_t)packed) & 7u) == 0);
  hipLaunchKernelGGL(pack4_kernel, dim3(pack_grid(n)), dim3(DLMCQ_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     codes, packed, n, vec);
  return launch_status();
}

extern "C" int dlmcq_unpack_int4(const uint8_t* packed, int8_t* codes, int64_t n, int32_t is_signed,
                                 dlmcq_stream_t stream) {
  if (n < 0) return DLMCQ_EINVAL;
  if (n == 0) return DLMCQ_OK;
  if (!codes || !packed) return DLMCQ_EINVAL;
  const int vec = aligned16(codes) && ((((uintptr_t)packed) & 7u) == 0);
  hipLaunchKernelGGL(unpack4_kernel, dim3(pack_grid(n)), dim3(DLMCQ_BLOCK), 0, reinterpret_cast<hipStream_t>(stream),
                     packed, codes, n, is_signed, vec);
  return launch_status();
}
