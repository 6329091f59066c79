// Definitions shared by the int8 convolution translation units (conv_i8.hip, conv_i8_pp.hip).  Not part of the ABI.
#pragma once

#include <type_traits>

#include "dlmcq_internal.h"
#include "conv_epilogue.h"

namespace dlmcq {

constexpr int CV_BM = 128;
constexpr int CV_BK = 64;
constexpr int CV_LD = CV_BK + 16;  // LDS row stride in bytes

struct ConvGeom {
  int N, H, W, C, K, R, S, stride, pad, dil, P, Q;
  int64_t M;          // N*P*Q (< 2^31)
  int nblk_m, nblk_n;
  FastDiv qdiv, pdiv; // row index -> (n, p, q) without 64-bit divisions
};

// output row m -> image n and the top-left input coordinate of its receptive field
__device__ __forceinline__ void row_origin(const ConvGeom& g, uint32_t m, int& n, int& h0, int& w0) {
  const uint32_t t = fdiv(m, g.qdiv);
  const int q = (int)(m - t * (uint32_t)g.Q);
  const uint32_t nn = fdiv(t, g.pdiv);
  const int p = (int)(t - nn * (uint32_t)g.P);
  n = (int)nn;
  h0 = p * g.stride - g.pad;
  w0 = q * g.stride - g.pad;
}

struct PadTable {   // 64 bytes of every byte value: a padded tap reads its K chunks at offsets 0 / 32 of one line
  int8_t b[256 * 64];
  constexpr PadTable() : b() {
    for (int v = 0; v < 256; ++v)
      for (int j = 0; j < 64; ++j) b[v * 64 + j] = (int8_t)v;
  }
};
static __device__ const PadTable g_pad_table = PadTable();

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


struct ConvSeg2 {
  const int8_t* x;
  const int8_t* w;
  const float* bias;
  const int32_t* wsum;
  const float* s_in;
  const float* zp_in;
  const float* s_w;
  ConvGeom g;
  int shift;
};

// A 16-byte global load the compiler does not know about (no s_waitcnt is generated for it: the caller counts vmcnt)
template <int OFF>
__device__ __forceinline__ void gload16(i32x4& dst, const int8_t* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}

// Buffer-addressed 16-byte accesses the compiler does not know about (no s_waitcnt is generated for them: the caller counts vmcnt).
// A byte offset beyond the resource's size makes a load return zeros and a store vanish, so a lane outside its tile still ISSUES the
// instruction: every wave issues the same number of vector-memory operations, which counted waits rely on.
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i make_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  return v4i{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// The fp32 streams (shortcut in, block output out) carry the non-temporal hint: measured twice (LABNOTES 13, 14) against temporal
// accesses - whole batches and cache-sized sub-batches walked through a stage.  -DDLMCQ_FP32_TEMPORAL builds the A/B library without it.
#ifdef DLMCQ_FP32_TEMPORAL
#define DLMCQ_NT ""
#else
#define DLMCQ_NT " nt"
#endif
__device__ __forceinline__ void bload16(f32x4& dst, int voff, const v4i& rsrc) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" DLMCQ_NT : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
template <int OFF>
__device__ __forceinline__ void bload16i(i32x4& dst, int voff, const v4i& rsrc) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF) : "memory");
}
// (a store of more than 8 bytes reads its data registers late: gfx940+ needs TWO wait states before a VALU instruction may
//  overwrite them, and the hazard recogniser does not see inside inline asm - with one, the last quad of every 16 lanes
//  can store the NEXT value of a dword)
__device__ __forceinline__ void bstore16(const f32x4& v, int voff, const v4i& rsrc) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" DLMCQ_NT "\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void bstore16i(const i32x4& v, int voff, const v4i& rsrc) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void bstore16i_nt(const i32x4& v, int voff, const v4i& rsrc) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
constexpr int BUF_BIG = 0x7fff0000;   // a byte offset beyond every buffer these kernels accept (< 2^31 - 64 KiB)

// conv3x3_i8.hip: the halo-tile kernel for 3x3 / stride 1 or 2 / pad 1 layers that emit only their consumer's codes
bool conv3x3_halo_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                          int32_t dilation, const ConvEpi& ep, const float* out, bool dual);
int conv3x3_halo_launch(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                        const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                        int32_t stride, int shift, const ConvEpi& ep, hipStream_t st, int lab = 0, void* lab_trace = nullptr);

// conv3x3_pipe_i8.hip: the halo-tile kernel persistent and software-pipelined across tiles (stride 1, 128 / 256 / 512 input channels, plain quantiser)
bool conv3x3_pipe_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int32_t stride, const ConvEpi& ep, int cus);
int conv3x3_pipe_launch(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                        const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int shift,
                        const ConvEpi& ep, hipStream_t st, int cus);
// compute units of the current device (cached: one device per process, dlmc/_native.py)
inline int device_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}

// conv_pw_i8.hip: pointwise codes-to-codes layers with the weights resident in LDS (MobileOne's 1x1 layers)
bool conv_pw_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                     int32_t dilation, const ConvEpi& ep, const float* out, bool dual);
int conv_pw_launch(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                   const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int shift,
                   const ConvEpi& ep, hipStream_t st, int lab = 0, void* lab_trace = nullptr);

// conv_pwr_i8.hip: 1 x 1 block ends with an fp32 shortcut (+ fp32 output) + ReLU + codes, the weights resident in LDS
bool conv_pwr_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                      int32_t dilation, const ConvEpi& ep, const float* out, const ConvSeg2* seg2);
int conv_pwr_launch(const int8_t* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum, const float* in_scale,
                    const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int32_t stride,
                    int shift, const ConvEpi& ep, hipStream_t st, const ConvSeg2* seg2 = nullptr);

// conv_dwm_i8.hip: depthwise 3 x 3 / stride 1 / padding 1 codes-to-codes layers on the matrix cores (diagonal weight fragments)
bool conv_dwm_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t R, int64_t S, int32_t stride, int32_t pad, const ConvEpi& ep,
                      const float* out, const void* x);
int conv_dwm_launch(const int8_t* x, const int8_t* w, const float* bias, const float* in_scale, const float* in_zero_point,
                    const float* w_scale, const float* w_offset, int64_t N, int64_t H, int64_t W, int64_t C, int x_signed,
                    const ConvEpi& ep, hipStream_t st);

// conv_stem_pool7_i8.hip: the ResNet first layer (7x7 / 2 + ReLU + 3x3 / 2 max-pool + quantiser) with the pooling in registers
bool stem_pool7_applies(int64_t Hp, int64_t Wp, int64_t K, int64_t R, int64_t S, int32_t stride, const float* out, const void* codes);
int stem_pool7_launch(const uint8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                      const float* in_zero_point, const float* w_scale, int64_t N, int64_t Hp, int64_t Wp, int64_t S, int shift,
                      const ConvEpi& ep, hipStream_t st);

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

}  // namespace dlmcq
