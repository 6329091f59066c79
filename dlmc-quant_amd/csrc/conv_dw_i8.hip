// Depthwise convolution on activation codes (groups = channels; modules/conv.py:13-19 with `groups`), the other half of a
// MobileOne / MobileNet unit.  One multiply-add per output element and tap: nothing for the matrix cores to do, and the
// layer is bound by HBM (1 byte read, 1 byte written per element), so this is plain vector arithmetic:
//   * a thread owns 16 (3 x 3 kernels) or 4 consecutive channels of one output pixel; neighbouring threads own the
//     neighbouring channel groups, so every tap is one coalesced row of C bytes and the 9 overlapping taps of neighbouring
//     pixels are served by the vector cache;
//   * the sums  S1 = SUM (q - zp) * qw  and  S0 = SUM (q - zp)  over the valid taps are small integers: they are kept in fp32
//     EXACTLY (|S1| <= 9 * 255 * 255), so the arithmetic is the integer arithmetic of conv_i8.hip;
//   * out = s_in * (s_w[c] * S1 + o_w[c] * S0) + bias[c]; asymmetric weights (w' = qw * s_w + o_w, ops.py:129-136) cost one
//     more multiply-add per element; ReLU and the consumer's quantiser (conv_epilogue.h) follow in registers.
#include "conv_i8_common.h"

namespace dlmcq {

struct DwGeom {
  int N, H, W, C4, R, S, stride, pad, P, Q;    // C4 = C / 4
  FastDiv cdiv, qdiv, pdiv;
};

__device__ __forceinline__ f32x4 bytes_u(uint32_t a) {
  return f32x4{(float)(a & 0xff), (float)((a >> 8) & 0xff), (float)((a >> 16) & 0xff), (float)(a >> 24)};   // v_cvt_f32_ubyte0..3
}
__device__ __forceinline__ f32x4 bytes_s(uint32_t a) {
  return f32x4{(float)(int8_t)a, (float)(int8_t)(a >> 8), (float)(int8_t)(a >> 16), (float)(int8_t)(a >> 24)};
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the first v_dot4_i32_i8 of a chain in its three-address form: the compiler renders __builtin_amdgcn_sdot4 as the accumulating
// v_dot4c_i32_i8, which needs a v_mov of the start value whenever that value is shared between chains (it always is here)
__device__ __forceinline__ int dot4_from(uint32_t a, uint32_t b, int c) {
  int d;
  asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
constexpr int DW_BIG = 0x7fff0000;      // a byte offset beyond every tensor the buffer-addressed tap loads accept

// 3 x 3, 16 channels per thread (C % 16 == 0): every tap is ONE 16-byte load per lane, all 9 taps requested before the first
// is used; a tap outside the image reads a clamped address and is replaced by the zero point.  The first version converted
// every byte of every tap to fp32 and multiplied there: ~80 vector instructions per output element, 0.73 TB/s - bound by its
// own arithmetic, not by HBM.  This one stays in integers: the 4 x 4 bytes (4 taps x 4 channels) of four tap dwords are
// transposed with v_perm_b32 (8 per block) so that a dword holds FOUR TAPS OF ONE CHANNEL, and v_dot4_i32_i8 multiplies it with
// the channel's four weights (packed once per workgroup into an LDS table, with the channel's constants); unsigned codes are
// re-centred by xor 0x80 with the constant (128 - zp) * SUM w added back.  Same exact integer sums S1, S0 as before, same fp32
// chain after them: bit-identical results at a third of the instructions.
struct DwTab {       // per channel, in LDS: two 16-byte records, each stored [position in the thread's 16 channels][channel group]
  uint32_t w[3];     // taps 0-3, 4-7, 8 (weights as signed bytes)        so that neighbouring lanes (neighbouring channel groups)
  int dzw;           // (shift - zp) * SUM w                               read neighbouring records: no bank conflicts
  float m, mo, b;    // s_in * s_w, s_in * o_w, bias
  uint32_t pad;
};

// FAST (round 4): 1 / 2 = codes-only layer with the plain quantiser, asymmetric / symmetric weights - what conv_dw3p2_i8_kernel<FAST> does for
// stride 1 (the fp32 chain on channel pairs, all four quantiser quads behind one branch, border taps as out-of-range buffer loads
// when the zero point is 0, table runs per lane), here for any stride (MobileOne's four stride-2 layers).  0 = everything else.
template <int FAST>
__global__ __launch_bounds__(DLMCQ_BLOCK) void conv_dw3_i8_kernel(const u32x4* __restrict__ x, const int8_t* __restrict__ w,
                                                                 float* __restrict__ out, const float* __restrict__ bias,
                                                                 const float* __restrict__ s_in, const float* __restrict__ zp_in,
                                                                 const float* __restrict__ s_w, const float* __restrict__ o_w,
                                                                 DwGeom g, int x_signed, ConvEpi ep) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dw_lds[];
  const int C = g.C4 * 4, C16 = g.C4 >> 2;
  constexpr int TS = FAST ? 17 : 0, PS = 9;                  // FAST: one run of records per channel group (conv_dw3p2_i8_kernel)
  u32x4* tabw = reinterpret_cast<u32x4*>(dw_lds);            // [16][C16]: {w0, w1, w2, dzw}                       FAST: [C16][17]
  f32x4* tabp = reinterpret_cast<f32x4*>(dw_lds) + (FAST ? C16 * 17 : C);   // [16][C16]: {m, mo, b, -}          FAST: [C16][9] {m, m', mo, mo'} of a channel pair,
  f32x2* tabb = reinterpret_cast<f32x2*>(tabp + C16 * PS);   //                                                          then [C16][9] {b, b'}
  const float sin = s_in[0], zp = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)zp;
  const int dz = (x_signed ? 0 : 128) - zpi;
  const bool asym = FAST ? FAST == 1 : o_w != nullptr;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    uint32_t pk[3] = {0u, 0u, 0u};
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int wv = w[k * C + c];
      sum += wv;
      pk[k >> 2] |= (uint32_t)(wv & 0xff) << (8 * (k & 3));
    }
    if constexpr (FAST != 0) {
      tabw[(c >> 4) * TS + (c & 15)] = u32x4{pk[0], pk[1], pk[2], (uint32_t)(dz * sum)};
      const int ps = (c >> 4) * PS + ((c & 15) >> 1), e = c & 1;
      reinterpret_cast<float*>(tabp + ps)[e] = sin * s_w[c];
      reinterpret_cast<float*>(tabp + ps)[2 + e] = asym ? sin * o_w[c] : 0.0f;
      reinterpret_cast<float*>(tabb + ps)[e] = bias ? bias[c] : 0.0f;
    } else {
      const int slot = (c & 15) * C16 + (c >> 4);
      tabw[slot] = u32x4{pk[0], pk[1], pk[2], (uint32_t)(dz * sum)};
      tabp[slot] = f32x4{sin * s_w[c], asym ? sin * o_w[c] : 0.0f, bias ? bias[c] : 0.0f, 0.0f};
    }
  }
  __syncthreads();
  const int64_t total = (int64_t)g.N * g.P * g.Q * C16;
  const int64_t xbytes = (int64_t)g.N * g.H * g.W * C;
  const bool bufpath = FAST != 0 && zpi == 0 && xbytes < (int64_t)DW_BIG;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(x), 0, bufpath ? (int)xbytes : 0, 0x00020000);
  const int dz9 = 9 * dz;
  const uint32_t zpw = (uint32_t)(zpi & 0xff) * 0x01010101u;
  const uint32_t xw = x_signed ? 0u : 0x80808080u;
  // codes-only layers: the ReLU is folded into the quantiser's clamp (code(relu(v)) = max(code(v), code(0)): conv_epilogue.h)
  const bool fold = ep.relu && !out && ep.codes;
  const EpiQuant eq(ep, fold);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t pix = fdiv((uint32_t)i, g.cdiv);                 // (cdiv divides by C / 16 here)
    const int c16 = (int)((uint32_t)i - pix * (uint32_t)C16);
    const uint32_t t = fdiv(pix, g.qdiv);
    const int q = (int)(pix - t * (uint32_t)g.Q);
    const uint32_t n = fdiv(t, g.pdiv);
    const int p = (int)(t - n * (uint32_t)g.P);
    const int h0 = p * g.stride - g.pad, w0 = q * g.stride - g.pad;
    u32x4 a[9];
    const u32x4* img = x + (int64_t)n * g.H * g.W * C16 + c16;
    const bool inside = h0 >= 0 && w0 >= 0 && h0 + 2 < g.H && w0 + 2 < g.W;
    if (bufpath) {                                        // (conv_dw3p2_i8_kernel: a tap outside the image reads beyond the tensor's end = zeros = the zero point)
      const int base0 = (int)(((uint32_t)n * (uint32_t)g.H + (uint32_t)h0) * (uint32_t)g.W + (uint32_t)w0) * C + c16 * 16;
      bool cok[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) cok[s] = (uint32_t)(w0 + s) < (uint32_t)g.W;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool rok = (uint32_t)(h0 + r) < (uint32_t)g.H;
#pragma unroll
        for (int s = 0; s < 3; ++s)
          a[r * 3 + s] = __builtin_amdgcn_raw_buffer_load_b128(rx, (rok && cok[s]) ? base0 + (r * g.W + s) * C : DW_BIG, 0, 0) ^ xw;
      }
    } else if (__builtin_amdgcn_ballot_w64(!inside) == 0) {      // the whole wave is away from the image border (almost always): no checks
      const u32x4* p0 = img + ((int64_t)h0 * g.W + w0) * C16;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) a[r * 3 + s] = p0[(r * g.W + s) * C16] ^ xw;
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int h = h0 + r, ww = w0 + s;
          const bool ok = h >= 0 && h < g.H && ww >= 0 && ww < g.W;
          const int hc = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h), wc = ww < 0 ? 0 : (ww >= g.W ? g.W - 1 : ww);
          const u32x4 v = img[((int64_t)hc * g.W + wc) * C16];
          a[r * 3 + s] = (ok ? v : u32x4{zpw, zpw, zpw, zpw}) ^ xw;   // q' = q - shift as a signed byte
        }
    }
    const int c = c16 * 16;
    uint32_t codes[4];
    f32x4 vq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {                                   // 4 channels at a time
      // bytes of a[k][d]: channels c + 4d .. + 3 of tap k.  Transposed: T[j] = taps 0-3, U[j] = taps 4-7, V[j] = tap 8 of channel j
      uint32_t T[4], U[4], V[4];
      {
        const uint32_t l01 = __builtin_amdgcn_perm(a[1][d], a[0][d], 0x05010400u), h01 = __builtin_amdgcn_perm(a[1][d], a[0][d], 0x07030602u);
        const uint32_t l23 = __builtin_amdgcn_perm(a[3][d], a[2][d], 0x05010400u), h23 = __builtin_amdgcn_perm(a[3][d], a[2][d], 0x07030602u);
        T[0] = __builtin_amdgcn_perm(l23, l01, 0x05040100u);
        T[1] = __builtin_amdgcn_perm(l23, l01, 0x07060302u);
        T[2] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
        T[3] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);
      }
      {
        const uint32_t l01 = __builtin_amdgcn_perm(a[5][d], a[4][d], 0x05010400u), h01 = __builtin_amdgcn_perm(a[5][d], a[4][d], 0x07030602u);
        const uint32_t l23 = __builtin_amdgcn_perm(a[7][d], a[6][d], 0x05010400u), h23 = __builtin_amdgcn_perm(a[7][d], a[6][d], 0x07030602u);
        U[0] = __builtin_amdgcn_perm(l23, l01, 0x05040100u);
        U[1] = __builtin_amdgcn_perm(l23, l01, 0x07060302u);
        U[2] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
        U[3] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) V[j] = (a[8][d] >> (8 * j)) & 0xffu;
      f32x4 v;
      if constexpr (FAST != 0) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {                               // channels c + 4d + 2jp, + 1
          int s1[2], s0[2] = {0, 0};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int j = 2 * jp + e;
            const u32x4 tw = tabw[c16 * TS + d * 4 + j];
            s1[e] = dot4_from(T[j], tw.x, (int)tw.w);
            s1[e] = __builtin_amdgcn_sdot4((int)U[j], (int)tw.y, s1[e], false);
            s1[e] = __builtin_amdgcn_sdot4((int)V[j], (int)tw.z, s1[e], false);
            if constexpr (FAST == 1) {
              s0[e] = dot4_from(T[j], 0x01010101u, dz9);
              s0[e] = __builtin_amdgcn_sdot4((int)U[j], 0x01010101, s0[e], false);
              s0[e] = __builtin_amdgcn_sdot4((int)V[j], 0x01010101, s0[e], false);
            }
          }
          const f32x4 mm = tabp[c16 * PS + d * 2 + jp];                // {m, m', mo, mo'}
          const f32x2 bb = tabb[c16 * PS + d * 2 + jp];
          f32x2 r = f32x2{(float)s1[0], (float)s1[1]} * f32x2{mm.x, mm.y};
          if constexpr (FAST == 1) r = r + f32x2{(float)s0[0], (float)s0[1]} * f32x2{mm.z, mm.w};
          r = r + bb;                                                   // (no bias: + 0 - the same code)
          v[2 * jp] = r.x;
          v[2 * jp + 1] = r.y;
        }
        vq[d] = v;
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 tw = tabw[(d * 4 + j) * C16 + c16];
        const f32x4 tp = tabp[(d * 4 + j) * C16 + c16];
        int s1 = __builtin_amdgcn_sdot4((int)T[j], (int)tw.x, (int)tw.w, false);
        s1 = __builtin_amdgcn_sdot4((int)U[j], (int)tw.y, s1, false);
        s1 = __builtin_amdgcn_sdot4((int)V[j], (int)tw.z, s1, false);
        float r = (float)s1 * tp.x;                                  // S1 = SUM (q - zp) * qw, exact
        if (asym) {
          int s0 = __builtin_amdgcn_sdot4((int)T[j], 0x01010101, 9 * dz, false);
          s0 = __builtin_amdgcn_sdot4((int)U[j], 0x01010101, s0, false);
          s0 = __builtin_amdgcn_sdot4((int)V[j], 0x01010101, s0, false);
          r = r + (float)s0 * tp.y;                                  // S0 = SUM (q - zp)
        }
        if (bias) r = r + tp.z;
        v[j] = r;
      }
      if (ep.relu && !fold) v = f32x4{relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)};
      const int64_t at = (int64_t)pix * g.C4 * 4 + c + d * 4;
      if (out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + at));
      if (ep.codes) codes[d] = eq.code4(v);
    }
    if constexpr (FAST != 0) eq.code4n_plain(vq, codes);
    if (ep.codes) __builtin_nontemporal_store(u32x4{codes[0], codes[1], codes[2], codes[3]},
                                              reinterpret_cast<u32x4*>(ep.codes + (int64_t)pix * g.C4 * 4 + c));
  }
}

// 3 x 3, stride 1, padding 1 (every depthwise layer of a MobileOne stage but its first): TWO horizontally adjacent output pixels
// per thread.  The pair shares 6 of its 9 + 9 taps (12 loads instead of 18), and the byte transposition is done per FILTER ROW -
// a dword holds columns q-1 .. q+2 of one channel - so one transposition serves both pixels: pixel A multiplies it with
// [w0 w1 w2 0], pixel B with [0 w0 w1 w2] (both packed once per workgroup).  Per output element: 3 instead of 5 byte permutes, the
// channel's LDS records read once per pair.  Same integer sums, same fp32 chain: bit-identical to conv_dw3_i8_kernel.
// FAST (round 4; the kernel is bound by its ~31 vector instructions per element, one per 4 clocks and SIMD): 1 / 2 = codes-only layer
// with the plain unsigned-byte quantiser (epi_plain), asymmetric / symmetric weights known at compile time - the fp32 chain on PAIRS
// of channels (v_pk_mul_f32 / v_pk_add_f32 on constants stored as pairs: the same roundings, half the instructions), the
// quantiser of both pixels behind one branch with its clamp left to v_cvt_pk_u8_f32 (EpiQuant::code4n_plain), no flag tests per
// element.  0 = everything else, as before.  Bit-identical where both apply.
template <int FAST>
__global__ __launch_bounds__(DLMCQ_BLOCK) void conv_dw3p2_i8_kernel(const u32x4* __restrict__ x, const int8_t* __restrict__ w,
                                                                   float* __restrict__ out, const float* __restrict__ bias,
                                                                   const float* __restrict__ s_in, const float* __restrict__ zp_in,
                                                                   const float* __restrict__ s_w, const float* __restrict__ o_w,
                                                                   DwGeom g, int x_signed, ConvEpi ep) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dw_lds[];
  const int C = g.C4 * 4, C16 = g.C4 >> 2;
  // Tables, one run of records per channel group (= per lane): a thread reads all its records at constant offsets from ONE base
  // address (the round-3 layout [record][group] cost an address computation per read: ~2 of the kernel's ~26 vector instructions
  // per element); the runs are padded to an odd number of 16-byte records so that neighbouring lanes start 4 banks apart.
  constexpr int TS = 17, PS = 9;
  u32x4* tabA = reinterpret_cast<u32x4*>(dw_lds);            // [C16][17]: {row 0, row 1, row 2 of pixel A's weights, dzw}
  u32x4* tabB = tabA + C16 * TS;                             // [C16][17]: {rows of pixel B's weights (one byte up), -}
  f32x4* tabp = reinterpret_cast<f32x4*>(tabB + C16 * TS);   // [C16][17]: {m, mo, b, -};  FAST: [C16][9] {m, m', mo, mo'} of a channel PAIR,
  f32x2* tabb = reinterpret_cast<f32x2*>(tabp + C16 * PS);   //                            then [C16][9] {b, b'} (zeros without a bias)
  const float sin = s_in[0], zp = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)zp;
  const int dz = (x_signed ? 0 : 128) - zpi;
  const bool asym = FAST ? FAST == 1 : o_w != nullptr;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    uint32_t ra[3];
    int sum = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ra[r] = 0u;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const int wv = w[(r * 3 + s_) * C + c];
        sum += wv;
        ra[r] |= (uint32_t)(wv & 0xff) << (8 * s_);
      }
    }
    const int slot = (c >> 4) * TS + (c & 15);
    tabA[slot] = u32x4{ra[0], ra[1], ra[2], (uint32_t)(dz * sum)};
    tabB[slot] = u32x4{ra[0] << 8, ra[1] << 8, ra[2] << 8, 0u};
    if constexpr (FAST) {
      const int ps = (c >> 4) * PS + ((c & 15) >> 1), e = c & 1;
      reinterpret_cast<float*>(tabp + ps)[e] = sin * s_w[c];
      reinterpret_cast<float*>(tabp + ps)[2 + e] = asym ? sin * o_w[c] : 0.0f;
      reinterpret_cast<float*>(tabb + ps)[e] = bias ? bias[c] : 0.0f;
    } else {
      tabp[slot] = f32x4{sin * s_w[c], asym ? sin * o_w[c] : 0.0f, bias ? bias[c] : 0.0f, 0.0f};
    }
  }
  __syncthreads();
  const int64_t xbytes = (int64_t)g.N * g.H * g.W * C;
  const bool bufpath = zpi == 0 && xbytes < (int64_t)DW_BIG;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(x), 0, bufpath ? (int)xbytes : 0, 0x00020000);
  const int dz9 = 9 * dz;
  const int QP = (g.Q + 1) >> 1;                             // pixel pairs per output row
  const int64_t total = (int64_t)g.N * g.P * QP * C16;
  const uint32_t zpw = (uint32_t)(zpi & 0xff) * 0x01010101u;
  const uint32_t xw = x_signed ? 0u : 0x80808080u;
  const bool fold = ep.relu && !out && ep.codes;
  const EpiQuant eq(ep, fold);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t pr = fdiv((uint32_t)i, g.cdiv);                  // (cdiv divides by C / 16)
    const int c16 = (int)((uint32_t)i - pr * (uint32_t)C16);
    const uint32_t t = fdiv(pr, g.qdiv);                            // (qdiv divides by QP here)
    const int qp = (int)(pr - t * (uint32_t)QP);
    const uint32_t n = fdiv(t, g.pdiv);
    const int p = (int)(t - n * (uint32_t)g.P);
    const int q = 2 * qp;
    const int h0 = p - 1, w0 = q - 1;                               // rows h0 .. h0+2, columns w0 .. w0+3
    const bool hasB = q + 1 < g.Q;
    u32x4 a[3][4];
    const u32x4* img = x + (int64_t)n * g.H * g.W * C16 + c16;
    const bool inside = h0 >= 0 && w0 >= 0 && h0 + 2 < g.H && w0 + 3 < g.W;
    if (bufpath) {
      // zero point 0 (every post-ReLU tensor): a tap outside the image is a buffer load beyond the tensor's end, which returns the zero
      // bytes the tap stands for - one compare per row and column, one select per tap, for every wave alike.  (The clamped reads of
      // the general path below cost a wave at the border ~250 vector instructions, and a third of the waves of a 28 x 28 image are.)
      const int base0 = (int)(((uint32_t)n * (uint32_t)g.H + (uint32_t)h0) * (uint32_t)g.W + (uint32_t)w0) * C + c16 * 16;   // (wraps for h0 = -1 / w0 = -1: used only where valid)
      bool cok[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) cok[s_] = (uint32_t)(w0 + s_) < (uint32_t)g.W;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool rok = (uint32_t)(h0 + r) < (uint32_t)g.H;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const int vo = (rok && cok[s_]) ? base0 + (r * g.W + s_) * C : DW_BIG;
          a[r][s_] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0) ^ xw;
        }
      }
    } else if (__builtin_amdgcn_ballot_w64(!inside) == 0) {
      const u32x4* p0 = img + ((int64_t)h0 * g.W + w0) * C16;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) a[r][s_] = p0[(r * g.W + s_) * C16] ^ xw;
    } else {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const int h = h0 + r, ww = w0 + s_;
          const bool ok = h >= 0 && h < g.H && ww >= 0 && ww < g.W;
          const int hc = h < 0 ? 0 : (h >= g.H ? g.H - 1 : h), wc = ww < 0 ? 0 : (ww >= g.W ? g.W - 1 : ww);
          const u32x4 v = img[((int64_t)hc * g.W + wc) * C16];
          a[r][s_] = (ok ? v : u32x4{zpw, zpw, zpw, zpw}) ^ xw;
        }
    }
    const int c = c16 * 16;
    const int64_t pixA = ((int64_t)n * g.P + p) * g.Q + q;
    uint32_t codesA[4], codesB[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t Rw[3][4];        // Rw[r][j]: columns w0 .. w0+3 of channel c + 4d + j in row r
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const uint32_t l01 = __builtin_amdgcn_perm(a[r][1][d], a[r][0][d], 0x05010400u), h01 = __builtin_amdgcn_perm(a[r][1][d], a[r][0][d], 0x07030602u);
        const uint32_t l23 = __builtin_amdgcn_perm(a[r][3][d], a[r][2][d], 0x05010400u), h23 = __builtin_amdgcn_perm(a[r][3][d], a[r][2][d], 0x07030602u);
        Rw[r][0] = __builtin_amdgcn_perm(l23, l01, 0x05040100u);
        Rw[r][1] = __builtin_amdgcn_perm(l23, l01, 0x07060302u);
        Rw[r][2] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
        Rw[r][3] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);
      }
      f32x4 vA, vB;
      if constexpr (FAST != 0) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {                               // channels c + 4d + 2jp, + 1
          int sA[2], sB[2], zA[2] = {0, 0}, zB[2] = {0, 0};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int j = 2 * jp + e;
            const u32x4 ta = tabA[c16 * TS + d * 4 + j];
            const u32x4 tb = tabB[c16 * TS + d * 4 + j];
            sA[e] = dot4_from(Rw[0][j], ta.x, (int)ta.w);
            sA[e] = __builtin_amdgcn_sdot4((int)Rw[1][j], (int)ta.y, sA[e], false);
            sA[e] = __builtin_amdgcn_sdot4((int)Rw[2][j], (int)ta.z, sA[e], false);
            sB[e] = dot4_from(Rw[0][j], tb.x, (int)ta.w);
            sB[e] = __builtin_amdgcn_sdot4((int)Rw[1][j], (int)tb.y, sB[e], false);
            sB[e] = __builtin_amdgcn_sdot4((int)Rw[2][j], (int)tb.z, sB[e], false);
            if constexpr (FAST == 1) {
              zA[e] = dot4_from(Rw[0][j], 0x00010101u, dz9);
              zA[e] = __builtin_amdgcn_sdot4((int)Rw[1][j], 0x00010101, zA[e], false);
              zA[e] = __builtin_amdgcn_sdot4((int)Rw[2][j], 0x00010101, zA[e], false);
              zB[e] = dot4_from(Rw[0][j], 0x01010100u, dz9);
              zB[e] = __builtin_amdgcn_sdot4((int)Rw[1][j], 0x01010100, zB[e], false);
              zB[e] = __builtin_amdgcn_sdot4((int)Rw[2][j], 0x01010100, zB[e], false);
            }
          }
          const f32x4 mm = tabp[c16 * PS + d * 2 + jp];             // {m, m', mo, mo'}
          const f32x2 bb = tabb[c16 * PS + d * 2 + jp];
          const f32x2 m2 = f32x2{mm.x, mm.y}, mo2 = f32x2{mm.z, mm.w};
          f32x2 rA = f32x2{(float)sA[0], (float)sA[1]} * m2, rB = f32x2{(float)sB[0], (float)sB[1]} * m2;
          if constexpr (FAST == 1) {
            rA = rA + f32x2{(float)zA[0], (float)zA[1]} * mo2;
            rB = rB + f32x2{(float)zB[0], (float)zB[1]} * mo2;
          }
          rA = rA + bb;            // (no bias: + 0, which only turns a -0 into +0 - the same code)
          rB = rB + bb;
          vA[2 * jp] = rA.x; vA[2 * jp + 1] = rA.y;
          vB[2 * jp] = rB.x; vB[2 * jp + 1] = rB.y;
        }
        const f32x4 vv[2] = {vA, vB};
        uint32_t ww[2];
        eq.code4n_plain(vv, ww);
        codesA[d] = ww[0];
        codesB[d] = ww[1];
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 ta = tabA[c16 * TS + d * 4 + j];
        const u32x4 tb = tabB[c16 * TS + d * 4 + j];
        const f32x4 tp = tabp[c16 * TS + d * 4 + j];
        int sA = __builtin_amdgcn_sdot4((int)Rw[0][j], (int)ta.x, (int)ta.w, false);
        sA = __builtin_amdgcn_sdot4((int)Rw[1][j], (int)ta.y, sA, false);
        sA = __builtin_amdgcn_sdot4((int)Rw[2][j], (int)ta.z, sA, false);
        int sB = __builtin_amdgcn_sdot4((int)Rw[0][j], (int)tb.x, (int)ta.w, false);
        sB = __builtin_amdgcn_sdot4((int)Rw[1][j], (int)tb.y, sB, false);
        sB = __builtin_amdgcn_sdot4((int)Rw[2][j], (int)tb.z, sB, false);
        float rA = (float)sA * tp.x, rB = (float)sB * tp.x;        // S1 = SUM (q - zp) * qw, exact
        if (asym) {
          int zA = __builtin_amdgcn_sdot4((int)Rw[0][j], 0x00010101, 9 * dz, false);
          zA = __builtin_amdgcn_sdot4((int)Rw[1][j], 0x00010101, zA, false);
          zA = __builtin_amdgcn_sdot4((int)Rw[2][j], 0x00010101, zA, false);
          int zB = __builtin_amdgcn_sdot4((int)Rw[0][j], 0x01010100, 9 * dz, false);
          zB = __builtin_amdgcn_sdot4((int)Rw[1][j], 0x01010100, zB, false);
          zB = __builtin_amdgcn_sdot4((int)Rw[2][j], 0x01010100, zB, false);
          rA = rA + (float)zA * tp.y;                                // S0 = SUM (q - zp)
          rB = rB + (float)zB * tp.y;
        }
        if (bias) {
          rA = rA + tp.z;
          rB = rB + tp.z;
        }
        vA[j] = rA;
        vB[j] = rB;
      }
      if (ep.relu && !fold) {
        vA = f32x4{relu_nan(vA.x), relu_nan(vA.y), relu_nan(vA.z), relu_nan(vA.w)};
        vB = f32x4{relu_nan(vB.x), relu_nan(vB.y), relu_nan(vB.z), relu_nan(vB.w)};
      }
      const int64_t at = pixA * C + c + d * 4;
      if (out) {
        __builtin_nontemporal_store(vA, reinterpret_cast<f32x4*>(out + at));
        if (hasB) __builtin_nontemporal_store(vB, reinterpret_cast<f32x4*>(out + at + C));
      }
      if (ep.codes) {
        codesA[d] = eq.code4(vA);
        codesB[d] = eq.code4(vB);
      }
    }
    if (ep.codes) {
      __builtin_nontemporal_store(u32x4{codesA[0], codesA[1], codesA[2], codesA[3]}, reinterpret_cast<u32x4*>(ep.codes + pixA * C + c));
      if (hasB)
        __builtin_nontemporal_store(u32x4{codesB[0], codesB[1], codesB[2], codesB[3]}, reinterpret_cast<u32x4*>(ep.codes + (pixA + 1) * C + c));
    }
  }
}

// Any R, S <= 7 and C % 4 == 0: one dword of codes per thread and tap.
__global__ __launch_bounds__(DLMCQ_BLOCK) void conv_dw_i8_kernel(const uint32_t* __restrict__ x, const uint32_t* __restrict__ w,
                                                                float* __restrict__ out, const float* __restrict__ bias,
                                                                const float* __restrict__ s_in, const float* __restrict__ zp_in,
                                                                const float* __restrict__ s_w, const float* __restrict__ o_w,
                                                                DwGeom g, int x_signed, ConvEpi ep) {
  const int64_t total = (int64_t)g.N * g.P * g.Q * g.C4;
  const float sin = s_in[0], zp = zp_in ? zp_in[0] : 0.0f;
  const EpiQuant eq(ep);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t pix = fdiv((uint32_t)i, g.cdiv);
    const int c4 = (int)((uint32_t)i - pix * (uint32_t)g.C4);
    const uint32_t t = fdiv(pix, g.qdiv);
    const int q = (int)(pix - t * (uint32_t)g.Q);
    const uint32_t n = fdiv(t, g.pdiv);
    const int p = (int)(t - n * (uint32_t)g.P);
    const int h0 = p * g.stride - g.pad, w0 = q * g.stride - g.pad;
    f32x4 s1 = {0.0f, 0.0f, 0.0f, 0.0f}, s0 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int r = 0; r < g.R; ++r) {
      const int h = h0 + r;
      if (h < 0 || h >= g.H) continue;
      for (int s = 0; s < g.S; ++s) {
        const int ww = w0 + s;
        if (ww < 0 || ww >= g.W) continue;                       // padded tap: x' = 0
        const uint32_t a = x[(((int64_t)n * g.H + h) * g.W + ww) * g.C4 + c4];
        f32x4 av = x_signed ? bytes_s(a) : bytes_u(a);
        const f32x4 bv = bytes_s(w[(r * g.S + s) * g.C4 + c4]);
        av = f32x4{av.x - zp, av.y - zp, av.z - zp, av.w - zp};
        s1 = f32x4{__builtin_fmaf(av.x, bv.x, s1.x), __builtin_fmaf(av.y, bv.y, s1.y), __builtin_fmaf(av.z, bv.z, s1.z),
                   __builtin_fmaf(av.w, bv.w, s1.w)};
        s0 = f32x4{s0.x + av.x, s0.y + av.y, s0.z + av.z, s0.w + av.w};
      }
    }
    const int c = c4 * 4;
    const f32x4 sw = *reinterpret_cast<const f32x4*>(s_w + c);
    f32x4 v = f32x4{s1.x * (sin * sw.x), s1.y * (sin * sw.y), s1.z * (sin * sw.z), s1.w * (sin * sw.w)};
    if (o_w) {
      const f32x4 ow = *reinterpret_cast<const f32x4*>(o_w + c);
      v = f32x4{v.x + s0.x * (sin * ow.x), v.y + s0.y * (sin * ow.y), v.z + s0.z * (sin * ow.z), v.w + s0.w * (sin * ow.w)};
    }
    if (bias) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + c);
      v = f32x4{v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w};
    }
    if (ep.relu) v = f32x4{relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)};
    const int64_t at = (int64_t)pix * g.C4 * 4 + c;
    if (out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + at));
    if (ep.codes) __builtin_nontemporal_store(eq.code4(v), reinterpret_cast<uint32_t*>(ep.codes + at));
  }
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_conv2d_dw_i8_nhwc(const void* x, const int8_t* w, float* out, const float* bias, const float* in_scale,
                                       const float* in_zero_point, const float* w_scale, const float* w_offset, int64_t N,
                                       int64_t H, int64_t W, int64_t C, int64_t R, int64_t S, int32_t stride, int32_t pad,
                                       int32_t x_is_unsigned, int32_t relu, void* codes, const float* q_scale,
                                       const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g,
                                       dlmcq_stream_t stream) {
  if (N < 0 || H < 1 || W < 1 || C < 4 || (C & 3) || R < 1 || S < 1 || R > 7 || S > 7 || stride < 1 || pad < 0) return DLMCQ_EINVAL;
  const int64_t P = (H + 2 * pad - R) / stride + 1, Q = (W + 2 * pad - S) / stride + 1;
  if (P < 1 || Q < 1) return DLMCQ_EINVAL;
  const uint32_t ctl = (uint32_t)q_form & (DLMCQ_FORCE_TILED | DLMCQ_ROUTE_ONLY);     // (include/dlmcq.h: control bits of `q_form`)
  q_form &= ~(DLMCQ_FORCE_TILED | DLMCQ_ROUTE_ONLY);
  if (N == 0) return DLMCQ_OK;
  if (!x || !w || !(out || codes) || !in_scale || !w_scale) return DLMCQ_EINVAL;
  if (codes && (!q_scale || q_lo > q_hi || q_lo < -128 || q_hi > 255 || q_hi - q_lo > 255 || q_form < DLMCQ_FORM_EMULATE ||
                q_form > DLMCQ_FORM_SYMMETRIC))
    return DLMCQ_EINVAL;
  if (!aligned4(x) || !aligned4(w) || !aligned16(w_scale) || (w_offset && !aligned16(w_offset)) || (bias && !aligned16(bias)) ||
      (out && !aligned16(out)) || (codes && !aligned4(codes)))
    return DLMCQ_EALIGN;
  if (N * P * Q * (C / 4) >= (1ll << 31) || N * H * W * (C / 4) >= (1ll << 31)) return DLMCQ_ERANGE;
  DwGeom g;
  g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C4 = (int)(C / 4); g.R = (int)R; g.S = (int)S; g.stride = stride; g.pad = pad;
  g.P = (int)P; g.Q = (int)Q;
  g.cdiv = make_fastdiv((uint32_t)g.C4);
  g.qdiv = make_fastdiv((uint32_t)Q);
  g.pdiv = make_fastdiv((uint32_t)P);
  ConvEpi ep{};
  ep.relu = relu != 0;
  ep.codes = static_cast<uint8_t*>(codes);
  ep.q_scale = q_scale;
  ep.q_zp = q_zero_point;
  ep.q_lo = (float)q_lo;
  ep.q_hi = (float)q_hi;
  ep.q_g = q_ste_g;
  ep.q_form = q_form;
  const int64_t total = N * P * Q * (C / 4);
  const int64_t blocks = (total + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!(ctl & DLMCQ_FORCE_TILED) && conv_dwm_applies(N, H, W, C, R, S, stride, pad, ep, out, x)) {   // codes-only 3 x 3 / 1 / 1 layers: the multiply-adds on the matrix cores
    if (ctl & DLMCQ_ROUTE_ONLY) return DLMCQ_ROUTE_DWM;
    return conv_dwm_launch(static_cast<const int8_t*>(x), w, bias, in_scale, in_zero_point, w_scale, w_offset, N, H, W, C,
                           x_is_unsigned ? 0 : 1, ep, st);
  }
  if (ctl & DLMCQ_ROUTE_ONLY) return DLMCQ_ROUTE_DW;
  const bool wide = R == 3 && S == 3 && C % 16 == 0 && C <= 2048 && aligned16(x) && (!codes || aligned16(codes));
  if (wide && stride == 1 && pad == 1 && C <= 1024) {     // two output pixels per thread (LDS: 48 B per channel)
    g.cdiv = make_fastdiv((uint32_t)(C / 16));
    const int64_t qpairs = (Q + 1) / 2;
    g.qdiv = make_fastdiv((uint32_t)qpairs);
    const int64_t b16 = (N * P * qpairs * (C / 16) + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
    const int fast = (!out && epi_plain(ep)) ? (w_offset ? 1 : 2) : 0;
#define DLMCQ_DWP2(F) hipLaunchKernelGGL(conv_dw3p2_i8_kernel<F>, dim3((uint32_t)(b16 < 4096 ? b16 : 4096)), dim3(DLMCQ_BLOCK), (size_t)(C / 16) * (3 * 17 * 16), st, \
                                         static_cast<const u32x4*>(x), w, out, bias, in_scale, in_zero_point, w_scale, w_offset, g,              \
                                         x_is_unsigned ? 0 : 1, ep)
    if (fast == 1) DLMCQ_DWP2(1);
    else if (fast == 2) DLMCQ_DWP2(2);
    else DLMCQ_DWP2(0);
#undef DLMCQ_DWP2
  } else if (wide) {
    g.cdiv = make_fastdiv((uint32_t)(C / 16));
    const int64_t b16 = (N * P * Q * (C / 16) + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
    // every workgroup packs the layer's weights into its LDS table first: a grid of a few workgroups per CU, each walking many pixels
    const int fast = (!out && epi_plain(ep)) ? (w_offset ? 1 : 2) : 0;
#define DLMCQ_DW3(F) hipLaunchKernelGGL(conv_dw3_i8_kernel<F>, dim3((uint32_t)(b16 < 4096 ? b16 : 4096)), dim3(DLMCQ_BLOCK),                 \
                                        F ? (size_t)(C / 16) * (17 * 16 + 9 * 16 + 9 * 8) : (size_t)C * sizeof(DwTab), st, static_cast<const u32x4*>(x), w, \
                                        out, bias, in_scale, in_zero_point, w_scale, w_offset, g, x_is_unsigned ? 0 : 1, ep)
    if (fast == 1) DLMCQ_DW3(1);
    else if (fast == 2) DLMCQ_DW3(2);
    else DLMCQ_DW3(0);
#undef DLMCQ_DW3
  } else {
    hipLaunchKernelGGL(conv_dw_i8_kernel, dim3((uint32_t)(blocks < (1 << 20) ? blocks : (1 << 20))), dim3(DLMCQ_BLOCK), 0, st,
                       static_cast<const uint32_t*>(x), reinterpret_cast<const uint32_t*>(w), out, bias, in_scale, in_zero_point, w_scale,
                       w_offset, g, x_is_unsigned ? 0 : 1, ep);
  }
  return launch_status();
}
