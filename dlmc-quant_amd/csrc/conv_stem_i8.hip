// The network's first convolution (3 input channels: 7x7/2 in ResNet, 3x3/2 in RepVGG / MobileOne) on the matrix
// cores, and the max-pool that follows it, in the integer-code domain.
//
// The generic int8 kernel (conv_i8.hip) steps its reduction 64 input channels at a time, so a 3-channel layer
// cannot use it; the reference (and the module path of this build) run it as an fp32 convolution of the
// fake-quantised image, followed by separate bias-add, ReLU, max-pool and next-layer quantise passes over a
// 64-channel 112x112 fp32 tensor - at batch 512 that is 3.8 ms of a 13 ms forward for 3 % of the MACs.
//
// Layout trick: quantise the image into a zero-point-PADDED NHWC buffer with 4 bytes per pixel
// ([N][H+2p][W+2p][4], border = the code of x' = 0, 4th byte unused).  For an output pixel and a filter row r the
// S <= 8 taps x 4 bytes it needs are then 32 CONTIGUOUS, 4-byte-aligned bytes - exactly one K = 32 operand of
// v_mfma_i32_32x32x32_i8 (lanes 0-31: taps 0-3, lanes 32-63: taps 4-7), loaded straight from global memory with no
// bounds checks (the border is physically there) and no LDS staging.  Taps beyond S and the 4th channel meet zero
// weights.  One MFMA per filter row per 32 output channels; the weights of all rows live in registers for the whole
// (persistent) kernel.  Epilogue as in conv_i8.hip: exact int32 sum -> one rounding chain -> bias, ReLU, and the
// consumer's activation code (conv_epilogue.h).
//
// Max-pool commutes with the (monotone) quantiser: code(max(v)) = max(code(v)).  So the pool runs on the codes,
// 1 byte per element instead of 4, and never sees fp32.
#include "conv_i8_common.h"

namespace dlmcq {

// ---------------------------------------------------------------- image -> padded NHWC4 codes
struct ImgGeom {
  int N, C, H, W, pad, Hp, Wp;
  int64_t sn, sc, sh, sw;   // element strides of the fp32 input (any memory format)
  FastDiv wdiv, hdiv;
};

__global__ __launch_bounds__(DLMCQ_BLOCK) void quantize_pad_nhwc4_kernel(const float* __restrict__ x, uint32_t* __restrict__ out,
                                                                         ImgGeom g, ConvEpi q) {
  // the quantiser is EpiQuant::code4 (conv_epilogue.h): the four forms' arithmetic at ~10 operations per element instead of
  // a correctly rounded division each (this kernel was bound by them, not by HBM), bit-identical codes
  const EpiQuant eq(q);
  const uint32_t border = eq.code4(f32x4{0.0f, 0.0f, 0.0f, 0.0f});   // x' = 0 (zero padding of the fake-quantised image) in every channel
  const uint32_t keep = g.C >= 4 ? 0xffffffffu : (1u << (8 * g.C)) - 1u;   // bytes of channels that do not exist stay 0
  const int64_t total = (int64_t)g.N * g.Hp * g.Wp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t t = fdiv((uint32_t)i, g.wdiv);
    const int wp = (int)((uint32_t)i - t * (uint32_t)g.Wp);
    const uint32_t n = fdiv(t, g.hdiv);
    const int hp = (int)(t - n * (uint32_t)g.Hp);
    const int h = hp - g.pad, w = wp - g.pad;
    uint32_t px = border;
    if (h >= 0 && h < g.H && w >= 0 && w < g.W) {
      const float* p = x + (int64_t)n * g.sn + (int64_t)h * g.sh + (int64_t)w * g.sw;
      f32x4 v;
      v.x = p[0];
      v.y = g.C > 1 ? p[g.sc] : 0.0f;
      v.z = g.C > 2 ? p[2 * g.sc] : 0.0f;
      v.w = g.C > 3 ? p[3 * g.sc] : 0.0f;
      px = eq.code4(v) & keep;
    }
    __builtin_nontemporal_store(px, out + i);
  }
}

// The same, four pixels of one image row per thread (W % 4 = 0, unit pixel stride, 16-byte-aligned rows: every image this project
// quantises): one 16-byte load per channel plane and one 16-byte store instead of C + 1 four-byte accesses per pixel, so a 16-lane
// group of a load touches two whole cache lines instead of half of one, and the quantiser runs on the C real channels only
// (code4 on a channel-planar quad, bytes dealt to the four pixel words by v_perm_b32).  Same bytes as the kernel above.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte store at a 4-byte-aligned address
__global__ __launch_bounds__(DLMCQ_BLOCK) void quantize_pad_nhwc4_x4_kernel(const float* __restrict__ x, uint32_t* __restrict__ out,
                                                                            ImgGeom g, ConvEpi q, FastDiv gdiv, int groups) {
  const EpiQuant eq(q);
  const uint32_t border = eq.code4(f32x4{0.0f, 0.0f, 0.0f, 0.0f});
  const int64_t total = (int64_t)g.N * g.Hp * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t t = fdiv((uint32_t)i, gdiv);
    const int gi = (int)((uint32_t)i - t * (uint32_t)groups);
    const uint32_t n = fdiv(t, g.hdiv);
    const int hp = (int)(t - n * (uint32_t)g.Hp);
    const int h = hp - g.pad;
    uint32_t* orow = out + (int64_t)t * g.Wp;             // t = n * Hp + hp
    uint32_t px0 = border, px1 = border, px2 = border, px3 = border;
    if (h >= 0 && h < g.H) {
      const float* p = x + (int64_t)n * g.sn + (int64_t)h * g.sh + 4 * gi;
      uint32_t w0 = eq.code4(__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))), w1 = 0u, w2 = 0u, w3 = 0u;
      if (g.C > 1) w1 = eq.code4(__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + g.sc)));
      if (g.C > 2) w2 = eq.code4(__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 2 * g.sc)));
      if (g.C > 3) w3 = eq.code4(__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 3 * g.sc)));
      // pixel j = bytes j of (w0, w1, w2, w3)
      const uint32_t a0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), a1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);   // [w0.b0 w1.b0 w0.b1 w1.b1], [w0.b2 w1.b2 w0.b3 w1.b3]
      const uint32_t b0 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), b1 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
      px0 = __builtin_amdgcn_perm(b0, a0, 0x05040100u);
      px1 = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
      px2 = __builtin_amdgcn_perm(b1, a1, 0x05040100u);
      px3 = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
    }
    __builtin_nontemporal_store(u32x4_a4{px0, px1, px2, px3}, reinterpret_cast<u32x4_a4*>(orow + g.pad + 4 * gi));
    if (gi == 0)
      for (int k = 0; k < g.pad; ++k) orow[k] = border;
    if (gi == groups - 1)
      for (int k = 0; k < g.pad; ++k) orow[g.pad + g.W + k] = border;
  }
}

// ------------------------------------------------------------------ weights -> [K][R][8 taps][4]
__global__ __launch_bounds__(DLMCQ_BLOCK) void quantize_weight_stem_kernel(const float* __restrict__ w, int8_t* __restrict__ wq,
                                                                           int32_t* __restrict__ wsum,
                                                                           const float* __restrict__ scale, int C, int R, int S,
                                                                           float lo, float hi) {
  // one block per output channel; SYMMETRIC form (FSPTQuant/base.py:149-152), as quantize_weight_krsc_kernel
  const int k = blockIdx.x;
  const float s = scale[k];
  __shared__ int part[DLMCQ_BLOCK / DLMCQ_WAVE];
  int sum = 0;
  for (int i = threadIdx.x; i < R * 32; i += blockDim.x) {
    const int r = i >> 5, tap = (i >> 2) & 7, c = i & 3;
    int q = 0;
    if (tap < S && c < C) {
      const float v = w[(((int64_t)k * C + c) * R + r) * S + tap];
      q = code_of(clamp_nan(ste_round(v / s), lo, hi));
    }
    wq[(int64_t)k * R * 32 + i] = (int8_t)q;
    sum += q;
  }
  for (int o = DLMCQ_WAVE / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, DLMCQ_WAVE);
  if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) part[threadIdx.x / DLMCQ_WAVE] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < DLMCQ_BLOCK / DLMCQ_WAVE; ++i) t += part[i];
    wsum[k] = t;
  }
}

// ----------------------------------------------------------------------------- the convolution
struct StemGeom {
  int N, Hp, Wp, K, R, stride, P, Q;
  int S, C;      // filter width and real input channels (ASYM only: which operand bytes count towards SUM x')
  int64_t M;     // N*P*Q
  int ntiles;    // ceil(M / 32)
  FastDiv qdiv, pdiv;
};

constexpr int ST_MAXR = 7;
constexpr int ST_EP_LD = 68;   // floats per staged row (64 + 4 pad)

// One wave = one 32-pixel x 64-channel output tile at a time (grid-stride over tiles); 4 independent waves per
// workgroup.  blockIdx.y selects the 64-channel slab.
// ASYM: asymmetric per-channel weights w' = qw * s_w[k] + o_w[k] (ops.py:129-136; ep.w_off): the extra term o_w[k] * SUM x' of the
// receptive field, SUM x' = s_in * (SUM q' + (shift - zp) * R * S * C), the code sum taken from the operand fragments with
// v_dot4_i32_i8 against a 0/1 mask of the bytes that are real taps and real channels (as in conv_i8.hip's ASYM instantiation).
// SWAP (codes-only output, K a multiple of 64): operands exchanged as in conv_i8.hip's SWAP kernels - a lane's accumulators are 16
// consecutive channels of ONE pixel (the weight fragments are loaded in the matching row order), so the epilogue needs no
// transposition through LDS, folds the ReLU into the quantiser's clamp and reads its per-channel constants, pre-multiplied once
// per workgroup, as broadcast ds_read_b128; the per-pixel code sum of ASYM is a per-lane scalar.
template <int R, bool ASYM = false, bool SWAP = false>
__global__ __launch_bounds__(256) void conv_stem_i8_kernel(const uint8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                           float* __restrict__ out, const float* __restrict__ bias,
                                                           const int32_t* __restrict__ wsum, const float* __restrict__ s_in,
                                                           const float* __restrict__ zp_in, const float* __restrict__ s_w,
                                                           StemGeom g, int shift, ConvEpi ep) {
  __shared__ __attribute__((aligned(16))) float stage[4 * 32 * ST_EP_LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.y * 64;
  const int hsel = lane >> 5;
  const uint32_t xorw = shift ? 0x80808080u : 0u;
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin = s_in[0];
  const EpiQuant eq(ep, SWAP && ep.relu);
  const bool plainq = epi_plain(ep);         // unsigned bytes, no zero point: EpiQuant::code4n_plain (one uniform branch per 16 channels)

  // weights of this slab: fragment (r, j) = 16 bytes of channel n0 + j*32 + (lane & 31), taps hsel*4 .. +3
  // (SWAP: row d of a block is channel 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3): accumulator register i = channel 16 hsel + i)
  i32x4 bf[R][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int d31 = lane & 31;
    const int k = n0 + j * 32 + (SWAP ? 16 * ((d31 >> 2) & 1) + 4 * (d31 >> 3) + (d31 & 3) : d31);
#pragma unroll
    for (int r = 0; r < R; ++r)
      bf[r][j] = k < g.K ? *reinterpret_cast<const i32x4*>(w + ((int64_t)k * R + r) * 32 + hsel * 16) : i32x4{0, 0, 0, 0};
  }
  float mult[2], bv[2];
  int corr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = n0 + j * 32 + (lane & 31);
    const bool ok = k < g.K;
    mult[j] = ok ? sin * s_w[k] : 0.0f;
    corr[j] = ok ? (shift - zpi) * wsum[k] : 0;
    bv[j] = (ok && bias) ? bias[k] : 0.0f;
  }
  float* stg = stage + wave * (32 * ST_EP_LD);
  const int er = lane >> 4, ec = (lane & 15) * 4;
  const int rowbytes = g.Wp * 4;
  __shared__ int s0tab[ASYM ? 4 * 32 : 1];
  float woff[2] = {0.0f, 0.0f};
  int maskw[4] = {0, 0, 0, 0};
  if (ASYM) {
    const int cm = g.C >= 4 ? 0x01010101 : (g.C == 3 ? 0x00010101 : (g.C == 2 ? 0x00000101 : 0x00000001));
#pragma unroll
    for (int d = 0; d < 4; ++d) maskw[d] = (hsel * 4 + d < g.S) ? cm : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = n0 + j * 32 + (lane & 31);
      woff[j] = k < g.K ? sin * ep.w_off[k] : 0.0f;
    }
  }

  // SWAP: the epilogue's constants by channel of this slab, ready-made: s_in * s_w | (shift - zp) * SUM qw | bias | s_in * o_w
  __shared__ __attribute__((aligned(16))) float ctab[SWAP ? 4 * 64 : 4];
  if constexpr (SWAP) {
    if (threadIdx.x < 64) {
      const int k = n0 + threadIdx.x;
      const bool ok = k < g.K;
      ctab[threadIdx.x] = ok ? sin * s_w[k] : 0.0f;
      reinterpret_cast<int*>(ctab)[64 + threadIdx.x] = ok ? (shift - zpi) * wsum[k] : 0;
      ctab[128 + threadIdx.x] = (ok && bias) ? bias[k] : 0.0f;
      ctab[192 + threadIdx.x] = (ASYM && ok) ? sin * ep.w_off[k] : 0.0f;
    }
    __syncthreads();
  }

  // the operand fragments of a tile are requested one tile ahead (before the previous tile's epilogue): a wave's loop is
  // load -> R MFMAs -> a long quantising epilogue, and without the prefetch every iteration exposed the load's latency
  auto fetch = [&](int tile, i32x4 (&f)[R]) {
    int64_t m = (int64_t)tile * 32 + (lane & 31);
    if (m >= g.M) m = g.M - 1;                       // ragged last tile / beyond the last tile: a valid pixel, never stored
    const uint32_t t = fdiv((uint32_t)m, g.qdiv);
    const int q = (int)((uint32_t)m - t * (uint32_t)g.Q);
    const uint32_t n = fdiv(t, g.pdiv);
    const int p = (int)(t - n * (uint32_t)g.P);
    const uint8_t* src = x + (((int64_t)n * g.Hp + (int64_t)p * g.stride) * g.Wp + (int64_t)q * g.stride) * 4 + hsel * 16;
#pragma unroll
    for (int r = 0; r < R; ++r) f[r] = *reinterpret_cast<const i32x4*>(src + (int64_t)r * rowbytes);
  };
  i32x4 afn[R];
  if ((int)(blockIdx.x * 4 + wave) < g.ntiles) fetch(blockIdx.x * 4 + wave, afn);
  for (int tile = blockIdx.x * 4 + wave; tile < g.ntiles; tile += gridDim.x * 4) {
    const int64_t m0 = (int64_t)tile * 32;
    i32x4 af[R];
#pragma unroll
    for (int r = 0; r < R; ++r) af[r] = afn[r];
    fetch(tile + (int)gridDim.x * 4, afn);
    i32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (SWAP) {      // the accumulators start from the zero-point term of their channels (four LDS reads instead of 16 additions)
        const i32x4* cop = reinterpret_cast<const i32x4*>(ctab + 64 + j * 32 + hsel * 16);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const i32x4 c4 = cop[q4];
          acc[j][4 * q4] = c4.x;
          acc[j][4 * q4 + 1] = c4.y;
          acc[j][4 * q4 + 2] = c4.z;
          acc[j][4 * q4 + 3] = c4.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0;
      }
    }
    int s0 = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const i32x4 a = i32x4{(int)(af[r].x ^ xorw), (int)(af[r].y ^ xorw), (int)(af[r].z ^ xorw), (int)(af[r].w ^ xorw)};
      if (ASYM) {
        s0 = __builtin_amdgcn_sdot4(a.x, maskw[0], s0, false);
        s0 = __builtin_amdgcn_sdot4(a.y, maskw[1], s0, false);
        s0 = __builtin_amdgcn_sdot4(a.z, maskw[2], s0, false);
        s0 = __builtin_amdgcn_sdot4(a.w, maskw[3], s0, false);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = SWAP ? __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[r][j], a, acc[j], 0, 0, 0)
                      : __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf[r][j], acc[j], 0, 0, 0);
    }
    if constexpr (SWAP) {
      // lane (p = lane & 31, hsel): channels n0 + 32 j + 16 hsel + 0..15 of pixel m0 + p
      f32x2 s0f2 = f32x2{0.0f, 0.0f};
      if (ASYM) {
        s0 += __shfl_xor(s0, 32, 64);
        s0 += (shift - zpi) * (R * g.S * g.C);
        s0f2 = f32x2{(float)s0, (float)s0};
      }
      uint8_t* cst = reinterpret_cast<uint8_t*>(stg);       // this wave's stage: 32 rows x 80 bytes of codes
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = j * 32 + hsel * 16;
        // the chain (sum * scale) + bias (+ sum of codes * offset) on PAIRS of channels: v_pk_mul_f32 / v_pk_add_f32, the same roundings
        f32x4 y[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 mu = *reinterpret_cast<const f32x4*>(ctab + cb + 4 * q4);
          const f32x4 bs = *reinterpret_cast<const f32x4*>(ctab + 128 + cb + 4 * q4);
          f32x2 ya = f32x2{(float)acc[j][4 * q4], (float)acc[j][4 * q4 + 1]} * f32x2{mu.x, mu.y} + f32x2{bs.x, bs.y};
          f32x2 yb = f32x2{(float)acc[j][4 * q4 + 2], (float)acc[j][4 * q4 + 3]} * f32x2{mu.z, mu.w} + f32x2{bs.z, bs.w};
          if (ASYM) {
            const f32x4 wo = *reinterpret_cast<const f32x4*>(ctab + 192 + cb + 4 * q4);
            ya = ya + s0f2 * f32x2{wo.x, wo.y};
            yb = yb + s0f2 * f32x2{wo.z, wo.w};
          }
          y[q4] = f32x4{ya.x, ya.y, yb.x, yb.y};
        }
        uint32_t wq[4];
        if (plainq) {
          eq.code4n_plain(y, wq);
        } else {
          bool uq[4];
          eq.code4n(y, wq, uq);
        }
        *reinterpret_cast<i32x4*>(cst + (lane & 31) * 80 + cb) = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave reads back only what it wrote itself
#pragma unroll
      for (int it = 0; it < 2; ++it) {                     // 16 rows x 64 bytes per pass: whole rows
        const int r = it * 16 + (lane >> 2);
        const int64_t row = m0 + r;
        const i32x4 c16 = *reinterpret_cast<const i32x4*>(cst + r * 80 + (lane & 3) * 16);
        if (row < g.M) __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(ep.codes + row * g.K + n0 + (lane & 3) * 16));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next tile overwrites the stage
      continue;
    }
    float s0r[ASYM ? 16 : 1];
    if (ASYM) {      // lanes p and p + 32 hold the two tap halves of pixel p; the sums go through LDS into the accumulator layout
      s0 += __shfl_xor(s0, 32, 64);
      s0 += (shift - zpi) * (R * g.S * g.C);
      if (hsel == 0) s0tab[wave * 32 + (lane & 31)] = s0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (a wave reads only its own 32 sums)
#pragma unroll
      for (int i = 0; i < 16; ++i) s0r[ASYM ? i : 0] = (float)s0tab[wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * hsel];
    }
    // ---- epilogue: accumulator layout (lane = channel, register = pixel) -> pixel-major through this wave's stage ----
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2) + 4 * hsel;
        float v = (float)(acc[j][i] + corr[j]) * mult[j] + bv[j];
        if (ASYM) v = v + s0r[ASYM ? i : 0] * woff[j];
        stg[r * ST_EP_LD + j * 32 + (lane & 31)] = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave reads back only what it wrote itself
    const int col = n0 + ec;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 4 + er;
      const int64_t row = m0 + r;
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * ST_EP_LD + ec);
      if (row < g.M && col < g.K) {
        const int64_t at = row * g.K + col;
        if (ep.relu) v = f32x4{relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)};
        if (out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + at));
        if (ep.codes) __builtin_nontemporal_store(eq.code4(v), reinterpret_cast<uint32_t*>(ep.codes + at));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next tile overwrites the stage
  }
}

// ------------------------------------------------------------------ first layer + 3x3/2 max-pool in one kernel
// The ResNet stem: conv -> ReLU -> MaxPool2d(3, 2, 1) -> the next layers' quantiser.  Done separately the convolution has
// to quantise and write all P x Q outputs (4x the pooled count; that epilogue, not the MFMAs, is its cost) and the pool
// reads them back.  Here a workgroup owns a 3 x 8 tile of POOLED pixels: it convolves the 7 x 17 conv pixels under it
// (1.24x recomputation at the tile seams; four 32-pixel MFMA tiles, one per wave), parks the
// integer sums in LDS, max-pools THEM (the dequantisation is monotone per channel), and dequantises, rectifies and
// quantises only the 24 pooled pixels: the same fp32 values and codes as ReLU -> max-pool -> fake-quant on the full map.
constexpr int SP_TH = 3, SP_TW = 8;                       // pooled tile (7 x 17 = 119 conv pixels = 4 MFMA tiles: one per wave)
constexpr int SP_RH = 2 * SP_TH + 1, SP_RW = 2 * SP_TW + 1;   // conv region under it (window 3, stride 2)
constexpr int SP_NPIX = SP_RH * SP_RW;                    // 119
constexpr int SP_TILES = (SP_NPIX + 31) / 32;             // 4
constexpr int SP_LD = 68;

struct StemPoolGeom {
  int N, Hp, Wp, K, stride, P, Q, PP, QP, tiles_h, tiles_w;
  uint32_t nwork;
  FastDiv twdiv, thdiv;
};


template <int R>
__global__ __launch_bounds__(256, 3) void conv_stem_pool_i8_kernel(const uint8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                float* __restrict__ out, const float* __restrict__ bias,
                                                                const int32_t* __restrict__ wsum, const float* __restrict__ s_in,
                                                                const float* __restrict__ zp_in, const float* __restrict__ s_w,
                                                                StemPoolGeom g, int shift, ConvEpi ep) {
  __shared__ __attribute__((aligned(16))) int region[SP_TILES * 32 * SP_LD];   // conv pixels x 64 channels: the int32 sums (+ corr)
  // The filter bank (R rows x 64 channels x 32 bytes) lives in LDS, not in 8 R registers per lane: 136 instead of 192
  // registers = three waves per SIMD instead of two, in a kernel whose items are short chains of dependent phases
  // (operands -> 14 MFMAs -> LDS -> barrier -> pool -> barrier).  16-byte half h of channel k sits at slot h ^ (k >> 3 & 1):
  // conflict-free ds_read_b128 for a fragment read (lane = channel, 32-byte rows).
  __shared__ i32x4 wbank[R * 64 * 2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hsel = lane >> 5;
  for (int i = threadIdx.x; i < R * 64 * 2; i += 256) {
    const int r = i >> 7, k = (i >> 1) & 63, h = i & 1;
    wbank[(r * 64 + k) * 2 + (h ^ ((k >> 3) & 1))] =
        k < g.K ? *reinterpret_cast<const i32x4*>(w + ((int64_t)k * R + r) * 32 + h * 16) : i32x4{0, 0, 0, 0};
  }
  __syncthreads();
  const uint32_t xorw = shift ? 0x80808080u : 0u;
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin = s_in[0];
  const EpiQuant eq(ep);
  const i32x4* const bfp = wbank + (lane & 31) * 2 + (hsel ^ ((lane >> 3) & 1));     // + (r * 64 + j * 32) * 2
  float mult[2], bv[2];
  int corr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = j * 32 + (lane & 31);
    const bool ok = k < g.K;
    mult[j] = ok ? sin * s_w[k] : 0.0f;
    corr[j] = ok ? (shift - zpi) * wsum[k] : 0;
    bv[j] = (ok && bias) ? bias[k] : 0.0f;
  }
  const int rowbytes = g.Wp * 4;
  // the pool phase dequantises channels c4 .. c4+3 of its pixels: their constants, once
  float pm[4], pb[4];
  int pc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = (threadIdx.x & 15) * 4 + c;
    pm[c] = k < g.K ? sin * s_w[k] : 0.0f;
    pb[c] = (k < g.K && bias) ? bias[k] : 0.0f;
    pc[c] = k < g.K ? (shift - zpi) * wsum[k] : 0;
  }
  const bool anyneg = __ballot(mult[0] < 0.0f || mult[1] < 0.0f) != 0;
  // persistent workgroups: the weight fragments above are loaded once, then (image, pooled tile) after tile; wave w owns
  // conv tile w of every item, and its operand loads for the NEXT item are issued before this item's epilogue and pool
  static_assert(SP_TILES == 4, "one conv tile per wave");
  const int idx = wave * 32 + (lane & 31);
  const int ry = idx / SP_RW, rx = idx - ry * SP_RW;
  auto locate = [&](uint32_t work, uint32_t& n, int& th, int& tw, bool& valid) -> const uint8_t* {
    const uint32_t t0 = fdiv(work, g.twdiv);
    tw = (int)(work - t0 * (uint32_t)g.tiles_w);
    n = fdiv(t0, g.thdiv);
    th = (int)(t0 - n * (uint32_t)g.tiles_h);
    const int p = 2 * th * SP_TH - 1 + ry, q = 2 * tw * SP_TW - 1 + rx;
    valid = idx < SP_NPIX && p >= 0 && p < g.P && q >= 0 && q < g.Q;
    const int pc = valid ? p : 0, qc = valid ? q : 0;            // an address that exists; the value is discarded
    return x + (((int64_t)n * g.Hp + (int64_t)pc * g.stride) * g.Wp + (int64_t)qc * g.stride) * 4 + hsel * 16;
  };
  // Operands travel TWO items ahead (two fragment sets), by loads the compiler does not see (gload16) and one counted wait per
  // item.  vmcnt counts loads and stores in issue order: with a distance of one, the wait for item i+1's operands at the top
  // of the loop also waited for item i's stores, issued just before - a store round trip exposed per item (and hipcc drains the
  // queue, vmcnt(0), whenever loads and stores are pending together).  With a distance of two the queue at the top of item k is
  //   L(k) | stores(k-2) | L(k+1) | stores(k-1)      (7 loads per set, at most 4 stores per wave and item)
  // and `s_waitcnt vmcnt(7)` retires L(k) while stores(k-1) and the rest of L(k+1) stay in flight.  Loads past the last item
  // re-read the last item (the count must not depend on the item); the queue is drained before the wave ends.
  static_assert(R <= 7, "the counted wait below assumes at most 7 loads per set (one per filter row)");
  if (blockIdx.x >= g.nwork) return;
  i32x4 af[2][R];
  uint32_t n_q[2] = {0, 0};
  int th_q[2] = {0, 0}, tw_q[2] = {0, 0};
  bool valid_q[2] = {false, false};
  auto fetch = [&](uint32_t w0, auto b_c) {
    constexpr int b = decltype(b_c)::value;
    const uint8_t* src = locate(w0 < g.nwork ? w0 : g.nwork - 1, n_q[b], th_q[b], tw_q[b], valid_q[b]);
#pragma unroll
    for (int r = 0; r < R; ++r) gload16<0>(af[b][r], reinterpret_cast<const int8_t*>(src + (int64_t)r * rowbytes));
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the weight fragments and constants above: nothing of the compiler's in the queue)
  fetch(blockIdx.x, std::integral_constant<int, 0>{});
  fetch(blockIdx.x + gridDim.x, std::integral_constant<int, 1>{});
  auto item = [&](uint32_t work, auto b_c) {
    constexpr int b = decltype(b_c)::value;
    const uint32_t n = n_q[b];
    const int th = th_q[b], tw = tw_q[b];
    const bool valid = valid_q[b];
    if constexpr (R == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (shorter filters: fewer loads per set - not tuned)
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(af[b][r]));   // valid from here on
    i32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const i32x4 a = i32x4{(int)(af[b][r].x ^ xorw), (int)(af[b][r].y ^ xorw), (int)(af[b][r].z ^ xorw), (int)(af[b][r].w ^ xorw)};
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bfp[(r * 64 + j * 32) * 2], acc[j], 0, 0, 0);
    }
    fetch(work + 2 * gridDim.x, b_c);                // the operands of the item after next
    // which of this tile's pixels exist: row r of the accumulator is pixel wave*32 + r, owned by lane r of each half-wave
    // What is parked in LDS is the exact integer sum a = acc + corr, not its dequantised value: v(a) = fl(fl(float(a)*m) + b)
    // and ReLU are monotone in a for m >= 0, so max-pooling a and THEN dequantising the pooled quarter gives the same fp32
    // values as pooling v - at one integer add per conv output instead of eight operations.  (m < 0 - a negative scale, which
    // no observer of this library produces - flips the order: those channels store -a and are negated back after the pool.)
    // Rows >= SP_NPIX of the last tile are never read by the pool, so only image borders need the validity mask: interior
    // tiles - the vast majority - skip it on a wave-uniform branch.
    const uint32_t vmask = (uint32_t)__ballot(valid || idx >= SP_NPIX);
    const uint32_t vm = hsel ? (vmask >> 4) : vmask;           // bit (i&3) + 8(i>>2) = row r of this half-wave
    int* dst = region + (wave * 32 + 4 * hsel) * SP_LD + (lane & 31);
    if (vmask == 0xffffffffu && !anyneg) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[((i & 3) + 8 * (i >> 2)) * SP_LD + j * 32] = acc[j][i];
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          int a = acc[j][i] + corr[j];      // (negated below: the constant must be inside)
          if (mult[j] < 0.0f) a = -a;
          else a = acc[j][i];
          if (!((vm >> ((i & 3) + 8 * (i >> 2))) & 1u)) a = INT32_MIN;     // outside the image: never wins the pool
          dst[((i & 3) + 8 * (i >> 2)) * SP_LD + j * 32] = a;
        }
    }
  __syncthreads();
  // ---- pool 3x3 / 2 in fp32, then the consumer's quantiser on the pooled pixels: up to 2 float4 per thread ----
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int pit = threadIdx.x + u * 256;
    if (pit >= SP_TH * SP_TW * 16) continue;
    const int pp = pit >> 4, c4 = (pit & 15) * 4;
    const int py = pp / SP_TW, px = pp - py * SP_TW;
    const int ph = th * SP_TH + py, pw = tw * SP_TW + px;
    if (ph >= g.PP || pw >= g.QP || c4 >= g.K) continue;
    i32x4 rmax[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {     // three-operand maxima (v_max3_i32): 4 instead of 8 per channel
      const int* rowp = region + ((2 * py + dy) * SP_RW + 2 * px) * SP_LD + c4;
      const i32x4 v0 = *reinterpret_cast<const i32x4*>(rowp), v1 = *reinterpret_cast<const i32x4*>(rowp + SP_LD),
                  v2 = *reinterpret_cast<const i32x4*>(rowp + 2 * SP_LD);
      rmax[dy] = i32x4{max(max(v0.x, v1.x), v2.x), max(max(v0.y, v1.y), v2.y), max(max(v0.z, v1.z), v2.z), max(max(v0.w, v1.w), v2.w)};
    }
    i32x4 am = i32x4{max(max(rmax[0].x, rmax[1].x), rmax[2].x), max(max(rmax[0].y, rmax[1].y), rmax[2].y),
                     max(max(rmax[0].z, rmax[1].z), rmax[2].z), max(max(rmax[0].w, rmax[1].w), rmax[2].w)};
    // the per-channel constant (shift - zp) * SUM qw commutes with the maximum: added to the pooled quarter only (channels
    // with a negative scale carry it inside, negated, and are negated back here)
    f32x4 m;
    m.x = (float)(pm[0] < 0.0f ? -am.x : am.x + pc[0]) * pm[0] + pb[0];
    m.y = (float)(pm[1] < 0.0f ? -am.y : am.y + pc[1]) * pm[1] + pb[1];
    m.z = (float)(pm[2] < 0.0f ? -am.z : am.z + pc[2]) * pm[2] + pb[2];
    m.w = (float)(pm[3] < 0.0f ? -am.w : am.w + pc[3]) * pm[3] + pb[3];
    if (ep.relu) m = f32x4{relu_nan(m.x), relu_nan(m.y), relu_nan(m.z), relu_nan(m.w)};
    const int64_t at = (((int64_t)n * g.PP + ph) * g.QP + pw) * g.K + c4;
    if (out) __builtin_nontemporal_store(m, reinterpret_cast<f32x4*>(out + at));
    if (ep.codes) __builtin_nontemporal_store(eq.code4(m), reinterpret_cast<uint32_t*>(ep.codes + at));
  }
  __syncthreads();     // the region is rewritten by the next item
  };
  for (uint32_t work = blockIdx.x;;) {
    item(work, std::integral_constant<int, 0>{});
    work += gridDim.x;
    if (work >= g.nwork) break;
    item(work, std::integral_constant<int, 1>{});
    work += gridDim.x;
    if (work >= g.nwork) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the loads issued past the last item
}

// ------------------------------------------------------------------------- max-pool on codes
struct PoolGeom {
  int N, H, W, C4, k, stride, pad, P, Q;   // C4 = C / 4 (dwords per pixel)
  FastDiv cdiv, qdiv, pdiv;
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// running bytewise maximum of dwords, kept as two packed-u16 halves (even bytes, odd bytes): v_pk_max_u16
struct ByteMax {
  uint32_t even = 0, odd = 0;
  __device__ __forceinline__ void take(uint32_t v) {
    const uint32_t e = v & 0x00ff00ffu, o = (v >> 8) & 0x00ff00ffu;
    even = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, even), __builtin_bit_cast(u16x2, e)));
    odd = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, odd), __builtin_bit_cast(u16x2, o)));
  }
  __device__ __forceinline__ uint32_t get() const { return even | (odd << 8); }
};

// one thread = V dwords (4 V channels) of one output pixel; signed codes are compared as unsigned after flipping the
// sign bit; padding never wins: the lowest code is 0 after the flip
template <int V>
__global__ __launch_bounds__(DLMCQ_BLOCK) void maxpool_codes_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y,
                                                                    PoolGeom g, uint32_t flip) {
  const int cv = g.C4 / V;   // vectors per pixel (g.cdiv divides by this)
  const int64_t total = (int64_t)g.N * g.P * g.Q * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t pix = fdiv((uint32_t)i, g.cdiv);
    const int c = (int)((uint32_t)i - pix * (uint32_t)cv) * V;
    const uint32_t t = fdiv(pix, g.qdiv);
    const int q = (int)(pix - t * (uint32_t)g.Q);
    const uint32_t n = fdiv(t, g.pdiv);
    const int p = (int)(t - n * (uint32_t)g.P);
    const int h0 = p * g.stride - g.pad, w0 = q * g.stride - g.pad;
    ByteMax m[V];
    for (int dh = 0; dh < g.k; ++dh) {
      const int h = h0 + dh;
      if (h < 0 || h >= g.H) continue;
      for (int dw = 0; dw < g.k; ++dw) {
        const int w = w0 + dw;
        if (w < 0 || w >= g.W) continue;
        const uint32_t* src = x + (((int64_t)n * g.H + h) * g.W + w) * g.C4 + c;
        if (V == 4) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(src);
          m[0].take(v.x ^ flip);
          m[1 % V].take(v.y ^ flip);
          m[2 % V].take(v.z ^ flip);
          m[3 % V].take(v.w ^ flip);
        } else {
          m[0].take(src[0] ^ flip);
        }
      }
    }
    uint32_t* dst = y + (int64_t)pix * g.C4 + c;
    if (V == 4) __builtin_nontemporal_store(u32x4{m[0].get() ^ flip, m[1 % V].get() ^ flip, m[2 % V].get() ^ flip, m[3 % V].get() ^ flip},
                                            reinterpret_cast<u32x4*>(dst));
    else dst[0] = m[0].get() ^ flip;
  }
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_quantize_pad_nhwc4(const float* x, void* out, const float* scale, const float* zero_point, int64_t N,
                                        int64_t C, int64_t H, int64_t W, int64_t stride_n, int64_t stride_c, int64_t stride_h,
                                        int64_t stride_w, int32_t pad, int32_t lo, int32_t hi, int32_t form, float ste_g,
                                        dlmcq_stream_t stream) {
  if (N < 0 || C < 1 || C > 4 || H < 1 || W < 1 || pad < 0 || lo > hi || lo < -128 || hi > 255 || hi - lo > 255)
    return DLMCQ_EINVAL;
  ConvEpi q{};                          // the image quantiser, evaluated by EpiQuant (needs a non-null `codes` to resolve)
  if (!epi_set_form(q, form, lo, hi)) return DLMCQ_EINVAL;      // (DLMCQ_EMIT_SHIFT128: the buffer holds `code - 128`, border included)
  if (N == 0) return DLMCQ_OK;
  if (!x || !out || !scale) return DLMCQ_EINVAL;
  if (!aligned4(out)) return DLMCQ_EALIGN;
  ImgGeom g;
  g.N = (int)N; g.C = (int)C; g.H = (int)H; g.W = (int)W; g.pad = pad;
  g.Hp = (int)H + 2 * pad; g.Wp = (int)W + 2 * pad;
  g.sn = stride_n; g.sc = stride_c; g.sh = stride_h; g.sw = stride_w;
  const int64_t total = N * g.Hp * g.Wp;
  if (total >= (1ll << 31)) return DLMCQ_ERANGE;
  g.wdiv = make_fastdiv((uint32_t)g.Wp);
  g.hdiv = make_fastdiv((uint32_t)g.Hp);
  const int64_t blocks = (total + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
  q.codes = static_cast<uint8_t*>(out);
  q.q_scale = scale;
  q.q_zp = zero_point;
  q.q_lo = (float)lo;
  q.q_hi = (float)hi;
  q.q_g = ste_g;
  if (W % 4 == 0 && stride_w == 1 && stride_h % 4 == 0 && stride_n % 4 == 0 && (C == 1 || stride_c % 4 == 0) && aligned16(x)) {
    const int groups = (int)(W / 4);
    const int64_t items = N * g.Hp * groups;
    const int64_t b4 = (items + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
    hipLaunchKernelGGL(quantize_pad_nhwc4_x4_kernel, dim3((uint32_t)(b4 < 65536 ? b4 : 65536)), dim3(DLMCQ_BLOCK), 0,
                       reinterpret_cast<hipStream_t>(stream), x, static_cast<uint32_t*>(out), g, q, make_fastdiv((uint32_t)groups), groups);
    return launch_status();
  }
  hipLaunchKernelGGL(quantize_pad_nhwc4_kernel, dim3((uint32_t)(blocks < 65536 ? blocks : 65536)), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), x, static_cast<uint32_t*>(out), g, q);
  return launch_status();
}

extern "C" int dlmcq_quantize_weight_stem_i8(const float* w, int8_t* wq, int32_t* wsum, const float* scale, int64_t K,
                                             int64_t C, int64_t R, int64_t S, int32_t lo, int32_t hi,
                                             dlmcq_stream_t stream) {
  if (K < 1 || C < 1 || C > 4 || R < 1 || R > ST_MAXR || S < 1 || S > 8 || lo > hi || lo < -128 || hi > 127)
    return DLMCQ_EINVAL;
  if (!w || !wq || !wsum || !scale) return DLMCQ_EINVAL;
  if (!aligned16(wq)) return DLMCQ_EALIGN;
  hipLaunchKernelGGL(quantize_weight_stem_kernel, dim3((uint32_t)K), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), w, wq, wsum, scale, (int)C, (int)R, (int)S, (float)lo, (float)hi);
  return launch_status();
}

static int stem_launch(const void* xpad, const int8_t* w, float* out, const float* bias, const int32_t* wsum, const float* in_scale,
                       const float* in_zero_point, const float* w_scale, const float* w_offset, int64_t C, int64_t N, int64_t Hp,
                       int64_t Wp, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t x_is_unsigned, int32_t relu, void* codes,
                       const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g,
                       dlmcq_stream_t stream) {
  if (N < 0 || Hp < 1 || Wp < 1 || K < 1 || R < 1 || R > ST_MAXR || S < 1 || S > 8 || stride < 1) return DLMCQ_EINVAL;
  if (Hp < R || Wp < S || (K & 3)) return DLMCQ_EINVAL;
  const int64_t P = (Hp - R) / stride + 1, Q = (Wp - S) / stride + 1;
  const int64_t M = N * P * Q;
  if (M == 0) return DLMCQ_OK;
  if (!xpad || !w || !(out || codes) || !wsum || !in_scale || !w_scale) return DLMCQ_EINVAL;
  if (codes && (!q_scale || q_lo > q_hi || q_lo < -128 || q_hi > 255 || q_hi - q_lo > 255 || q_form < DLMCQ_FORM_EMULATE ||
                q_form > DLMCQ_FORM_SYMMETRIC))
    return DLMCQ_EINVAL;
  if (!aligned4(xpad) || !aligned16(w) || (out && !aligned16(out)) || (codes && !aligned4(codes))) return DLMCQ_EALIGN;
  if (M >= (1ll << 31) || N * Hp * Wp >= (1ll << 31)) return DLMCQ_ERANGE;
  StemGeom g;
  g.N = (int)N; g.Hp = (int)Hp; g.Wp = (int)Wp; g.K = (int)K; g.R = (int)R; g.stride = stride; g.P = (int)P; g.Q = (int)Q;
  g.S = (int)S; g.C = (int)C;
  g.M = M;
  g.ntiles = (int)((M + 31) / 32);
  g.qdiv = make_fastdiv((uint32_t)Q);
  g.pdiv = make_fastdiv((uint32_t)P);
  ConvEpi ep{};
  ep.w_off = w_offset;
  ep.relu = relu != 0;
  ep.codes = static_cast<uint8_t*>(codes);
  ep.q_scale = q_scale;
  ep.q_zp = q_zero_point;
  ep.q_lo = (float)q_lo;
  ep.q_hi = (float)q_hi;
  ep.q_g = q_ste_g;
  ep.q_form = q_form;
  const int wgs = (g.ntiles + 3) / 4;
  const dim3 grid((uint32_t)(wgs < 2048 ? wgs : 2048), (uint32_t)((K + 63) / 64));
  const int shift = x_is_unsigned ? 128 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint8_t* xs = static_cast<const uint8_t*>(xpad);
#define DLMCQ_STEM_ARGS grid, dim3(256), 0, st, xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep
#define DLMCQ_STEM_CASE(RR) \
  case RR:                  \
    if (w_offset) hipLaunchKernelGGL((conv_stem_i8_kernel<RR, true>), DLMCQ_STEM_ARGS); \
    else hipLaunchKernelGGL((conv_stem_i8_kernel<RR, false>), DLMCQ_STEM_ARGS);         \
    break;
  // 3 x 3 first layers that emit only codes (RepVGG, MobileOne): the swapped epilogue
  if (R == 3 && !out && codes && K % 64 == 0 && aligned16(codes)) {
    if (w_offset) hipLaunchKernelGGL((conv_stem_i8_kernel<3, true, true>), DLMCQ_STEM_ARGS);
    else hipLaunchKernelGGL((conv_stem_i8_kernel<3, false, true>), DLMCQ_STEM_ARGS);
    return launch_status();
  }
  switch ((int)R) {
    DLMCQ_STEM_CASE(1) DLMCQ_STEM_CASE(2) DLMCQ_STEM_CASE(3) DLMCQ_STEM_CASE(4) DLMCQ_STEM_CASE(5) DLMCQ_STEM_CASE(6)
    default:
      if (w_offset) hipLaunchKernelGGL((conv_stem_i8_kernel<7, true>), DLMCQ_STEM_ARGS);
      else hipLaunchKernelGGL((conv_stem_i8_kernel<7, false>), DLMCQ_STEM_ARGS);
      break;
  }
#undef DLMCQ_STEM_CASE
#undef DLMCQ_STEM_ARGS
  return launch_status();
}

extern "C" int dlmcq_conv2d_i8_stem_fused(const void* xpad, const int8_t* w, float* out, const float* bias,
                                          const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                          const float* w_scale, int64_t N, int64_t Hp, int64_t Wp, int64_t K, int64_t R,
                                          int64_t S, int32_t stride, int32_t x_is_unsigned, int32_t relu, void* codes,
                                          const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                                          int32_t q_form, float q_ste_g, dlmcq_stream_t stream) {
  return stem_launch(xpad, w, out, bias, wsum, in_scale, in_zero_point, w_scale, nullptr, 4, N, Hp, Wp, K, R, S, stride,
                     x_is_unsigned, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g, stream);
}

extern "C" int dlmcq_conv2d_i8_stem_asym(const void* xpad, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                         const float* in_scale, const float* in_zero_point, const float* w_scale,
                                         const float* w_offset, int64_t C, int64_t N, int64_t Hp, int64_t Wp, int64_t K, int64_t R,
                                         int64_t S, int32_t stride, int32_t x_is_unsigned, int32_t relu, void* codes,
                                         const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                                         int32_t q_form, float q_ste_g, dlmcq_stream_t stream) {
  if (!w_offset || C < 1 || C > 4) return DLMCQ_EINVAL;
  return stem_launch(xpad, w, out, bias, wsum, in_scale, in_zero_point, w_scale, w_offset, C, N, Hp, Wp, K, R, S, stride,
                     x_is_unsigned, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g, stream);
}

extern "C" int dlmcq_maxpool_codes_nhwc(const void* x, void* y, int64_t N, int64_t H, int64_t W, int64_t C, int32_t kernel,
                                        int32_t stride, int32_t pad, int32_t x_is_unsigned, dlmcq_stream_t stream) {
  if (N < 0 || H < 1 || W < 1 || C < 4 || (C & 3) || kernel < 1 || stride < 1 || pad < 0 || 2 * pad > kernel) return DLMCQ_EINVAL;
  const int64_t P = (H + 2 * pad - kernel) / stride + 1, Q = (W + 2 * pad - kernel) / stride + 1;
  if (P < 1 || Q < 1) return DLMCQ_EINVAL;
  if (N == 0) return DLMCQ_OK;
  if (!x || !y) return DLMCQ_EINVAL;
  if (!aligned4(x) || !aligned4(y)) return DLMCQ_EALIGN;
  const bool vec = (C % 16) == 0 && aligned16(x) && aligned16(y);
  const int64_t total = N * P * Q * (C / (vec ? 16 : 4));
  if (N * P * Q * (C / 4) >= (1ll << 31) || N * H * W * (C / 4) >= (1ll << 31)) return DLMCQ_ERANGE;
  PoolGeom g;
  g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C4 = (int)(C / 4); g.k = kernel; g.stride = stride; g.pad = pad;
  g.P = (int)P; g.Q = (int)Q;
  g.cdiv = make_fastdiv((uint32_t)(vec ? g.C4 / 4 : g.C4));
  g.qdiv = make_fastdiv((uint32_t)Q);
  g.pdiv = make_fastdiv((uint32_t)P);
  const int64_t blocks = (total + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK;
  const dim3 grid((uint32_t)(blocks < 65536 ? blocks : 65536));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint32_t flip = x_is_unsigned ? 0u : 0x80808080u;
  if (vec) hipLaunchKernelGGL((maxpool_codes_kernel<4>), grid, dim3(DLMCQ_BLOCK), 0, st, static_cast<const uint32_t*>(x), static_cast<uint32_t*>(y), g, flip);
  else hipLaunchKernelGGL((maxpool_codes_kernel<1>), grid, dim3(DLMCQ_BLOCK), 0, st, static_cast<const uint32_t*>(x), static_cast<uint32_t*>(y), g, flip);
  return launch_status();
}

extern "C" int dlmcq_conv2d_i8_stem_pool_fused(const void* xpad, const int8_t* w, float* out, const float* bias,
                                               const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                               const float* w_scale, int64_t N, int64_t Hp, int64_t Wp, int64_t K, int64_t R,
                                               int64_t S, int32_t stride, int32_t x_is_unsigned, int32_t relu, void* codes,
                                               const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                                               int32_t q_form, float q_ste_g, dlmcq_stream_t stream) {
  if (N < 0 || Hp < 1 || Wp < 1 || K < 4 || K > 64 || (K & 3) || R < 1 || R > ST_MAXR || S < 1 || S > 8 || stride < 1)
    return DLMCQ_EINVAL;
  if (Hp < R || Wp < S) return DLMCQ_EINVAL;
  const int64_t P = (Hp - R) / stride + 1, Q = (Wp - S) / stride + 1;
  const int64_t PP = (P + 2 - 3) / 2 + 1, QP = (Q + 2 - 3) / 2 + 1;     // MaxPool2d(3, 2, 1), floor mode
  if (PP < 1 || QP < 1) return DLMCQ_EINVAL;
  if (N == 0) return DLMCQ_OK;
  if (!xpad || !w || !(out || codes) || !wsum || !in_scale || !w_scale) return DLMCQ_EINVAL;
  if (codes && (!q_scale || q_lo > q_hi || q_lo < -128 || q_hi > 255 || q_hi - q_lo > 255 || q_form < DLMCQ_FORM_EMULATE ||
                q_form > DLMCQ_FORM_SYMMETRIC))
    return DLMCQ_EINVAL;
  if (!aligned4(xpad) || !aligned16(w) || (out && !aligned16(out)) || (codes && !aligned4(codes))) return DLMCQ_EALIGN;
  StemPoolGeom g;
  g.N = (int)N; g.Hp = (int)Hp; g.Wp = (int)Wp; g.K = (int)K; g.stride = stride; g.P = (int)P; g.Q = (int)Q;
  g.PP = (int)PP; g.QP = (int)QP;
  g.tiles_h = (int)((PP + SP_TH - 1) / SP_TH);
  g.tiles_w = (int)((QP + SP_TW - 1) / SP_TW);
  const int64_t wgs = N * g.tiles_h * g.tiles_w;
  if (wgs >= (1ll << 31) || N * Hp * Wp >= (1ll << 31)) return DLMCQ_ERANGE;
  g.twdiv = make_fastdiv((uint32_t)g.tiles_w);
  g.thdiv = make_fastdiv((uint32_t)g.tiles_h);
  g.nwork = (uint32_t)wgs;
  ConvEpi ep{};
  ep.relu = relu != 0;
  ep.codes = static_cast<uint8_t*>(codes);
  ep.q_scale = q_scale;
  ep.q_zp = q_zero_point;
  ep.q_lo = (float)q_lo;
  ep.q_hi = (float)q_hi;
  ep.q_g = q_ste_g;
  ep.q_form = q_form;
  const int shift = x_is_unsigned ? 128 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const uint8_t* xs = static_cast<const uint8_t*>(xpad);
  // the ResNet shape (7 rows, stride 2, 64 channels, codes only): pooling in registers (conv_stem_pool7_i8.hip)
  if (stem_pool7_applies(Hp, Wp, K, R, S, stride, out, codes) && aligned16(codes) && aligned16(xpad))
    return stem_pool7_launch(xs, w, bias, wsum, in_scale, in_zero_point, w_scale, N, Hp, Wp, S, shift, ep, st);
#define DLMCQ_SP_ARGS dim3((uint32_t)(wgs < 1024 ? wgs : 1024)), dim3(256), 0, st, xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep
  switch ((int)R) {
    case 1: hipLaunchKernelGGL((conv_stem_pool_i8_kernel<1>), DLMCQ_SP_ARGS); break;
    case 2: hipLaunchKernelGGL((conv_stem_pool_i8_kernel<2>), DLMCQ_SP_ARGS); break;
    case 3: hipLaunchKernelGGL((conv_stem_pool_i8_kernel<3>), DLMCQ_SP_ARGS); break;
    case 4: hipLaunchKernelGGL((conv_stem_pool_i8_kernel<4>), DLMCQ_SP_ARGS); break;
    case 5: hipLaunchKernelGGL((conv_stem_pool_i8_kernel<5>), DLMCQ_SP_ARGS); break;
    case 6: hipLaunchKernelGGL((conv_stem_pool_i8_kernel<6>), DLMCQ_SP_ARGS); break;
    default: hipLaunchKernelGGL((conv_stem_pool_i8_kernel<7>), DLMCQ_SP_ARGS); break;
  }
#undef DLMCQ_SP_ARGS
  return launch_status();
}

