// The ResNet first layer - 7x7 / 2 convolution of the 3-channel image + ReLU + MaxPool2d(3, 2, 1) + the consumer's quantiser -
// in the shape the counters asked for (round 3).  conv_stem_i8.hip's conv_stem_pool_i8_kernel parks the int32 sums of 119
// convolution pixels x 64 channels in LDS per item and reads them back 2.25 times to pool them: it is bound by that LDS traffic
// (0.30 ms at batch 512, 0.09 of the HBM roofline) and fetches every operand fragment from global memory, 14x over.  Here the
// pooling never leaves the registers (reference ops being replaced: F.conv2d of the fake-quantised image, modules/conv.py:13-19,
// then the model's relu / maxpool, then FSPTQuant/base.py:108-109 of the next layer):
//
//   * v_mfma_i32_32x32x32_i8 with the weights as A and 32 convolution pixels of one row as B, so a lane IS a pixel and its 16
//     accumulator registers of a block are 16 consecutive channels (conv_i8.hip's swapped layout).
//   * Two accumulator sets per convolution row: E = the EVEN columns 2q, O = the ODD columns 2q - 1, lane q of both belonging to
//     pooled column q.  The 3-wide horizontal maximum is then max(O[q], E[q], O[q + 1]): two registers of the lane itself and one
//     of its neighbour (DPP wave_shl:1) - every lane ends up with a pooled pixel, none is wasted.  31 pooled columns per tile
//     (lane 31 has no right neighbour), two tiles per 56-wide row.
//   * The 3-high vertical maximum runs down the image: a wave walks a band of pooled rows, convolves two new rows per pooled
//     row and keeps the running maximum of the rows' horizontal maxima - 32 registers (the lower row of one window is the upper
//     row of the next).  Maxima are taken on the
//     exact integer sums (the dequantisation is monotone per channel; channels with a negative scale are bit-flipped, which
//     reverses their order, and flipped back after the pool), so only the pooled quarter is dequantised, rectified and quantised.
//   * Operands: a wave stages the four new image rows of a pooled row (536 bytes of its 134-pixel window each) into a private
//     16-row LDS ring by LDS-DMA, one pooled row ahead; the O fragments are aligned ds_read_b128 of it, the E fragments (8 bytes
//     further) two ds_read_b64.  Nothing is shared between waves: no barrier in the loop.  The weights (7 rows x 64 channels x
//     32 bytes) sit in LDS in fragment order (one linear ds_read_b128 per MFMA pair).
//   * The 31 x 64 code bytes of a pooled row are one contiguous 1 984-byte run of the NHWC output: staged through a private
//     2 KB LDS tile, stored by two full-width instructions.
// Shape served: 7 filter rows, <= 8 taps, stride 2, 64 output channels, codes only.  Everything else: conv_stem_pool_i8_kernel.
#include "conv_i8_common.h"

namespace dlmcq {

struct Pool7Geom {
  int N, Hp, Wp, P, Q, PP, QP;      // padded image, convolution output, pooled output
  int bands, rows_per_band, xtiles;
  uint32_t ntasks;
  FastDiv xtdiv, bdiv;
  int64_t xbytes;                   // bytes of the padded image buffer (loads are clamped into it)
};

constexpr int P7_ROWB = 544;        // bytes of a staged window row: 34 16-byte units
constexpr int P7_RING = 16 * P7_ROWB;
constexpr int P7_STG = 2048;        // code tile of one pooled row: 32 pixels x 64 bytes
constexpr int P7_NEG = -(1 << 30);  // "this convolution pixel does not exist": never wins a maximum (sums stay below 2^23)

// F32IN (lab builds only; round 5, review item 8: "the first layer in ONE launch, measured not estimated"): the kernel reads the fp32
// image itself - planes [N][C <= 3][H][W], unit pixel stride - and quantises a wave's window rows into its LDS ring (the arithmetic of
// quantize_pad_nhwc4_x4_kernel: EpiQuant::code4 on channel-planar quads of four pixels, bytes dealt to the pixels by v_perm_b32) instead of
// receiving them by LDS-DMA from the padded code buffer a separate launch has written.  Pixels outside the image are loaded as 0.0, whose
// code IS the border code.  Same codes out (tools/stem_fused_lab.py); timing: LABNOTES 16.
struct Pool7Img {
  const float* img;
  int64_t sn, sc, sh;      // element strides of image, channel plane, row
  int H, W, pad, C;
  ConvEpi iq;              // the image quantiser (scale, zero point, range, form, DLMCQ_EMIT_SHIFT128)
};

template <bool XS, bool F32IN = false>      // XS: uint8 codes, re-centred (^ 0x80) on every fragment read; false: the buffer already holds `code - 128` (or int8 codes)
__global__ __launch_bounds__(256, 2) void conv_stem_pool7_i8_kernel(const uint8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                                    const float* __restrict__ bias, const int32_t* __restrict__ wsum,
                                                                    const float* __restrict__ s_in, const float* __restrict__ zp_in,
                                                                    const float* __restrict__ s_w, Pool7Geom g, int shift, ConvEpi ep
#ifdef DLMCQ_LAB
                                                                    , Pool7Img im
#endif
                                                                    ) {
  __shared__ __attribute__((aligned(1024))) int8_t lds[4 * (P7_RING + P7_STG) + 4 * 64 * 4 + 14 * 1024];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  int8_t* const ring = lds + wave * (P7_RING + P7_STG);
  int8_t* const stg = ring + P7_RING;
  int8_t* const par = lds + 4 * (P7_RING + P7_STG);        // mult | corr | bias | order-reversal mask, 64 channels each
  // the weights as A fragments, in fragment order: (filter row r, channel block j) -> 64 lanes x 16 bytes (one linear ds_read_b128);
  // accumulator row d of a block is channel 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3), so that register i of lane half h is channel 16 h + i
  const int wfo = 4 * (P7_RING + P7_STG) + 4 * 64 * 4 + (tid & 63) * 16;       // this lane's fragment of (row 0, block 0), as an offset into lds
  if (tid < 64) {
    const int d = tid & 31, ch = 16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3), h = tid >> 5;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        reinterpret_cast<i32x4*>(par + 4 * 64 * 4)[(r * 2 + j) * 64 + tid] =
            *reinterpret_cast<const i32x4*>(w + ((int64_t)(j * 32 + ch) * 7 + r) * 32 + h * 16);
  }

  // ---- per-channel constants, once per workgroup ----
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  if (tid < 64) {
    const float m = s_in[0] * s_w[tid];
    reinterpret_cast<float*>(par)[tid] = m;
    reinterpret_cast<int*>(par)[64 + tid] = (shift - zpi) * wsum[tid];
    reinterpret_cast<float*>(par)[128 + tid] = bias ? bias[tid] : 0.0f;
    reinterpret_cast<int*>(par)[192 + tid] = m < 0.0f ? -1 : 0;
  }
  __syncthreads();
  bool anyneg = false;
  {
    const int nm = reinterpret_cast<const int*>(par)[192 + lane];
    anyneg = __ballot(nm != 0) != 0;
  }

  // ---- this wave's task: image n, band of pooled rows, x tile ----
  const uint32_t task = blockIdx.x * 4u + (uint32_t)wave;
  if (task >= g.ntasks) return;                            // (no barrier below this line)
  const uint32_t t0 = fdiv(task, g.xtdiv);
  const int xt = (int)(task - t0 * (uint32_t)g.xtiles);
  const uint32_t n = fdiv(t0, g.bdiv);
  const int band = (int)(t0 - n * (uint32_t)g.bands);
  const int q0 = 31 * xt;
  const int p0 = band * g.rows_per_band;
  const int p1 = p0 + g.rows_per_band < g.PP ? p0 + g.rows_per_band : g.PP;

  const uint32_t xorw = XS ? 0x80808080u : 0u;

  // ---- window rows by LDS-DMA: group G(p) = image rows 4 p + 5 .. 4 p + 8 (padded coordinates) -> ring slots ((p + 1) & 3) * 4 ..;
  // piece i of a group = (row i / 34, unit i % 34); three wave-instructions, the third with 8 lanes ----
  const int64_t img0 = (int64_t)n * g.Hp * g.Wp * 4;
  const int64_t win0 = (int64_t)(16 * q0 - 8);                      // byte offset of the window in a row (-8 for the first tile)
  const int rowbytes = g.Wp * 4;
  auto issue_group = [&](int p) {
    int8_t* const dst = ring + (((p + 1) & 3) * 4) * P7_ROWB;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = k * 64 + lane;
      const int r = i / 34, u = i - r * 34;
      int64_t off = img0 + (int64_t)(4 * p + 5 + r) * rowbytes + win0 + u * 16;
      off = off < 0 ? 0 : (off > g.xbytes - 16 ? g.xbytes - 16 : off);   // (bytes outside the buffer only ever reach pixels that do not exist)
      if (i < 4 * 34) __builtin_amdgcn_global_load_lds((gptr_t)(x + off), (lptr_t)(dst + k * 1024), 16, 0, 0);
    }
  };
#ifdef DLMCQ_LAB
  // F32IN: slot i = k * 64 + lane of a group = (row r = i / 36, aligned quad v = i % 36 of image columns 4 (q0 - 2 + v) ..): 36 quads cover the
  // window's 136 pixels (image columns 4 q0 - 5 ..) whatever its alignment; a quad lies wholly inside or wholly outside the image (W % 4 = 0)
  f32x4 stq[F32IN ? 3 : 1][F32IN ? 3 : 1];
  auto load_group = [&](int p) {
    if constexpr (F32IN) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int i = k * 64 + lane, r = i / 36, v = i - r * 36;
        const int h = 4 * p + 5 + r - im.pad, c0 = 4 * (q0 - 2 + v);
        const bool ok = i < 144 && h >= 0 && h < im.H && c0 >= 0 && c0 < im.W;
        const float* ptr = im.img + (int64_t)n * im.sn + (int64_t)(ok ? h : 0) * im.sh + (ok ? c0 : 0);
#pragma unroll
        for (int c = 0; c < 3; ++c)
          stq[F32IN ? k : 0][F32IN ? c : 0] = (ok && c < im.C) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr + c * im.sc)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
  };
  auto store_group = [&](int p) {
    if constexpr (F32IN) {
      const EpiQuant iq(im.iq);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int i = k * 64 + lane, r = i / 36, v = i - r * 36;
        const uint32_t w0 = iq.code4(stq[F32IN ? k : 0][0]);
        const uint32_t w1 = im.C > 1 ? iq.code4(stq[F32IN ? k : 0][F32IN ? 1 : 0]) : 0u;
        const uint32_t w2 = im.C > 2 ? iq.code4(stq[F32IN ? k : 0][F32IN ? 2 : 0]) : 0u;
        // pixel j = bytes j of (w0, w1, w2, 0): quantize_pad_nhwc4_x4_kernel's deal
        const uint32_t a0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), a1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
        const uint32_t b0 = __builtin_amdgcn_perm(0u, w2, 0x05010400u), b1 = __builtin_amdgcn_perm(0u, w2, 0x07030602u);
        const uint32_t px[4] = {__builtin_amdgcn_perm(b0, a0, 0x05040100u), __builtin_amdgcn_perm(b0, a0, 0x07060302u),
                                __builtin_amdgcn_perm(b1, a1, 0x05040100u), __builtin_amdgcn_perm(b1, a1, 0x07060302u)};
        int8_t* const dst = ring + (((p + 1) & 3) * 4 + r) * P7_ROWB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int off = 4 * v + j - 3;                 // the pixel's place in the 136-pixel window row
          if (i < 144 && off >= 0 && off < 136) *reinterpret_cast<uint32_t*>(dst + off * 4) = px[j];
        }
      }
    }
  };
#else
  auto load_group = [&](int) {};
  auto store_group = [&](int) {};
#endif
  // fragment addresses inside a ring row: O (odd column 2 q - 1): units l31 + hsel; E (even column 2 q): 8 bytes further
  const int fo = (l31 + hsel) * 16;
  auto slot_of = [&](int R) { return ((R - 1) & 15) * P7_ROWB; };

  // one convolution row y into (aE, aO): 7 filter rows x 2 channel blocks x {E, O}
  auto conv_row = [&](int y, i32x16 (&aE)[2], i32x16 (&aO)[2]) {
    int wfl = wfo;
    asm volatile("" : "+v"(wfl));       // (the fragments are re-read per row: hoisted out of the loop they would cost 56 registers; an
                                        //  opaque OFFSET, not an opaque pointer - that would lose the LDS address space: flat loads)
    // (the sums start from the matrix instruction's inline zero - its C operand in filter row 0 - not from 64 register moves per row)
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // software pipeline of depth one, pinned: the fragments of filter row r + 1 are requested, then the four MFMAs of row r issue
    // (left to itself the scheduler hoists all 35 LDS reads of a row to the top: 140 registers)
    i32x4 o, e, w0, w1;
    auto fetch = [&](int r, i32x4& oo, i32x4& ee, i32x4& ww0, i32x4& ww1) {
      const int8_t* const row = ring + slot_of(2 * y + r);
      oo = *reinterpret_cast<const i32x4*>(row + fo);
      const int2 e0 = *reinterpret_cast<const int2*>(row + fo + 8), e1 = *reinterpret_cast<const int2*>(row + fo + 16);
      ee = i32x4{e0.x, e0.y, e1.x, e1.y};
      ww0 = *reinterpret_cast<const i32x4*>(lds + wfl + (r * 2) * 1024);
      ww1 = *reinterpret_cast<const i32x4*>(lds + wfl + (r * 2 + 1) * 1024);
    };
    fetch(0, o, e, w0, w1);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      i32x4 o2 = o, e2 = e, v0 = w0, v1 = w1;
      if (r + 1 < 7) fetch(r + 1, o2, e2, v0, v1);
      __builtin_amdgcn_sched_barrier(0);
      const i32x4 ox = XS ? i32x4{(int)(o.x ^ xorw), (int)(o.y ^ xorw), (int)(o.z ^ xorw), (int)(o.w ^ xorw)} : o;
      const i32x4 ex = XS ? i32x4{(int)(e.x ^ xorw), (int)(e.y ^ xorw), (int)(e.z ^ xorw), (int)(e.w ^ xorw)} : e;
      aE[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, ex, r == 0 ? zero16 : aE[0], 0, 0, 0);
      aO[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, ox, r == 0 ? zero16 : aO[0], 0, 0, 0);
      aE[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, ex, r == 0 ? zero16 : aE[1], 0, 0, 0);
      aO[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, ox, r == 0 ? zero16 : aO[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      o = o2; e = e2; w0 = v0; w1 = v1;
    }
    if (anyneg) {      // channels with a negative scale: v(a) falls with a - flip the bits (a -> ~a reverses the order), flip back after the pool
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int nm = reinterpret_cast<const int*>(par)[192 + j * 32 + hsel * 16 + i];
          aE[j][i] ^= nm;
          aO[j][i] ^= nm;
        }
    }
  };
  // horizontal maximum of one convolution row: h[j][i] = max(O[q], E[q], O[q + 1]) for this lane's pooled column q (the right
  // neighbour's odd column by DPP wave_shl:1 - lane 31 reads lane 32, another channel half: its pooled pixel is never stored)
  // (the first tile's lane 0 has no odd column - 2 q - 1 = -1: there O is replaced by "never wins"; a wave-uniform branch, so that
  //  the second tile does not pay the select)
  const bool noleft = xt == 0 && l31 == 0;
  auto hmax = [&](int (&h)[2][16], const i32x16 (&aE)[2], const i32x16 (&aO)[2]) {
    if (xt == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          // wave_shl:1: lane q reads lane q + 1.  (Lane 63 has no source and reads 0, lane 31 reads the other channel half: both are pooled column
          // 31 of the tile, which is never stored - so no "never wins" value has to be moved into the destination first: 64 moves per row)
          const int right = __builtin_amdgcn_mov_dpp(aO[j][i], 0x130, 0xf, 0xf, true);
          h[j][i] = max(max(noleft ? P7_NEG : aO[j][i], aE[j][i]), right);
        }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int right = __builtin_amdgcn_mov_dpp(aO[j][i], 0x130, 0xf, 0xf, true);
          h[j][i] = max(max(aO[j][i], aE[j][i]), right);
        }
    }
  };
  const EpiQuant eq(ep, ep.relu != 0);                     // code(relu(v)) = max(code(v), code(0))
  const bool plainq = epi_plain(ep);                       // unsigned bytes, no zero point: EpiQuant::code4n_plain

  // ---- prologue: rows 4 p0 - 3 .. 4 p0 + 8 (three groups), then the upper row of the first window ----
  if constexpr (F32IN) {
#pragma unroll 1
    for (int pp = p0 - 2; pp <= p0; ++pp) {
      load_group(pp);
      store_group(pp);
    }
    if (p0 + 1 < p1) load_group(p0 + 1);     // (held in registers through the first iteration: a whole iteration between a group's loads and its use)
  } else {
    issue_group(p0 - 2);
    issue_group(p0 - 1);
    issue_group(p0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  i32x16 aE[2], aO[2];
  int m[2][16], h[2][16];       // running maximum of the window / the row just convolved, already maximised along x
  if (p0 > 0) {
    conv_row(2 * p0 - 1, aE, aO);
    hmax(m, aE, aO);
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) m[j][i] = P7_NEG;      // convolution row -1 does not exist
  }

  uint8_t* const out_img = ep.codes + (int64_t)n * g.PP * g.QP * 64;
  const int nvalid = (g.QP - q0 < 31 ? g.QP - q0 : 31) * 4;        // 16-byte units of a pooled row this tile stores
#pragma unroll 1
  for (int p = p0; p < p1; ++p) {
    // group G(p) - this pooled row's new image rows, requested one row ago - has landed; the code stores of the previous
    // row (younger: two instructions, or one when the tile has <= 16 pixels) stay in flight
    if constexpr (F32IN) {
      // the group loaded an iteration ago goes into the ring (rows this iteration does not read), then the next one is requested
      if (p + 1 < p1) store_group(p + 1);
      if (p + 2 < p1) load_group(p + 2);
    } else {
      if (p > p0) {
        if (nvalid > 64) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      }
      if (p + 1 < p1) issue_group(p + 1);
    }
    conv_row(2 * p, aE, aO);
    hmax(h, aE, aO);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) m[j][i] = max(m[j][i], h[j][i]);
    conv_row(2 * p + 1, aE, aO);
    hmax(h, aE, aO);
    // ---- the pooled pixel of this lane: max over the three rows; then (a + corr) * mult + bias, ReLU folded into the quantiser ----
    int paro = 4 * (P7_RING + P7_STG);
    asm volatile("" : "+v"(paro));      // (constants re-read per pooled row, not kept in 96 registers)
    const int8_t* const parl = lds + paro;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int cb = j * 32 + hsel * 16;
      f32x4 y[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(parl + (cb + 4 * qd) * 4);
        const i32x4 co = *reinterpret_cast<const i32x4*>(parl + (64 + cb + 4 * qd) * 4);
        const f32x4 bs = *reinterpret_cast<const f32x4*>(parl + (128 + cb + 4 * qd) * 4);
        int a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] = max(m[j][4 * qd + e], h[j][4 * qd + e]);
          m[j][4 * qd + e] = h[j][4 * qd + e];             // the lower row of this window is the upper row of the next
        }
        if (anyneg) {
          const i32x4 nm = *reinterpret_cast<const i32x4*>(parl + (192 + cb + 4 * qd) * 4);
          a[0] ^= nm.x; a[1] ^= nm.y; a[2] ^= nm.z; a[3] ^= nm.w;
        }
        // (the first-layer kernels multiply and add separately - conv_stem_i8.hip - and so does this one: same bits)
        y[qd] = f32x4{(float)(a[0] + co.x) * mu.x + bs.x, (float)(a[1] + co.y) * mu.y + bs.y, (float)(a[2] + co.z) * mu.z + bs.z,
                      (float)(a[3] + co.w) * mu.w + bs.w};
      }
      uint32_t wq[4];
      if (plainq) {
        eq.code4n_plain(y, wq);
      } else {
        bool uq[4];
        eq.code4n(y, wq, uq);
      }
      // 16-byte unit u = 2 j + hsel of pixel l31 goes to slot u ^ ((l31 >> 2) & 3) of its 64-byte row (conflict-free both ways)
      *reinterpret_cast<i32x4*>(stg + l31 * 64 + (((2 * j + hsel) ^ ((l31 >> 2) & 3)) << 4)) = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
    }
    // ---- store: pooled pixels q0 .. q0 + 30 of row p are 31 x 64 contiguous bytes ----
    uint8_t* const dst = out_img + ((int64_t)p * g.QP + q0) * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = k * 64 + lane, q = t >> 2, u = t & 3;
      const i32x4 c16 = *reinterpret_cast<const i32x4*>(stg + q * 64 + ((u ^ ((q >> 2) & 3)) << 4));
      if (t < nvalid) __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(dst + t * 16));
    }
  }
}

// Whether the in-register pooling kernel takes this first layer, and the launch (conv_stem_i8.hip asks).
bool stem_pool7_applies(int64_t Hp, int64_t Wp, int64_t K, int64_t R, int64_t S, int32_t stride, const float* out, const void* codes) {
  if (R != 7 || S > 8 || stride != 2 || K != 64 || out || !codes) return false;
  const int64_t P = (Hp - R) / stride + 1, Q = (Wp - S) / stride + 1;
  if ((P & 1) || (Q & 1) || P < 2 || Q < 2) return false;            // every pooled pixel's window ends inside the convolution output
  return true;
}

int stem_pool7_launch(const uint8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                      const float* in_zero_point, const float* w_scale, int64_t N, int64_t Hp, int64_t Wp, int64_t S, int shift,
                      const ConvEpi& ep, hipStream_t st) {
  Pool7Geom g;
  g.N = (int)N; g.Hp = (int)Hp; g.Wp = (int)Wp;
  g.P = (int)((Hp - 7) / 2 + 1);
  g.Q = (int)((Wp - S) / 2 + 1);
  g.PP = g.P / 2;                                                    // MaxPool2d(3, 2, 1) on an even size
  g.QP = g.Q / 2;
  g.xtiles = (g.QP + 30) / 31;
  g.rows_per_band = 14;
  g.bands = (g.PP + g.rows_per_band - 1) / g.rows_per_band;
  const int64_t tasks = N * g.bands * g.xtiles;
  if (tasks >= (1ll << 31) || N * Hp * Wp * 4 >= (1ll << 40)) return DLMCQ_ERANGE;
  g.ntasks = (uint32_t)tasks;
  g.xtdiv = make_fastdiv((uint32_t)g.xtiles);
  g.bdiv = make_fastdiv((uint32_t)g.bands);
  g.xbytes = N * Hp * Wp * 4;
#ifdef DLMCQ_LAB
#define P7_LAB_ARG , Pool7Img{}
#else
#define P7_LAB_ARG
#endif
  if (shift) hipLaunchKernelGGL(conv_stem_pool7_i8_kernel<true>, dim3((uint32_t)((tasks + 3) / 4)), dim3(256), 0, st, x, w, bias, wsum, in_scale,
                                in_zero_point, w_scale, g, shift, ep P7_LAB_ARG);
  else hipLaunchKernelGGL(conv_stem_pool7_i8_kernel<false>, dim3((uint32_t)((tasks + 3) / 4)), dim3(256), 0, st, x, w, bias, wsum, in_scale,
                          in_zero_point, w_scale, g, shift, ep P7_LAB_ARG);
  return launch_status();
}

#ifdef DLMCQ_LAB
// lab library only: the first layer in ONE launch - image fp32 [N][C][H][W] (strides in elements) -> pooled codes [N][PP][QP][64]
int stem_pool7_f32_launch(const float* img, int64_t sn, int64_t sc, int64_t sh, int C, int H, int W, int pad, const ConvEpi& iq, const int8_t* w,
                          const float* bias, const int32_t* wsum, const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                          int64_t S, const ConvEpi& ep, hipStream_t st) {
  const int64_t Hp = H + 2 * pad, Wp = W + 2 * pad;
  Pool7Geom g;
  g.N = (int)N; g.Hp = (int)Hp; g.Wp = (int)Wp;
  g.P = (int)((Hp - 7) / 2 + 1);
  g.Q = (int)((Wp - S) / 2 + 1);
  g.PP = g.P / 2;
  g.QP = g.Q / 2;
  g.xtiles = (g.QP + 30) / 31;
  g.rows_per_band = 14;
  g.bands = (g.PP + g.rows_per_band - 1) / g.rows_per_band;
  const int64_t tasks = N * g.bands * g.xtiles;
  if (tasks >= (1ll << 31) || (W & 3) || pad != 3 || C > 3) return DLMCQ_EINVAL;
  g.ntasks = (uint32_t)tasks;
  g.xtdiv = make_fastdiv((uint32_t)g.xtiles);
  g.bdiv = make_fastdiv((uint32_t)g.bands);
  g.xbytes = 0;
  Pool7Img im{img, sn, sc, sh, H, W, pad, C, iq};
  hipLaunchKernelGGL((conv_stem_pool7_i8_kernel<false, true>), dim3((uint32_t)((tasks + 3) / 4)), dim3(256), 0, st, nullptr, w, bias, wsum, in_scale,
                     in_zero_point, w_scale, g, 0, ep, im);
  return launch_status();
}
#endif

}  // namespace dlmcq

#ifdef DLMCQ_LAB
// lab library only (tools/stem_fused_lab.py): image quantiser + 7x7 / 2 convolution + ReLU + MaxPool2d(3, 2, 1) + the consumer's quantiser as
// ONE launch.  `a_*`: the image quantiser (dlmcq_quantize_pad_nhwc4's arguments; DLMCQ_EMIT_SHIFT128 expected for an unsigned range - the
// kernel multiplies `code - 128`, `in_zero_point` is then `zp - 128`); the rest: dlmcq_conv2d_i8_stem_pool's.
extern "C" int dlmcq_x_stem_pool7_f32(const float* img, int64_t N, int64_t C, int64_t H, int64_t W, int64_t stride_n, int64_t stride_c,
                                      int64_t stride_h, int32_t pad, const float* a_scale, const float* a_zero_point, int32_t a_lo, int32_t a_hi,
                                      int32_t a_form, float a_g, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                                      const float* in_zero_point, const float* w_scale, int64_t S, int32_t relu, void* codes, const float* q_scale,
                                      const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g, dlmcq_stream_t stream) {
  using namespace dlmcq;
  if (!img || !w || !wsum || !in_scale || !w_scale || !codes || !a_scale || !q_scale || N < 1 || C < 1 || C > 3 || S > 8) return DLMCQ_EINVAL;
  if (!aligned16(img) || (stride_n & 3) || (stride_c & 3) || (stride_h & 3) || !aligned16(codes) || !aligned16(w)) return DLMCQ_EALIGN;
  ConvEpi iq{};
  if (!epi_set_form(iq, a_form, a_lo, a_hi)) return DLMCQ_EINVAL;
  iq.codes = reinterpret_cast<uint8_t*>(uintptr_t(1));      // (EpiQuant resolves its quantiser only for a non-null `codes`)
  iq.q_scale = a_scale; iq.q_zp = a_zero_point; iq.q_lo = (float)a_lo; iq.q_hi = (float)a_hi; iq.q_g = a_g;
  ConvEpi ep{};
  ep.relu = relu != 0; ep.codes = static_cast<uint8_t*>(codes); ep.q_scale = q_scale; ep.q_zp = q_zero_point;
  ep.q_lo = (float)q_lo; ep.q_hi = (float)q_hi; ep.q_g = q_ste_g; ep.q_form = q_form;
  return stem_pool7_f32_launch(img, stride_n, stride_c, stride_h, (int)C, (int)H, (int)W, pad, iq, w, bias, wsum, in_scale, in_zero_point, w_scale, N, S,
                               ep, reinterpret_cast<hipStream_t>(stream));
}
#endif
