// Fused int8-dequant x GEMM convolution / linear on the gfx950 matrix cores (the quantised
// conv/linear of modules/conv.py:13-19, modules/linear.py:12-13 where it really is a dense contraction).
//
// Reference semantics:  out = conv(x', w') + bias  with  x' = (q - zp) * s_in   (FSPTQuant/base.py:108-109)
//                                                        w' = qw * s_w[k]       (FSPTQuant/base.py:149-152)
// computed here as      out = s_in * s_w[k] * ( SUM q'*qw  +  (shift - zp) * SUM qw )  + bias[k]
// with q' = q - shift the int8 operand (shift = 128 for uint8 codes, 0 for int8 codes), exact int32
// accumulation on v_mfma_i32_32x32x32_i8, ONE rounding chain at the end.  Padded taps contribute x' = 0,
// i.e. q = zp, so out-of-bounds operand bytes are filled with (zp - shift) and SUM qw runs over all taps.
//
// Layouts (chosen for the matrix cores; torch sees them as channels_last tensors, no copy):
//   activations  int8  NHWC   - for a fixed tap the 64 reduction bytes of a BK step are contiguous
//   weights      int8  KRSC   - same reduction order (r, s, c), produced by quantize_weight_krsc_kernel
//   output       fp32  NHWC   - lanes of an accumulator register hold 32 consecutive channels: 128-B stores
// Implicit GEMM: M = N*P*Q output pixels, N = K output channels, K = R*S*C.  At ResNet sizes with fp32 outputs
// this kernel is HBM-bound (4 B written per MAC-row vs 1 B read), so the structure favours streaming: BM = 128
// pixels x BN in {64, 128} channels per workgroup, 4 waves (one 32-row slab each), BK = 64, double-buffered LDS
// with register staging (one barrier per K step), rows padded to 80 B so ds_read_b128 fragments are conflict-free.
#include <cstdlib>
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

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_i8_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                     float* __restrict__ out, const float* __restrict__ bias,
                                                     const int32_t* __restrict__ wsum, const float* __restrict__ s_in,
                                                     const float* __restrict__ zp_in, const float* __restrict__ s_w,
                                                     ConvGeom g, int shift) {
  __shared__ __attribute__((aligned(16))) int8_t lds[2 * (CV_BM + BN) * CV_LD];
  int8_t* ldsA = lds;
  int8_t* ldsB = lds + 2 * CV_BM * CV_LD;
  constexpr int NT = BN / 32;          // 32x32 output tiles per wave along N
  constexpr int BLOADS = BN / 64;      // 16-B weight loads per thread per K step

  // XCD-aware tile order: the workgroups that share an activation tile (same m-block, different n-blocks)
  // are consecutive in `tile`, and consecutive tiles are dealt to the SAME XCD (its L2 then serves the re-reads).
  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t qd = nwg >> 3, rm = nwg & 7u;
  const uint32_t tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
  const int bn = tile % g.nblk_n, bm = tile / g.nblk_n;
  const int64_t m0 = (int64_t)bm * CV_BM;
  const int n0 = bn * BN;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const uint32_t padw = (uint32_t)(uint8_t)(int8_t)(zpi - shift) * 0x01010101u;
  const uint32_t xorw = shift ? 0x80808080u : 0u;   // uint8 code -> int8 operand: q - 128 == q ^ 0x80

  // ---- staging assignment: thread -> (row, 16-byte segment) ----
  const int seg = tid & 3, srow = tid >> 2;       // rows srow and srow + 64 of the A tile
  int a_n[2], a_h0[2], a_w0[2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + srow + 64 * i;
    a_ok[i] = m < g.M;
    row_origin(g, a_ok[i] ? (uint32_t)m : 0u, a_n[i], a_h0[i], a_w0[i]);
  }
  const int cchunks = g.C / CV_BK;
  const int nsteps = g.R * g.S * cchunks;
  const int64_t wrow = (int64_t)g.R * g.S * g.C;  // bytes per output channel in KRSC

  // Register staging, TWO steps ahead (two register sets; the loop is unrolled by two so that their roles are
  // static); (r, s, c-chunk) advance incrementally - no integer division in the loop.
  i32x4 ra0[2], ra1[2], rb0[BLOADS], rb1[BLOADS];
  int f_cc = 0, f_s = 0, f_r = 0;   // tap / channel chunk of the NEXT fetch
  auto fetch = [&](i32x4* pa, i32x4* pb) {
    const int c0 = f_cc * CV_BK + seg * 16;
    const int rs = f_r * g.S + f_s;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int h = a_h0[i] + f_r * g.dil, ww = a_w0[i] + f_s * g.dil;
      if (a_ok[i] && h >= 0 && h < g.H && ww >= 0 && ww < g.W) {
        const int64_t off = (((int64_t)a_n[i] * g.H + h) * g.W + ww) * g.C + c0;
        const i32x4 v = *reinterpret_cast<const i32x4*>(x + off);
        pa[i] = i32x4{(int)(v.x ^ xorw), (int)(v.y ^ xorw), (int)(v.z ^ xorw), (int)(v.w ^ xorw)};
      } else {
        pa[i] = i32x4{(int)padw, (int)padw, (int)padw, (int)padw};
      }
    }
#pragma unroll
    for (int i = 0; i < BLOADS; ++i) {
      const int k = n0 + srow + 64 * i;
      if (k < g.K)
        pb[i] = *reinterpret_cast<const i32x4*>(w + (int64_t)k * wrow + (int64_t)rs * g.C + c0);
      else
        pb[i] = i32x4{0, 0, 0, 0};
    }
    if (++f_cc == cchunks) {
      f_cc = 0;
      if (++f_s == g.S) {
        f_s = 0;
        ++f_r;
      }
    }
  };
  auto stage = [&](int buf, const i32x4* pa, const i32x4* pb) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<i32x4*>(ldsA + (buf * CV_BM + srow + 64 * i) * CV_LD + seg * 16) = pa[i];
#pragma unroll
    for (int i = 0; i < BLOADS; ++i)
      *reinterpret_cast<i32x4*>(ldsB + (buf * BN + srow + 64 * i) * CV_LD + seg * 16) = pb[i];
  };

  i32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0;

  const int arow = wave * 32 + (lane & 31), kq = (lane >> 5) * 16;
  auto multiply = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < CV_BK / 32; ++ks) {
      const i32x4 af = *reinterpret_cast<const i32x4*>(ldsA + (buf * CV_BM + arow) * CV_LD + ks * 32 + kq);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const i32x4 bf = *reinterpret_cast<const i32x4*>(ldsB + (buf * BN + j * 32 + (lane & 31)) * CV_LD + ks * 32 + kq);
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[j], 0, 0, 0);
      }
    }
  };

  fetch(ra0, rb0);                       // step 0
  if (nsteps > 1) fetch(ra1, rb1);       // step 1
  stage(0, ra0, rb0);
  __syncthreads();
  const int npairs = (nsteps + 1) / 2;
  for (int pr = 0; pr < npairs; ++pr) {
    const int step = 2 * pr;
    // even step: LDS buffer 0 holds `step`, set 1 holds step+1, set 0 is free for step+2
    if (step + 2 < nsteps) fetch(ra0, rb0);
    multiply(0);
    if (step + 1 < nsteps) stage(1, ra1, rb1);
    __syncthreads();
    // odd step (skipped as a whole when nsteps is odd and this is the last pair)
    if (step + 1 < nsteps) {
      if (step + 3 < nsteps) fetch(ra1, rb1);
      multiply(1);
      if (step + 2 < nsteps) stage(0, ra0, rb0);
    }
    __syncthreads();
  }

  // ---- epilogue: one rounding chain; lanes 0-31 of a register hold 32 consecutive output channels ----
  const float sin = s_in[0];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 32 + (lane & 31);
    if (col >= g.K) continue;
    const float mult = sin * s_w[col];
    const int corr = (shift - zpi) * wsum[col];
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t row = m0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (row < g.M) __builtin_nontemporal_store((float)(acc[j][i] + corr) * mult + bv, out + row * g.K + col);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// LDS-DMA variant: operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging), three LDS
// buffers, two K steps in flight, ONE raw s_barrier per step with a counted s_waitcnt vmcnt.
//   * a wave-instruction lands 64 lanes x 16 B = 16 rows x 64 B contiguously, so rows cannot be padded; bank
//     conflicts are avoided by an XOR swizzle applied on the SOURCE side: LDS slot p of row r holds the logical
//     16-byte segment p ^ ((r >> 2) & 3), and the fragment reads apply the same involution;
//   * DMA cannot transform or synthesise bytes: the uint8 -> int8 shift (q ^ 0x80) is applied to the A fragment
//     after the ds_read, and padded taps / rows beyond K read a 16-byte line of a constant table instead.
struct PadTable {   // 64 bytes of every byte value: a padded tap reads its K chunks at offsets 0 / 32 of one line
  int8_t b[256 * 64];
  constexpr PadTable() : b() {
    for (int v = 0; v < 256; ++v)
      for (int j = 0; j < 64; ++j) b[v * 64 + j] = (int8_t)v;
  }
};
__device__ const PadTable g_pad_table = PadTable();

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ABL (timing-only ablations, never shipped): 1 = no operand DMA inside the K loop, 2 = no MFMA, 4 = no LDS fragment reads
// A second (input, weight) pair accumulated into the same output tile - the shortcut convolution of a residual
// block's first unit, so that  conv3(x) + downsample(y)  is one kernel and neither addend travels through HBM.
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

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ADIR: the A operand (activations) bypasses LDS.  A wave multiplies only its own 32 rows, so staging A in LDS buys no
// reuse - it only halves the LDS ring's depth and doubles its DMA traffic.  With ADIR each lane loads its own
// fragment bytes (row = lane & 31, 16 bytes of the K step) straight into registers, NBUF - 1 steps ahead; the ring
// holds the shared B operand (weights) only.
template <int BM, int BN, int NBUF = 3, int BK = CV_BK, int ABL = 0, bool DUAL = false, bool ADIR = false>
__global__ __launch_bounds__(256, (BM * BN >= 256 * 256 ? 1 : (BM == 256 || BN == 256 || DUAL ? 2 : 3))) void conv_i8_dma_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                         float* __restrict__ out, const float* __restrict__ bias,
                                                         const int32_t* __restrict__ wsum,
                                                         const float* __restrict__ s_in,
                                                         const float* __restrict__ zp_in,
                                                         const float* __restrict__ s_w, ConvGeom g, int shift,
                                                         ConvEpi ep, ConvSeg2 sg) {
  constexpr int PF = NBUF - 1;  // K steps in flight
  constexpr int TILE_A = ADIR ? 0 : BM * BK, TILE_B = BN * BK, TILE = TILE_A + TILE_B;
  constexpr int KS = BK / 32;           // MFMA K chunks per step
  constexpr int MT = BM / 128;  // 32-row slabs per wave along M (a wave owns BM/4 consecutive rows)
  constexpr int NT = BN / 32;
  constexpr int SLOTS = BK / 16;        // 16-byte slots per LDS row (one row = BK bytes = one full 64/128-B line)
  constexpr int RPI = 64 / SLOTS;       // tile rows covered by one wave-instruction (1 KiB)
  constexpr int RPB = 256 / BK;         // tile rows per 256-byte LDS bank row: the swizzle key is row / RPB
  constexpr int AI = BM / (RPI * 4);    // A wave-instructions per wave per step
  constexpr int BI = BN / (RPI * 4);    // B wave-instructions per wave per step
  constexpr int EP_BYTES = 4 * 32 * 68 * 4;   // the epilogue stage (MT == 1): 4 waves x 32 rows x 68 floats
  constexpr int LDS_BYTES = (ADIR && NBUF * TILE < EP_BYTES) ? EP_BYTES : NBUF * TILE;
  __shared__ __attribute__((aligned(1024))) int8_t lds[LDS_BYTES];

  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t qd = nwg >> 3, rm = nwg & 7u;
  const uint32_t tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
  const int bn = tile % g.nblk_n, bm = tile / g.nblk_n;
  const int64_t m0 = (int64_t)bm * BM;
  const int n0 = bn * BN;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wrow0 = wave * (BM / 4);            // first tile row of this wave
  const int hsel = lane >> 5;
  i32x16 acc[MT][NT];

  // ---- DMA assignment: wave-instruction i of this wave covers tile rows (i*4 + wave)*16 .. +15 ----
  const int lrow = lane / SLOTS, pslot = lane % SLOTS;
  int a_seg[AI], b_seg[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) a_seg[i] = pslot ^ ((((i * 4 + wave) * RPI + lrow) / RPB) & (SLOTS - 1));
#pragma unroll
  for (int i = 0; i < BI; ++i) b_seg[i] = pslot ^ ((((i * 4 + wave) * RPI + lrow) / RPB) & (SLOTS - 1));

  // One (input, weight) pair as a source of K steps.  DUAL kernels have two; their steps form ONE sequence through the
  // same LDS ring (the second pair's first steps are already in flight while the first pair's last steps multiply).
  constexpr int NA = ADIR ? MT : AI;   // A rows this lane addresses: its own fragment rows (ADIR) or its DMA rows
  struct Feed {
    const int8_t* x;
    const int8_t* padline;     // stored UNshifted: the xor happens on read
    const int8_t* b_src[BI];
    int a_n[NA], a_h0[NA], a_w0[NA];
    bool a_ok[NA];
    int cc, s, r, cchunks, nsteps;
    uint32_t xorw;
    // ADIR: running pointers, so that a K step costs two 64-bit adds per operand row instead of the whole
    // (bounds check, pixel address, tap offset) computation - that arithmetic, not memory, was what bounded the loop
    const int8_t* ap[NA];   // this lane's A bytes for the current tap and channel chunk (or the pad line)
    int a_inc[NA];          // BK for a real pixel, 0 for a padded tap
    const int8_t* bp[BI];   // this lane's B source for the current step
    int b_inc[BI];
  };
  auto retap = [&](Feed& f, const ConvGeom& gg) {   // ADIR: A pointers of tap (f.r, f.s), channel chunk 0
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int h = f.a_h0[i] + f.r * gg.dil, ww = f.a_w0[i] + f.s * gg.dil;
      const bool in = f.a_ok[i] && h >= 0 && h < gg.H && ww >= 0 && ww < gg.W;
      f.ap[i] = in ? f.x + (((int64_t)f.a_n[i] * gg.H + h) * gg.W + ww) * gg.C + (ADIR ? hsel : a_seg[ADIR ? 0 : i]) * 16 : f.padline;
      f.a_inc[i] = in ? BK : 0;
    }
  };
  auto make_feed = [&](Feed& f, const int8_t* __restrict__ xx, const int8_t* __restrict__ ww, const ConvGeom& gg, int zpi,
                       int shf) {
    f.x = xx;
    f.padline = g_pad_table.b + ((zpi & 0xff) << 6);
    f.xorw = shf ? 0x80808080u : 0u;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int64_t m = m0 + (ADIR ? wrow0 + i * 32 + (lane & 31) : (i * 4 + wave) * RPI + lrow);
      f.a_ok[i] = m < gg.M;
      row_origin(gg, f.a_ok[i] ? (uint32_t)m : 0u, f.a_n[i], f.a_h0[i], f.a_w0[i]);
    }
    const int64_t wrow = (int64_t)gg.R * gg.S * gg.C;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int k = n0 + (i * 4 + wave) * RPI + lrow;
      f.b_src[i] = k < gg.K ? ww + (int64_t)k * wrow : nullptr;
    }
    f.cc = f.s = f.r = 0;
    f.cchunks = gg.C / BK;
    f.nsteps = gg.R * gg.S * f.cchunks;
    if (ADIR || BM * BN >= 256 * 256) {
#pragma unroll
      for (int i = 0; i < BI; ++i) {     // KRSC: the reduction index is contiguous, a step is BK bytes further
        f.bp[i] = f.b_src[i] ? f.b_src[i] + b_seg[i] * 16 : g_pad_table.b;
        f.b_inc[i] = f.b_src[i] ? BK : 0;
      }
      retap(f, gg);
    }
  };
  i32x4 areg[ADIR ? NBUF : 1][ADIR ? MT : 1][ADIR ? KS : 1];   // ADIR: the A fragments of the steps in flight
  auto issue = [&](Feed& f, const ConvGeom& gg, auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
    int8_t* base = lds + SL * TILE;
    if (ADIR) {
      // B first: its DMA lands in LDS and is awaited by the whole workgroup; the A registers are private
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        if (!(ABL & 16)) __builtin_amdgcn_global_load_lds((gptr_t)f.bp[i], (lptr_t)(base + (i * 4 + wave) * 1024), 16, 0, 0);
        f.bp[i] += f.b_inc[i];
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        // explicit global-address-space loads (a generic pointer would become flat_load, which also counts on lgkmcnt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          areg[ADIR ? SL : 0][ADIR ? i : 0][ADIR ? ks : 0] =
              *reinterpret_cast<const __attribute__((address_space(1))) i32x4*>((gptr_t)(f.ap[i] + ks * 32));
        f.ap[i] += f.a_inc[i];
      }
      if (++f.cc == f.cchunks) {
        f.cc = 0;
        if (++f.s == gg.S) {
          f.s = 0;
          ++f.r;
        }
        retap(f, gg);     // (past the last tap nothing is issued any more; the pointers are simply not used)
      }
      return;
    }
    const int rs = f.r * gg.S + f.s;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int8_t* src = f.b_src[i] ? f.b_src[i] + (int64_t)rs * gg.C + f.cc * BK + b_seg[i] * 16 : g_pad_table.b;
      if (!(ABL & 16)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + TILE_A + (i * 4 + wave) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int h = f.a_h0[i] + f.r * gg.dil, ww = f.a_w0[i] + f.s * gg.dil;
      const bool in = f.a_ok[i] && h >= 0 && h < gg.H && ww >= 0 && ww < gg.W;
      const int8_t* src = in ? f.x + (((int64_t)f.a_n[i] * gg.H + h) * gg.W + ww) * gg.C + f.cc * BK + a_seg[i] * 16 : f.padline;
      if (!(ABL & 8)) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (i * 4 + wave) * 1024), 16, 0, 0);
    }
    constexpr bool BIG = BM * BN >= 256 * 256;   // keeps the running pointers of the pinned steady state in step
    if (BIG) {
#pragma unroll
      for (int i = 0; i < BI; ++i) f.bp[i] += f.b_inc[i];
#pragma unroll
      for (int i = 0; i < NA; ++i) f.ap[i] += f.a_inc[i];
    }
    if (++f.cc == f.cchunks) {
      f.cc = 0;
      if (++f.s == gg.S) {
        f.s = 0;
        ++f.r;
      }
      if (BIG) retap(f, gg);
    }
  };

  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  Feed fm;                                   // the layer's own pair
  make_feed(fm, x, w, g, zpi, shift);
  Feed fs;                                   // DUAL: the shortcut pair, reduced FIRST (its sum waits in registers)
  int zpi2 = 0;
  if (DUAL) {
    const float zf2 = sg.zp_in ? sg.zp_in[0] : 0.0f;
    zpi2 = (int)__builtin_rintf(zf2);
    make_feed(fs, sg.x, sg.w, sg.g, zpi2, sg.shift);
  }
  const int nfirst = DUAL ? fs.nsteps : 0;
  const int nsteps = nfirst + fm.nsteps;
  int issued = 0;
  auto issue_next = [&](auto slot_c) {
    if (DUAL && issued < nfirst) issue(fs, sg.g, slot_c);
    else issue(fm, g, slot_c);
    ++issued;
  };

#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][j][i] = 0;
  float extra[DUAL ? MT : 1][DUAL ? NT : 1][16];

  static_for<PF>([&](auto i) {
    if (decltype(i)::value < nsteps) issue_next(i);
  });
  constexpr int GROUP = (ADIR ? MT * KS : AI) + BI;   // vector-memory instructions per step per wave
  // ABL & 64 (timing study, never shipped): lane 0 of wave 0 of the middle workgroup stamps the shader clock at the phase
  // boundaries of its first 16 steps into the buffer passed as ep.residual
  unsigned long long* trace = nullptr;
  if ((ABL & 64) && blockIdx.x == gridDim.x / 2 && tid == 0) trace = (unsigned long long*)ep.residual;
  auto stamp = [&](int step, int k) {
    if ((ABL & 64) && trace && step < 16) trace[step * 8 + k] = __builtin_readcyclecounter();
  };
  auto one_step = [&](int step, auto slot_c) {
    constexpr int U = decltype(slot_c)::value;          // ring slot of this step; step + PF goes to slot (U + PF) % NBUF
    stamp(step, 0);
    // step's own loads must have landed; the younger groups stay in flight
    if (ABL & (1 | 8 | 16 | 32)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (step + PF - 1 < nsteps) {          // PF-1 younger groups stay in flight
      constexpr int KEEP = (PF - 1) * GROUP;
      if (KEEP == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (KEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (KEEP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (KEEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (KEEP == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if (KEEP == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (KEEP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail: drain (conservative)
    }
    stamp(step, 1);
    __builtin_amdgcn_s_barrier();                 // everyone's step-k bytes are in LDS; everyone left multiply(k-1)
    stamp(step, 2);
    constexpr bool LATE = BM * BN >= 256 * 256 && !(ABL & 64);   // one wave per SIMD: issue the loads BEHIND the MFMAs (they execute meanwhile)
    if (!LATE && !(ABL & 1) && step + PF < nsteps) issue_next(std::integral_constant<int, (U + PF) % NBUF>{});  // the slot multiply(k-1) released
    stamp(step, 3);
    if (DUAL && step == nfirst) {
      // the shortcut pair is complete: dequantise its sum into registers and start the layer's own sum from zero
      const float sin2 = sg.s_in[0];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
        const bool cok = col < g.K;
        const float mult = cok ? sin2 * sg.s_w[col] : 0.0f;
        const int corr = cok ? (sg.shift - zpi2) * sg.wsum[col] : 0;
        const float bv = (cok && sg.bias) ? sg.bias[col] : 0.0f;
#pragma unroll
        for (int mi = 0; mi < (DUAL ? MT : 1); ++mi)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            extra[mi][DUAL ? j : 0][i] = (float)(acc[mi][j][i] + corr) * mult + bv;
            acc[mi][j][i] = 0;
          }
      }
    }
    const uint32_t xorw = (DUAL && step < nfirst) ? fs.xorw : fm.xorw;
    const int8_t* base = lds + U * TILE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int sg_ = ks * 2 + hsel;
      i32x4 af[MT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int arow = wrow0 + mi * 32 + (lane & 31);
        i32x4 t;
        if (ADIR) t = areg[ADIR ? U : 0][ADIR ? mi : 0][ADIR ? ks : 0];
        else t = (ABL & 4) ? i32x4{lane, step, ks, mi}
                           : *reinterpret_cast<const i32x4*>(base + arow * BK + ((sg_ ^ ((arow / RPB) & (SLOTS - 1))) << 4));
        af[mi] = i32x4{(int)(t.x ^ xorw), (int)(t.y ^ xorw), (int)(t.z ^ xorw), (int)(t.w ^ xorw)};
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int brow = j * 32 + (lane & 31);
        const i32x4 bf = (ABL & 4) ? i32x4{j, lane, step, ks}
                                   : *reinterpret_cast<const i32x4*>(base + TILE_A + brow * BK + ((sg_ ^ ((brow / RPB) & (SLOTS - 1))) << 4));
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          if (ABL & 2) {
            asm volatile("" ::"v"(af[mi]), "v"(bf));      // keep the fragments live without multiplying
            acc[mi][j][0] += bf.x;
          } else {
            acc[mi][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[mi], bf, acc[mi][j], 0, 0, 0);
          }
        }
      }
    }
    if (LATE) {
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 1) && step + PF < nsteps) issue_next(std::integral_constant<int, (U + PF) % NBUF>{});
    }
  };
  // Largest tile (one wave per SIMD): steady-state steps as ONE scheduling region each, with the instruction stream
  // pinned so that the operand loads and the second K chunk's fragment reads sit in the shadow of the MFMAs (a wave
  // that is stuck issuing a load cannot issue MFMAs: back to back the 8 loads cost 184 clocks each, spaced out ~60)
  int s_first = 0;
  if constexpr (BM * BN >= 256 * 256 && !ADIR && !DUAL && !(ABL & ~64) && MT == 2 && NT == 8 && KS == 2) {
    // one operand load of the step being fetched (pieces 0..BI-1: weights, BI..BI+AI-1: activations)
    auto issue_piece = [&](Feed& f, const ConvGeom& gg, auto slot_c, auto piece_c) {
      constexpr int SL = decltype(slot_c)::value, PC = decltype(piece_c)::value;
      int8_t* base = lds + SL * TILE;
      if constexpr (PC < BI) {
        __builtin_amdgcn_global_load_lds((gptr_t)f.bp[PC], (lptr_t)(base + TILE_A + (PC * 4 + wave) * 1024), 16, 0, 0);
        f.bp[PC] += f.b_inc[PC];
      } else {
        constexpr int i = PC - BI;
        __builtin_amdgcn_global_load_lds((gptr_t)f.ap[i], (lptr_t)(base + (i * 4 + wave) * 1024), 16, 0, 0);
        f.ap[i] += f.a_inc[i];
      }
    };
    auto steady = [&](auto slot_c) {
      constexpr int U = decltype(slot_c)::value;
      constexpr int KEEP = (PF - 1) * GROUP;
      static_assert(KEEP == 8 && AI + BI == 8, "vmcnt immediate / piece count");
      const int tstep = issued - PF;
      stamp(tstep, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      stamp(tstep, 1);
      __builtin_amdgcn_s_barrier();
      stamp(tstep, 2);
      const int8_t* base = lds + U * TILE;
      auto afrag = [&](int ks, int mi) {
        const int arow = wrow0 + mi * 32 + (lane & 31);
        const i32x4 t = *reinterpret_cast<const i32x4*>(base + arow * BK + (((ks * 2 + hsel) ^ ((arow / RPB) & (SLOTS - 1))) << 4));
        return i32x4{(int)(t.x ^ fm.xorw), (int)(t.y ^ fm.xorw), (int)(t.z ^ fm.xorw), (int)(t.w ^ fm.xorw)};
      };
      auto bfrag = [&](int ks, int j) {
        const int brow = j * 32 + (lane & 31);
        return *reinterpret_cast<const i32x4*>(base + TILE_A + brow * BK + (((ks * 2 + hsel) ^ ((brow / RPB) & (SLOTS - 1))) << 4));
      };
      i32x4 a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) a0[mi] = afrag(0, mi);
#pragma unroll
      for (int j = 0; j < NT; ++j) b0[j] = bfrag(0, j);
      __builtin_amdgcn_sched_barrier(0);
      // 8 groups: two MFMAs of the first K chunk, ONE operand load of step k+2, one or two fragment reads of the second
      static_for<8>([&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0[mi], b0[i], acc[mi][i], 0, 0, 0);
        issue_piece(fm, g, std::integral_constant<int, (U + PF) % NBUF>{}, i_c);
        if constexpr (i == 0) {
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) a1[mi] = afrag(1, mi);
        }
        b1[i] = bfrag(1, i);
        __builtin_amdgcn_sched_barrier(0);
      });
      stamp(tstep, 3);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1[mi], b1[j], acc[mi][j], 0, 0, 0);
      if (ABL & 64) {
        __builtin_amdgcn_sched_barrier(0);
        stamp(tstep, 4);
      }
      ++issued;
      if (++fm.cc == fm.cchunks) {
        fm.cc = 0;
        if (++fm.s == g.S) {
          fm.s = 0;
          ++fm.r;
        }
        retap(fm, g);
      }
    };
    for (; s_first + NBUF + PF <= nsteps; s_first += NBUF) static_for<NBUF>(steady);
  }
  for (int s0 = s_first; s0 < nsteps; s0 += NBUF)
    static_for<NBUF>([&](auto u) {
      if (s0 + decltype(u)::value < nsteps) {
        one_step(s0 + decltype(u)::value, u);
        if (ABL & 64) {
          asm volatile("s_nop 0" ::: "memory");
          stamp(s0 + decltype(u)::value, 4);
        }
      }
    });
  if ((ABL & 64)) { ep.residual = nullptr; }

  const float sin = s_in[0];
  const EpiQuant eq(ep);
  constexpr bool EP_FITS = EP_BYTES <= LDS_BYTES;   // the LDS epilogue stage re-uses the operand buffers
  if (EP_FITS && (g.K & 3) == 0) {
    // ---- epilogue through LDS: accumulator layout (lane = channel, register = row) -> row-major, so that each
    // lane stores 16 B and each wave-instruction writes 4 rows x 256 contiguous bytes: 16 dwordx4 stores per lane
    // instead of 64 dword stores (the 1x1 layers are bound by this output stream).  Two passes of 64 channels. ----
    constexpr int EP_LD = 68;                       // floats per staged row (64 + 4 pad)
    __builtin_amdgcn_s_barrier();                   // every wave is done reading the operand buffers
    float* stg = reinterpret_cast<float*>(lds) + wave * (32 * EP_LD);
    const int er = lane >> 4, ec = (lane & 15) * 4;
    constexpr int NH = NT / 2 + (NT & 1);
    static_for<MT * NH>([&](auto pass_c) {    // (compile-time indices: the accumulators must stay in registers)
      constexpr int mi = decltype(pass_c)::value / NH, h = decltype(pass_c)::value % NH;
      // the shortcut tile is requested first: its latency hides behind the dequantise-and-stage phase below
      f32x4 idt[8];
      if (ep.residual) {
        const int colr = n0 + h * 64 + ec;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int64_t row = m0 + wrow0 + mi * 32 + it * 4 + er;
          idt[it] = (row < g.M && colr < g.K) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ep.residual + row * g.K + colr))
                                              : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
      }
      static_for<2>([&](auto jj_c) {
        constexpr int jj = decltype(jj_c)::value;
        constexpr int j = h * 2 + jj < NT ? h * 2 + jj : NT - 1;
        if (h * 2 + jj >= NT) return;
        const int col = n0 + j * 32 + (lane & 31);
        const bool cok = col < g.K;
        const float mult = cok ? sin * s_w[col] : 0.0f;
        const int corr = cok ? (shift - zpi) * wsum[col] : 0;
        const float bv = (cok && bias) ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          float v = (float)(acc[mi][j][i] + corr) * mult + bv;
          if (DUAL) v = v + extra[DUAL ? mi : 0][DUAL ? j : 0][i];
          stg[r * EP_LD + jj * 32 + (lane & 31)] = v;
        }
      });
      // a wave only reads back what it wrote itself: no block barrier, just the LDS counter
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int col = n0 + h * 64 + ec;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + er;
        const int64_t row = m0 + wrow0 + mi * 32 + r;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * EP_LD + ec);
        if (row < g.M && col < g.K) {
          const int64_t at = row * g.K + col;
          if (ep.residual) v = f32x4{v.x + idt[it].x, v.y + idt[it].y, v.z + idt[it].z, v.w + idt[it].w};
          if (ep.relu) v = f32x4{relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)};
          if (out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + at));
          if (ep.codes)
            __builtin_nontemporal_store(eq.code4(v), reinterpret_cast<uint32_t*>(ep.codes + at));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next pass overwrites the stage
    });
    return;
  }
  if constexpr (BM * BN < 256 * 256) {   // (the largest tile is only launched where the staged epilogue applies)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 32 + (lane & 31);
    if (col >= g.K) continue;
    const float mult = sin * s_w[col];
    const int corr = (shift - zpi) * wsum[col];
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t row = m0 + wrow0 + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        const int64_t at = row * g.K + col;
        float v = (float)(acc[mi][j][i] + corr) * mult + bv;
        if (DUAL) v = v + extra[DUAL ? mi : 0][DUAL ? j : 0][i];
        if (ep.residual) v = v + ep.residual[at];
        if (ep.relu) v = relu_nan(v);
        if (out) __builtin_nontemporal_store(v, out + at);
        if (ep.codes) ep.codes[at] = (uint8_t)eq.exact(v);
      }
  }
  }
}

// ------------------------------------------------------------------------------------------------------
// Autonomous-wave variant: NO LDS staging of operands and NO barriers.  Every wave owns a 64-row x (NT*32)-column output
// tile and loads its own A fragments (2 row slabs) and B fragments (NT column slabs) straight from global memory into
// registers, one K step ahead of the multiplies; the four waves of a workgroup work on four different row tiles of the
// same columns, so their B loads coincide in the vector L1.  Motivation: in the LDS-ring kernels the
// DMA, LDS-read and MFMA phases add up instead of overlapping - every variant of them lands at 1.0-1.2 POP/s; here
// nothing synchronises, waves drift apart and the phases of different waves interleave.
template <int NT>
__global__ __launch_bounds__(256, 1) void conv_i8_aw_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                            float* __restrict__ out, const float* __restrict__ bias,
                                                            const int32_t* __restrict__ wsum, const float* __restrict__ s_in,
                                                            const float* __restrict__ zp_in, const float* __restrict__ s_w,
                                                            ConvGeom g, int shift, ConvEpi ep) {
  constexpr int BK = CV_BK, KS = BK / 32, MT = 2;
  constexpr int EP_LD = 68;
  __shared__ __attribute__((aligned(16))) float stage[4 * 32 * EP_LD];   // epilogue transposition only (private per wave)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hsel = lane >> 5;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  const int n0 = blockIdx.y * (NT * 32);
  if (m0 >= g.M) return;                                   // no barrier anywhere: a wave may simply leave
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const int8_t* padline = g_pad_table.b + ((zpi & 0xff) << 6);
  const uint32_t xorw = shift ? 0x80808080u : 0u;

  int a_n[MT], a_h0[MT], a_w0[MT];
  bool a_ok[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int64_t m = m0 + mi * 32 + (lane & 31);
    a_ok[mi] = m < g.M;
    row_origin(g, a_ok[mi] ? (uint32_t)m : 0u, a_n[mi], a_h0[mi], a_w0[mi]);
  }
  const int8_t* ap[MT];
  int a_inc[MT];
  int t_r = 0, t_s = 0, t_cc = 0;
  auto retap = [&]() {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int h = a_h0[mi] + t_r * g.dil, ww = a_w0[mi] + t_s * g.dil;
      const bool in = a_ok[mi] && h >= 0 && h < g.H && ww >= 0 && ww < g.W;
      ap[mi] = in ? x + (((int64_t)a_n[mi] * g.H + h) * g.W + ww) * g.C + hsel * 16 : padline;
      a_inc[mi] = in ? BK : 0;
    }
  };
  retap();
  const int64_t wrow = (int64_t)g.R * g.S * g.C;
  const int8_t* bp[NT];
  int b_inc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int k = n0 + j * 32 + (lane & 31);
    bp[j] = k < g.K ? w + (int64_t)k * wrow + hsel * 16 : g_pad_table.b;     // KRSC: the reduction index is contiguous
    b_inc[j] = k < g.K ? BK : 0;
  }
  const int cchunks = g.C / BK;
  const int nsteps = g.R * g.S * cchunks;

  typedef const __attribute__((address_space(1))) i32x4* gvec_t;
  auto fetch = [&](i32x4 (&a)[MT][KS], i32x4 (&b)[KS][NT]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[mi][ks] = *reinterpret_cast<gvec_t>((gptr_t)(ap[mi] + ks * 32));
      ap[mi] += a_inc[mi];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) b[ks][j] = *reinterpret_cast<gvec_t>((gptr_t)(bp[j] + ks * 32));
      bp[j] += b_inc[j];
    }
    if (++t_cc == cchunks) {
      t_cc = 0;
      if (++t_s == g.S) {
        t_s = 0;
        ++t_r;
      }
      retap();
    }
  };
  i32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][j][i] = 0;
  auto multiply = [&](i32x4 (&a)[MT][KS], i32x4 (&b)[KS][NT]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      i32x4 af[MT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        af[mi] = i32x4{(int)(a[mi][ks].x ^ xorw), (int)(a[mi][ks].y ^ xorw), (int)(a[mi][ks].z ^ xorw), (int)(a[mi][ks].w ^ xorw)};
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[mi], b[ks][j], acc[mi][j], 0, 0, 0);
    }
  };
  i32x4 a0[MT][KS], b0[KS][NT], a1[MT][KS], b1[KS][NT];
  fetch(a0, b0);
  for (int step = 0; step < nsteps; step += 2) {
    if (step + 1 < nsteps) fetch(a1, b1);
    multiply(a0, b0);
    if (step + 1 < nsteps) {
      if (step + 2 < nsteps) fetch(a0, b0);
      multiply(a1, b1);
    }
  }

  // ---- epilogue: as in the LDS-ring kernel, 32 rows x 64 columns at a time through this wave's private stage ----
  const float sin = s_in[0];
  const EpiQuant eq(ep);
  float* stg = stage + wave * (32 * EP_LD);
  const int er = lane >> 4, ec = (lane & 15) * 4;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int h = 0; h < NT / 2 + (NT & 1); ++h) {
      f32x4 idt[8];
      const int colv = n0 + h * 64 + ec;
      if (ep.residual) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int64_t row = m0 + mi * 32 + it * 4 + er;
          idt[it] = (row < g.M && colv + 3 < g.K) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ep.residual + row * g.K + colv))
                                                  : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = h * 2 + jj;
        if (j >= NT) continue;
        const int col = n0 + j * 32 + (lane & 31);
        const bool cok = col < g.K;
        const float mult = cok ? sin * s_w[col] : 0.0f;
        const int corr = cok ? (shift - zpi) * wsum[col] : 0;
        const float bv = (cok && bias) ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (i & 3) + 8 * (i >> 2) + 4 * hsel;
          stg[r * EP_LD + jj * 32 + (lane & 31)] = (float)(acc[mi][j][i] + corr) * mult + bv;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + er;
        const int64_t row = m0 + mi * 32 + r;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * EP_LD + ec);
        if (row < g.M && colv + 3 < g.K) {
          const int64_t at = row * g.K + colv;
          if (ep.residual) v = f32x4{v.x + idt[it].x, v.y + idt[it].y, v.z + idt[it].z, v.w + idt[it].w};
          if (ep.relu) v = f32x4{relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)};
          if (out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + at));
          if (ep.codes) __builtin_nontemporal_store(eq.code4(v), reinterpret_cast<uint32_t*>(ep.codes + at));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Wave-specialised variant: 8 waves per workgroup - waves 0-3 only multiply (MFMA + ds_read), waves 4-7 only feed
// (global_load_lds).  Issuing an LDS-DMA instruction costs the issuing wave ~60-185 cycles (MI355X_MICROARCH.md,
// cycle constants), i.e. 4 of them per K step cost more than that step's 8 MFMAs; in the single-role kernel above
// every wave pays both in sequence.  Here each SIMD hosts one consumer and one loader wave of the workgroup and the
// two run concurrently (separate pipes).  Same protocol: 3 LDS buffers, loaders wait vmcnt for step k, ONE
// s_barrier per step joined by all 8 waves, loaders then refill the buffer step k-1 released.
template <int BN>
__global__ __launch_bounds__(512) void conv_i8_ws_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                        float* __restrict__ out, const float* __restrict__ bias,
                                                        const int32_t* __restrict__ wsum, const float* __restrict__ s_in,
                                                        const float* __restrict__ zp_in, const float* __restrict__ s_w,
                                                        ConvGeom g, int shift) {
  constexpr int BM = 128, NBUF = 3;
  constexpr int TILE_A = BM * CV_BK, TILE_B = BN * CV_BK, TILE = TILE_A + TILE_B;
  constexpr int NT = BN / 32, AI = BM / 64, BI = BN / 64;
  __shared__ __attribute__((aligned(1024))) int8_t lds[NBUF * TILE];

  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t qd = nwg >> 3, rm = nwg & 7u;
  const uint32_t tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
  const int bn = tile % g.nblk_n, bm = tile / g.nblk_n;
  const int64_t m0 = (int64_t)bm * BM;
  const int n0 = bn * BN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const int cchunks = g.C / CV_BK;
  const int nsteps = g.R * g.S * cchunks;

  if (wave >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - 4;
    const int8_t* padline = g_pad_table.b + ((zpi & 0xff) << 6);
    const int lrow = lane >> 2, pslot = lane & 3;
    int a_n[AI], a_h0[AI], a_w0[AI], a_seg[AI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = (i * 4 + lw) * 16 + lrow;
      a_seg[i] = pslot ^ ((row >> 2) & 3);
      const int64_t m = m0 + row;
      a_ok[i] = m < g.M;
      row_origin(g, a_ok[i] ? (uint32_t)m : 0u, a_n[i], a_h0[i], a_w0[i]);
    }
    int b_seg[BI];
    const int8_t* b_src[BI];
    const int64_t wrow = (int64_t)g.R * g.S * g.C;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = (i * 4 + lw) * 16 + lrow;
      b_seg[i] = pslot ^ ((row >> 2) & 3);
      const int k = n0 + row;
      b_src[i] = k < g.K ? w + (int64_t)k * wrow : nullptr;
    }
    int f_cc = 0, f_s = 0, f_r = 0, f_buf = 0;
    auto issue = [&]() {
      int8_t* base = lds + f_buf * TILE;
      const int rs = f_r * g.S + f_s;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int h = a_h0[i] + f_r * g.dil, ww = a_w0[i] + f_s * g.dil;
        const int8_t* src = padline;
        if (a_ok[i] && h >= 0 && h < g.H && ww >= 0 && ww < g.W)
          src = x + (((int64_t)a_n[i] * g.H + h) * g.W + ww) * g.C + f_cc * CV_BK + a_seg[i] * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (i * 4 + lw) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int8_t* src = b_src[i] ? b_src[i] + (int64_t)rs * g.C + f_cc * CV_BK + b_seg[i] * 16 : g_pad_table.b;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + TILE_A + (i * 4 + lw) * 1024), 16, 0, 0);
      }
      if (++f_cc == cchunks) {
        f_cc = 0;
        if (++f_s == g.S) {
          f_s = 0;
          ++f_r;
        }
      }
      if (++f_buf == NBUF) f_buf = 0;
    };
    issue();
    if (nsteps > 1) issue();
    for (int step = 0; step < nsteps; ++step) {
      if (step + 1 < nsteps) {
        if (AI + BI == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (step + 2 < nsteps) issue();
    }
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const uint32_t xorw = shift ? 0x80808080u : 0u;
  i32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0;
  const int arow = wave * 32 + (lane & 31), hsel = lane >> 5;
  const int a_sw = (arow >> 2) & 3;
  int c_buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    __builtin_amdgcn_s_barrier();
    const int8_t* base = lds + c_buf * TILE;
#pragma unroll
    for (int ks = 0; ks < CV_BK / 32; ++ks) {
      const int sg = ks * 2 + hsel;
      i32x4 af = *reinterpret_cast<const i32x4*>(base + arow * 64 + ((sg ^ a_sw) << 4));
      af = i32x4{(int)(af.x ^ xorw), (int)(af.y ^ xorw), (int)(af.z ^ xorw), (int)(af.w ^ xorw)};
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int brow = j * 32 + (lane & 31);
        const i32x4 bf = *reinterpret_cast<const i32x4*>(base + TILE_A + brow * 64 + ((sg ^ ((brow >> 2) & 3)) << 4));
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[j], 0, 0, 0);
      }
    }
    if (++c_buf == NBUF) c_buf = 0;
  }
  const float sin = s_in[0];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 32 + (lane & 31);
    if (col >= g.K) continue;
    const float mult = sin * s_w[col];
    const int corr = (shift - zpi) * wsum[col];
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t row = m0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (row < g.M) __builtin_nontemporal_store((float)(acc[j][i] + corr) * mult + bv, out + row * g.K + col);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// 256-row wave-specialised kernel for the layers with a long reduction (3x3 taps, many input channels), where
// the ablations above show the operand stream - not the matrix cores - setting the pace.  One workgroup per CU:
//   8 consumer waves (MFMA + ds_read only), arranged (8/WN) x WN with WN = BN/64, each owning a (256*WN/8) x 64
//   output slab; 4 loader waves (global_load_lds only) feeding a 3-buffer ring of (256 + BN) x 64-byte tiles.
// Versus the 128 x 128 tile this halves (BN = 256) the operand bytes per MAC, and the loaders' DMA issue runs
// beside the consumers' MFMAs instead of in front of them.  Same hand-off as above: loaders wait (counted vmcnt)
// for step k, one s_barrier per step joined by all 12 waves, loaders refill the buffer step k-1 released.
template <int BN, int NBUF = 3>
__global__ __launch_bounds__(768) void conv_i8_ws256_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                           float* __restrict__ out, const float* __restrict__ bias,
                                                           const int32_t* __restrict__ wsum,
                                                           const float* __restrict__ s_in,
                                                           const float* __restrict__ zp_in,
                                                           const float* __restrict__ s_w, ConvGeom g, int shift) {
  constexpr int BM = 256, PF = NBUF - 1;
  constexpr int TILE_A = BM * CV_BK, TILE_B = BN * CV_BK, TILE = TILE_A + TILE_B;
  constexpr int WN = BN / 64, WM = 8 / WN;      // consumer grid
  constexpr int WROWS = BM / WM, MT = WROWS / 32, NT = 2;
  constexpr int AI = BM / 64, BI = BN / 64;     // DMA wave-instructions per LOADER wave per step (16 rows each)
  constexpr int EP_LD = 68;                     // floats per staged epilogue row (64 + 4 pad)
  constexpr int EP_BYTES = 8 * 32 * EP_LD * 4;  // the epilogue stage re-uses the operand ring
  constexpr int LDS_BYTES = NBUF * TILE > EP_BYTES ? NBUF * TILE : EP_BYTES;
  __shared__ __attribute__((aligned(1024))) int8_t lds[LDS_BYTES];

  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t qd = nwg >> 3, rm = nwg & 7u;
  const uint32_t tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
  const int bn = tile % g.nblk_n, bm = tile / g.nblk_n;
  const int64_t m0 = (int64_t)bm * BM;
  const int n0 = bn * BN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const int cchunks = g.C / CV_BK;
  const int nsteps = g.R * g.S * cchunks;

  if (wave >= 8) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - 8;
    const int8_t* padline = g_pad_table.b + ((zpi & 0xff) << 6);
    const int lrow = lane >> 2, pslot = lane & 3;
    int a_n[AI], a_h0[AI], a_w0[AI], a_seg[AI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int row = (i * 4 + lw) * 16 + lrow;
      a_seg[i] = pslot ^ ((row >> 2) & 3);
      const int64_t m = m0 + row;
      a_ok[i] = m < g.M;
      row_origin(g, a_ok[i] ? (uint32_t)m : 0u, a_n[i], a_h0[i], a_w0[i]);
    }
    int b_seg[BI];
    const int8_t* b_src[BI];
    const int64_t wrow = (int64_t)g.R * g.S * g.C;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int row = (i * 4 + lw) * 16 + lrow;
      b_seg[i] = pslot ^ ((row >> 2) & 3);
      const int k = n0 + row;
      b_src[i] = k < g.K ? w + (int64_t)k * wrow : nullptr;
    }
    int f_cc = 0, f_s = 0, f_r = 0, f_buf = 0;
    auto issue = [&]() {
      int8_t* base = lds + f_buf * TILE;
      const int rs = f_r * g.S + f_s;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int h = a_h0[i] + f_r * g.dil, ww = a_w0[i] + f_s * g.dil;
        const int8_t* src = padline;
        if (a_ok[i] && h >= 0 && h < g.H && ww >= 0 && ww < g.W)
          src = x + (((int64_t)a_n[i] * g.H + h) * g.W + ww) * g.C + f_cc * CV_BK + a_seg[i] * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + (i * 4 + lw) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int8_t* src = b_src[i] ? b_src[i] + (int64_t)rs * g.C + f_cc * CV_BK + b_seg[i] * 16 : g_pad_table.b;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + TILE_A + (i * 4 + lw) * 1024), 16, 0, 0);
      }
      if (++f_cc == cchunks) {
        f_cc = 0;
        if (++f_s == g.S) {
          f_s = 0;
          ++f_r;
        }
      }
      if (++f_buf == NBUF) f_buf = 0;
    };
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (i < nsteps) issue();
    for (int step = 0; step < nsteps; ++step) {
      if (step + PF - 1 < nsteps) {   // keep the PF-1 younger steps' instructions in flight
        constexpr int KEEP = (PF - 1) * (AI + BI);
        if (KEEP == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (KEEP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (KEEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (KEEP == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (KEEP == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (KEEP == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (KEEP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (KEEP == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (KEEP == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (KEEP == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (KEEP == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (step + PF < nsteps) issue();
    }
    __builtin_amdgcn_s_barrier();   // pairs with the consumers' pre-epilogue barrier
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const uint32_t xorw = shift ? 0x80808080u : 0u;
  const int wm = wave / WN, wn = wave % WN;
  const int wrow0 = wm * WROWS, wcol0 = wn * 64;
  i32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][j][i] = 0;
  const int hsel = lane >> 5;
  int c_buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    __builtin_amdgcn_s_barrier();
    const int8_t* base = lds + c_buf * TILE;
#pragma unroll
    for (int ks = 0; ks < CV_BK / 32; ++ks) {
      const int sg = ks * 2 + hsel;
      i32x4 bf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int brow = wcol0 + j * 32 + (lane & 31);
        bf[j] = *reinterpret_cast<const i32x4*>(base + TILE_A + brow * 64 + ((sg ^ ((brow >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int arow = wrow0 + mi * 32 + (lane & 31);
        i32x4 af = *reinterpret_cast<const i32x4*>(base + arow * 64 + ((sg ^ ((arow >> 2) & 3)) << 4));
        af = i32x4{(int)(af.x ^ xorw), (int)(af.y ^ xorw), (int)(af.z ^ xorw), (int)(af.w ^ xorw)};
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[mi][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[j], acc[mi][j], 0, 0, 0);
      }
    }
    if (++c_buf == NBUF) c_buf = 0;
  }

  // ---- epilogue through LDS (per wave: 32 rows x 64 channels per pass), dwordx4 stores of 256 contiguous bytes ----
  __builtin_amdgcn_s_barrier();                     // all consumers are done with the operand ring
  float* stg = reinterpret_cast<float*>(lds) + wave * (32 * EP_LD);
  const float sin = s_in[0];
  const int er = lane >> 4, ec = (lane & 15) * 4;
  float mult[NT], bv[NT];
  int corr[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wcol0 + j * 32 + (lane & 31);
    const bool cok = col < g.K;
    mult[j] = cok ? sin * s_w[col] : 0.0f;
    corr[j] = cok ? (shift - zpi) * wsum[col] : 0;
    bv[j] = (cok && bias) ? bias[col] : 0.0f;
  }
  const bool vec_ok = (g.K & 3) == 0;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        stg[r * EP_LD + j * 32 + (lane & 31)] = (float)(acc[mi][j][i] + corr[j]) * mult[j] + bv[j];
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int col = n0 + wcol0 + ec;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 4 + er;
      const int64_t row = m0 + wrow0 + mi * 32 + r;
      if (row >= g.M) continue;
      if (vec_ok) {
        if (col < g.K)
          __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(stg + r * EP_LD + ec),
                                      reinterpret_cast<f32x4*>(out + row * g.K + col));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < g.K) out[row * g.K + col + e] = stg[r * EP_LD + ec + e];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// Weights fp32 KCRS -> int8 KRSC codes (form SYMMETRIC, FSPTQuant/base.py:149-152: q = clamp(R(w/s_k), lo, hi))
// plus SUM_k = sum of the codes of output channel k.  One workgroup per output channel.
__global__ __launch_bounds__(DLMCQ_BLOCK) void quantize_weight_krsc_kernel(const float* __restrict__ w, int8_t* __restrict__ wq,
                                                                          int32_t* __restrict__ wsum,
                                                                          const float* __restrict__ scale, int C, int RS,
                                                                          float lo, float hi) {
  __shared__ int part[DLMCQ_BLOCK / DLMCQ_WAVE];
  const int64_t k = blockIdx.x;
  const float s = scale[k];
  const int n = C * RS;
  int acc = 0;
  for (int i = threadIdx.x; i < n; i += DLMCQ_BLOCK) {
    const int c = i / RS, rs = i - c * RS;                    // input order (c, r, s)
    const float q = clamp_nan(ste_round(w[k * n + i] / s), lo, hi);
    const int code = code_of(q);
    wq[k * n + (int64_t)rs * C + c] = (int8_t)code;           // output order (r, s, c)
    acc += code;
  }
#pragma unroll
  for (int off = DLMCQ_WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DLMCQ_WAVE);
  if ((threadIdx.x & (DLMCQ_WAVE - 1)) == 0) part[threadIdx.x / DLMCQ_WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) wsum[k] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_quantize_weight_krsc_i8(const float* w, int8_t* wq, int32_t* wsum, const float* scale,
                                             int64_t K, int64_t C, int64_t R, int64_t S, int32_t lo, int32_t hi,
                                             dlmcq_stream_t stream) {
  if (K < 0 || C < 1 || R < 1 || S < 1 || lo > hi || lo < -128 || hi > 127) return DLMCQ_EINVAL;
  if (K == 0) return DLMCQ_OK;
  if (!w || !wq || !wsum || !scale) return DLMCQ_EINVAL;
  if (K >= (1ll << 31) || C * R * S >= (1ll << 31)) return DLMCQ_ERANGE;
  hipLaunchKernelGGL(quantize_weight_krsc_kernel, dim3((uint32_t)K), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), w, wq, wsum, scale, (int)C, (int)(R * S), (float)lo, (float)hi);
  return launch_status();
}

static int conv_variant() {
  // 1 = LDS-DMA pipeline (default), 0 = register-staged reference variant.  Read once; for A/B measurements only.
  static const int v = [] {
    const char* e = getenv("DLMCQ_CONV_VARIANT");
    return e ? atoi(e) : 1;
  }();
  return v;
}

// Kernel variants (DLMCQ_CONV_VARIANT / dlmcq_x_conv2d_i8_variant; everything but 1 exists for A/B measurements and tests):
//    1  default dispatch: 128x128 (or 128x64) tiles, weights through a 3-deep LDS-DMA ring, activations straight to
//       registers (ADIR) - except 1x1 reductions into <= 64 channels, which keep both operands in the ring; 128x256
//       tiles on grids of <= 32768 rows with >= 512 output channels; the DUAL kernel when a second operand pair is given
//    0  register-staged 2-buffer kernel (the first version)            2  LDS-DMA ring for both operands (no ADIR)
//    3  256-row tiles where eligible                                   4  wave-specialised 128-row (loader + MFMA waves)
//    5 / 6 / 7  ADIR with a 4- / 3- / 5-deep ring                      8  256x128 ADIR
//    9  wave-specialised 256-row tiles (12 waves)                      10 / 11  ADIR with 128-byte K steps, 3 / 2 buffers
//   12  autonomous waves: no LDS operands, no barriers                 15  128x256 ADIR
//   16 / 17  256x256 tiles, one workgroup per CU (16: pinned instruction stream in the steady state; 17: ADIR)
//   13 / 14 / 18  timing-study builds (phase stamps; tools/conv_trace.py) of 6 / 2 / 16
// What each taught is in DESIGN.md 5.1.
static int conv_launch(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                       const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N, int64_t H,
                       int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                       int32_t dilation, int32_t x_is_unsigned, dlmcq_stream_t stream, int variant,
                       const ConvEpi& ep = ConvEpi{}, const ConvSeg2* seg2 = nullptr) {
  const bool fused = ep.residual || ep.codes || ep.relu || seg2;
  const bool fused_dual = seg2 != nullptr;
  if (fused && !(variant == 1 || variant == 2 || variant == 3 || (variant >= 5 && variant <= 18 && variant != 9))) return DLMCQ_EINVAL;   // only the LDS-DMA kernel fuses
  if (N < 0 || H < 1 || W < 1 || C < 1 || K < 1 || R < 1 || S < 1 || stride < 1 || pad < 0 || dilation < 1)
    return DLMCQ_EINVAL;
  if (C % CV_BK != 0) return DLMCQ_EINVAL;  // the K step is 64 input channels
  const int64_t P = (H + 2 * pad - dilation * (R - 1) - 1) / stride + 1;
  const int64_t Q = (W + 2 * pad - dilation * (S - 1) - 1) / stride + 1;
  if (P < 1 || Q < 1) return DLMCQ_EINVAL;
  const int64_t M = N * P * Q;
  if (M == 0) return DLMCQ_OK;
  if (!x || !w || !(out || ep.codes) || !wsum || !in_scale || !w_scale) return DLMCQ_EINVAL;
  if (ep.codes && (!ep.q_scale || ep.q_lo > ep.q_hi || ep.q_lo < -128.0f || ep.q_hi > 255.0f || ep.q_hi - ep.q_lo > 255.0f ||
                   ep.q_form < DLMCQ_FORM_EMULATE || ep.q_form > DLMCQ_FORM_SYMMETRIC))
    return DLMCQ_EINVAL;
  if (!aligned16(x) || !aligned16(w) || (out && !aligned16(out)) || (ep.residual && !aligned16(ep.residual)) ||
      (ep.codes && !aligned4(ep.codes)))
    return DLMCQ_EALIGN;
  if (M >= (1ll << 31) || N * H * W * C >= (1ll << 40) || K >= (1 << 24)) return DLMCQ_ERANGE;
  ConvGeom g;
  g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K; g.R = (int)R; g.S = (int)S;
  g.stride = stride; g.pad = pad; g.dil = dilation; g.P = (int)P; g.Q = (int)Q; g.M = M;
  g.qdiv = make_fastdiv((uint32_t)Q);
  g.pdiv = make_fastdiv((uint32_t)P);
  const int shift = x_is_unsigned ? 128 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int8_t* xs = reinterpret_cast<const int8_t*>(x);
  static const int bn_force = [] { const char* e = getenv("DLMCQ_CONV_BN"); return e ? atoi(e) : 0; }();   // A/B measurements only
  // 64-wide tiles: narrow outputs, and the dual kernel on short reductions (two operand pairs + the shortcut sum keep
  // 190 VGPRs at 128 columns; measured on ResNet-50's first dual layer, 64 -> 256 channels at 56x56: 655 vs 701 us)
  const bool dual_short = seg2 && K <= 256 && R * S * C <= 256;
  const int bnn = (bn_force == 64 || dual_short) ? 64 : ((K <= 64 || (K % 128) != 0) ? 64 : 128);
  // 256-row tiles halve the weight-operand traffic per MAC; they pay off when the reduction is long (3x3 taps or
  // many input channels) and there are enough row tiles to fill the chip.  DLMCQ_CONV_VARIANT=2 forces 128 rows.
  const int64_t ksteps = R * S * (C / CV_BK);
  const int bmm = (variant == 3 && bnn == 128 && ksteps >= 8 && M >= 256 * 512) ? 256 : 128;   // measured: no gain
  // the 256-row wave-specialised kernel: only on request (variant 9)
  const bool ws256 = variant == 9;   // measured: no faster than the 128-row kernel, so not the default
  if (ws256) {
    const int bn256 = (K % 256 == 0) ? 256 : ((K % 128 == 0) ? 128 : 64);
    g.nblk_m = (int)((M + 255) / 256);
    g.nblk_n = (int)((K + bn256 - 1) / bn256);
    const int64_t nwg2 = (int64_t)g.nblk_m * g.nblk_n;
    if (nwg2 >= (1ll << 31)) return DLMCQ_ERANGE;
#define DLMCQ_CONV_ARGS_256 dim3((uint32_t)nwg2), dim3(768), 0, st, xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift
    if (bn256 == 256) hipLaunchKernelGGL((conv_i8_ws256_kernel<256>), DLMCQ_CONV_ARGS_256);
    else if (bn256 == 128) hipLaunchKernelGGL((conv_i8_ws256_kernel<128>), DLMCQ_CONV_ARGS_256);
    else hipLaunchKernelGGL((conv_i8_ws256_kernel<64>), DLMCQ_CONV_ARGS_256);
#undef DLMCQ_CONV_ARGS_256
    return launch_status();
  }
  g.nblk_m = (int)((M + bmm - 1) / bmm);
  g.nblk_n = (int)((K + bnn - 1) / bnn);
  const int64_t nwg = (int64_t)g.nblk_m * g.nblk_n;
  if (nwg >= (1ll << 31)) return DLMCQ_ERANGE;
#define DLMCQ_CONV_ARGS dim3((uint32_t)nwg), dim3(256), 0, st, xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift
#define DLMCQ_CONV_ARGS_WS dim3((uint32_t)nwg), dim3(512), 0, st, xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift
  if (variant == 4) {
    if (bnn == 64) hipLaunchKernelGGL((conv_i8_ws_kernel<64>), DLMCQ_CONV_ARGS_WS);
    else hipLaunchKernelGGL((conv_i8_ws_kernel<128>), DLMCQ_CONV_ARGS_WS);
  } else if (variant == 0) {
    if (bnn == 64) hipLaunchKernelGGL((conv_i8_kernel<64>), DLMCQ_CONV_ARGS);
    else hipLaunchKernelGGL((conv_i8_kernel<128>), DLMCQ_CONV_ARGS);
  } else if (seg2) {
    ConvSeg2 s2 = *seg2;
    s2.g.nblk_m = g.nblk_m;
    s2.g.nblk_n = g.nblk_n;
    if (s2.g.M != g.M || s2.g.K != g.K || s2.g.P != g.P || s2.g.Q != g.Q || bmm != 128) return DLMCQ_EINVAL;
    if (variant != 2) {   // A operand direct to registers
      if (bnn == 64) hipLaunchKernelGGL((conv_i8_dma_kernel<128, 64, 3, CV_BK, 0, true, true>), DLMCQ_CONV_ARGS, ep, s2);
      else hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 3, CV_BK, 0, true, true>), DLMCQ_CONV_ARGS, ep, s2);
    } else if (bnn == 64) hipLaunchKernelGGL((conv_i8_dma_kernel<128, 64, 3, CV_BK, 0, true>), DLMCQ_CONV_ARGS, ep, s2);
    else hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 3, CV_BK, 0, true>), DLMCQ_CONV_ARGS, ep, s2);
  } else if (variant == 8 && bnn == 128 && ksteps >= 8 && M >= 256 * 512) {   // 256-row tiles, A direct
    g.nblk_m = (int)((M + 255) / 256);
    hipLaunchKernelGGL((conv_i8_dma_kernel<256, 128, 3, CV_BK, 0, false, true>), dim3((uint32_t)((int64_t)g.nblk_m * g.nblk_n)), dim3(256), 0, st,
                       xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, ConvSeg2{});
  } else if ((variant == 15 || (variant == 1 && K >= 512 && M <= 32768 && !fused_dual)) && K % 256 == 0) {   // 128 x 256 tiles, A direct: a third fewer
    // operand bytes per MAC; pays where the grid is small anyway (the 7x7 stage: +10-17 %), loses tiles elsewhere
    g.nblk_n = (int)(K / 256);
    hipLaunchKernelGGL((conv_i8_dma_kernel<128, 256, 3, CV_BK, 0, false, true>), dim3((uint32_t)((int64_t)g.nblk_m * g.nblk_n)), dim3(256), 0, st,
                       xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, ConvSeg2{});
  } else if (variant == 16 && K % 256 == 0) {   // 256 x 256 tiles, one workgroup per CU: 4x fewer operand bytes per MAC
    g.nblk_m = (int)((M + 255) / 256);
    g.nblk_n = (int)(K / 256);
    hipLaunchKernelGGL((conv_i8_dma_kernel<256, 256, 3, CV_BK, 0, false, false>), dim3((uint32_t)((int64_t)g.nblk_m * g.nblk_n)), dim3(256), 0, st,
                       xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, ConvSeg2{});
  } else if (variant == 17 && K % 256 == 0) {   // ... with the A operand direct to registers
    g.nblk_m = (int)((M + 255) / 256);
    g.nblk_n = (int)(K / 256);
    hipLaunchKernelGGL((conv_i8_dma_kernel<256, 256, 3, CV_BK, 0, false, true>), dim3((uint32_t)((int64_t)g.nblk_m * g.nblk_n)), dim3(256), 0, st,
                       xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, ConvSeg2{});
  } else if (variant == 18 && K % 256 == 0) {   // timing study of the 256 x 256 tile
    g.nblk_m = (int)((M + 255) / 256);
    g.nblk_n = (int)(K / 256);
    hipLaunchKernelGGL((conv_i8_dma_kernel<256, 256, 3, CV_BK, 64, false, false>), dim3((uint32_t)((int64_t)g.nblk_m * g.nblk_n)), dim3(256), 0, st,
                       xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, ConvSeg2{});
  } else if (variant == 13 && bnn == 128) {   // timing study
    hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 3, CV_BK, 64, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (variant == 14 && bnn == 128) {
    hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 3, CV_BK, 64, false, false>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (variant == 12 && (K & 3) == 0) {     // autonomous waves: no LDS operands, no barriers
    const int64_t mt = (M + 255) / 256;
    if (K % 128 == 0 || K > 96) {
      hipLaunchKernelGGL((conv_i8_aw_kernel<4>), dim3((uint32_t)mt, (uint32_t)((K + 127) / 128)), dim3(256), 0, st, xs, w, out, bias, wsum,
                         in_scale, in_zero_point, w_scale, g, shift, ep);
    } else {
      hipLaunchKernelGGL((conv_i8_aw_kernel<2>), dim3((uint32_t)mt, (uint32_t)((K + 63) / 64)), dim3(256), 0, st, xs, w, out, bias, wsum,
                         in_scale, in_zero_point, w_scale, g, shift, ep);
    }
  } else if (variant == 10 && bnn == 128 && C % 128 == 0) {   // 128-byte K steps (half the barriers), A direct
    hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 3, 128, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (variant == 11 && bnn == 128 && C % 128 == 0) {   // ... 2 buffers
    hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 2, 128, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (variant == 5) {              // A operand direct to registers, 4-deep ring
    if (bnn == 64) hipLaunchKernelGGL((conv_i8_dma_kernel<128, 64, 4, CV_BK, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
    else hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 4, CV_BK, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (variant == 6 || (variant == 1 && !(R * S == 1 && bnn == 64 && C >= 256))) {   // ... 3-deep: the default
    // (measured per layer on ResNet-50 b512: 3-20 % faster than staging A through LDS - one more workgroup per CU -
    //  except for the 1x1 reductions into 64 channels)
    if (bnn == 64) hipLaunchKernelGGL((conv_i8_dma_kernel<128, 64, 3, CV_BK, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
    else hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 3, CV_BK, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (variant == 7) {              // ... 5-deep
    if (bnn == 64) hipLaunchKernelGGL((conv_i8_dma_kernel<128, 64, 5, CV_BK, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
    else hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128, 5, CV_BK, 0, false, true>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (bmm == 256) {
    hipLaunchKernelGGL((conv_i8_dma_kernel<256, 128>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else if (bnn == 64) {
    hipLaunchKernelGGL((conv_i8_dma_kernel<128, 64>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  } else {
    hipLaunchKernelGGL((conv_i8_dma_kernel<128, 128>), DLMCQ_CONV_ARGS, ep, ConvSeg2{});
  }
#undef DLMCQ_CONV_ARGS
#undef DLMCQ_CONV_ARGS_WS
  return launch_status();
}

extern "C" int dlmcq_conv2d_i8_nhwc_f32(const void* x, const int8_t* w, float* out, const float* bias,
                                        const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                        const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                                        int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                                        int32_t x_is_unsigned, dlmcq_stream_t stream) {
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, conv_variant());
}

extern "C" int dlmcq_conv2d_i8_nhwc_fused(const void* x, const int8_t* w, float* out, const float* bias,
                                          const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                          const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                                          int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                                          int32_t x_is_unsigned, const float* residual, int32_t relu, void* codes,
                                          const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                                          int32_t q_form, float q_ste_g, dlmcq_stream_t stream) {
  ConvEpi ep{};
  ep.residual = residual;
  ep.relu = relu != 0;
  ep.codes = static_cast<uint8_t*>(codes);
  ep.q_scale = q_scale;
  ep.q_zp = q_zero_point;
  ep.q_lo = (float)q_lo;
  ep.q_hi = (float)q_hi;
  ep.q_g = q_ste_g;
  ep.q_form = q_form;
  const int v = conv_variant();
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, ((v >= 1 && v <= 3) || (v >= 5 && v <= 17 && v != 9)) ? v : 1, ep);
}

extern "C" int dlmcq_conv2d_i8_nhwc_dual(const void* x, const int8_t* w, float* out, const float* bias,
                                         const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                         const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                                         int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                                         int32_t x_is_unsigned, const void* x2, const int8_t* w2, const float* bias2,
                                         const int32_t* wsum2, const float* in_scale2, const float* in_zero_point2,
                                         const float* w_scale2, int64_t H2, int64_t W2, int64_t C2, int64_t R2,
                                         int64_t S2, int32_t stride2, int32_t pad2, int32_t dilation2,
                                         int32_t x2_is_unsigned, int32_t relu, void* codes, const float* q_scale,
                                         const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form,
                                         float q_ste_g, dlmcq_stream_t stream) {
  if (H2 < 1 || W2 < 1 || C2 < 1 || R2 < 1 || S2 < 1 || stride2 < 1 || pad2 < 0 || dilation2 < 1 || C2 % CV_BK != 0)
    return DLMCQ_EINVAL;
  if (N > 0 && (!x2 || !w2 || !wsum2 || !in_scale2 || !w_scale2)) return DLMCQ_EINVAL;
  if (!aligned16(x2) || !aligned16(w2)) return DLMCQ_EALIGN;
  if (N * H2 * W2 * C2 >= (1ll << 40)) return DLMCQ_ERANGE;
  ConvSeg2 s2{};
  s2.x = static_cast<const int8_t*>(x2);
  s2.w = w2;
  s2.bias = bias2;
  s2.wsum = wsum2;
  s2.s_in = in_scale2;
  s2.zp_in = in_zero_point2;
  s2.s_w = w_scale2;
  s2.shift = x2_is_unsigned ? 128 : 0;
  ConvGeom& g = s2.g;
  const int64_t P = (H2 + 2 * pad2 - dilation2 * (R2 - 1) - 1) / stride2 + 1;
  const int64_t Q = (W2 + 2 * pad2 - dilation2 * (S2 - 1) - 1) / stride2 + 1;
  if (P < 1 || Q < 1) return DLMCQ_EINVAL;
  g.N = (int)N; g.H = (int)H2; g.W = (int)W2; g.C = (int)C2; g.K = (int)K; g.R = (int)R2; g.S = (int)S2;
  g.stride = stride2; g.pad = pad2; g.dil = dilation2; g.P = (int)P; g.Q = (int)Q; g.M = N * P * Q;
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
  const int v = conv_variant();
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, v == 2 ? 2 : 1, ep, &s2);
}

// NOT part of the ABI: timing study - variants 13 (A direct) / 14 (A through LDS) stamp the shader clock at the phase
// boundaries of one wave's first 16 K steps into `trace` (16 x 8 uint64).
extern "C" int dlmcq_x_conv2d_i8_trace(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                       const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                                       int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride,
                                       int32_t pad, int32_t dilation, int32_t x_is_unsigned, dlmcq_stream_t stream,
                                       int32_t variant, void* trace) {
  ConvEpi ep{};
  ep.residual = static_cast<const float*>(trace);
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, variant, ep);
}

// NOT part of the ABI (absent from include/dlmcq.h): the same call with an explicit kernel variant, for the tests and
// for A/B measurements in one process.  0 register-staged, 1 LDS-DMA (default), 3 LDS-DMA with 256-row tiles where
// eligible, 4 wave-specialised loader/consumer, 9 wave-specialised 256-row tiles.  (Deeper rings - 4 / 5 LDS buffers - and 128-byte K steps were also
// measured through the NBUF / BK template parameters and lost 5-40 %: occupancy matters more here)
extern "C" int dlmcq_x_conv2d_i8_variant(const void* x, const int8_t* w, float* out, const float* bias,
                                         const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                         const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                                         int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                                         int32_t x_is_unsigned, dlmcq_stream_t stream, int32_t variant) {
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, variant);
}
