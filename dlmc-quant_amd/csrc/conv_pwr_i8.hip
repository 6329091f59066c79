// A residual block's LAST 1x1 convolution - int8 reduction + fp32 shortcut + ReLU (+ the fp32 block output) + the consumer's codes -
// where no chain kernel takes it (the last block of a ResNet stage: the next layer reads other pixels), with the weights RESIDENT
// in LDS.  (modules/conv.py:13-19 on FSPTQuant/base.py:108-109 operands, then the model's `out += identity; relu` and the next
// layer's quantiser; same integers, same fp32 chain, the same bytes as conv_i8_mfma_kernel's residual epilogue.)
//
// These layers are fp32 streams (4 B shortcut in, 4 B out, 1 B codes per element for a 256..512-deep reduction); the tiled kernel
// runs them at 3.8 TB/s because its phases - operand round trips of a 12-24-step K loop behind a barrier each, the dequantising
// epilogue, the stores - add up inside a workgroup (LABNOTES 12).  conv_pw_i8.hip's structure instead: a workgroup keeps a 128-channel
// slice of the weights in LDS and never synchronises after its prologue; every wave walks 32-pixel blocks on its own - activation
// fragments straight to registers, four passes of C / 32 MFMAs (weights as A: a lane owns 16 consecutive channels of one pixel),
// the pass's 32 x 32 fp32 values through a wave-private LDS stage into row-major order (8 adjacent lanes = one pixel's 128 bytes:
// whole-line shortcut loads and output stores, conv_chain_i8.hip's form 2), shortcut, ReLU, store, the plain quantiser
// (conv_epilogue.h) on four rows at once, the codes through the wave's code stage and out as whole rows at the end of the block.
//
// Every vector-memory operation is an asm buffer instruction and every wait is counted.  Per block and wave, in issue order
// (O = 4 if the fp32 output is stored, else 0):
//     pass 0: L1 (4 loads: pass 1's shortcut rows)            S0 (O stores)
//     pass 1: L2                                               S1
//     pass 2: L3                                               S2
//     pass 3: L0' (next block's pass 0), REQ (NA fragment loads of the next block)   S3
//     end:    CS (4 code stores)
// so: loop top (fragments and L0' landed) vmcnt(O + 4); passes 1, 2 need L(p): vmcnt(O + 4); pass 3 needs L3: vmcnt(O + 4 + NA).
// Lanes outside the tensor issue the same instructions at an out-of-range offset (loads return zeros, stores vanish).
// tools/lint_pw.py walks the listing with exactly this queue: no instruction may touch a register whose load is in flight, no scratch.
#include "conv_i8_common.h"

namespace dlmcq {

struct PwrArgs {
  const int8_t* x;         // [M][C] codes
  const int8_t* w;         // [K][C] int8
  const float* s_w;        // [K]
  const int32_t* wsum;     // [K] SUM qw
  const float* bias;       // [K] or null
  const float* s_in;
  const float* zp_in;      // null: 0
  const float* residual;   // [M][K] fp32
  float* out;              // [M][K] fp32 (OUTF instantiations)
  // layout of those two (one layout per call): bytes from row to row / from a 64-channel chunk to the next.  Row-major [M][K]: K * 4, 256.
  // Chunk-major [K / 64][M][64] (DLMCQ_FP32_*_CHUNK_MAJOR, conv_chain_i8.hip's block tensors): 256, M * 256
  int f_rowb, f_chunk;
  int M, K, shift, nslice; // nslice = K / 128 column slices; workgroup b: slice (b >> 3) % nslice, group ((b >> 3) / nslice) * 8 + (b & 7)
  int nblk;                // 32-pixel blocks (M is a multiple of 32: the launcher checks)
  // C2 > 0 (a stage's first block): the shortcut is a second 1x1 convolution into the same pixels, reduced first - x2 codes
  // [N][H2][W2][C2] sampled at (p * stride2, q * stride2), w2 [K][C2] (conv_i8.hip's DUAL form; no fp32 shortcut then)
  const int8_t* x2;
  const int8_t* w2;
  const float* s_w2;
  const int32_t* wsum2;
  const float* bias2;
  const float* s_in2;
  const float* zp_in2;
  int shift2, P, Q, H2, W2, stride2;
  FastDiv qdiv, pdiv;
};

constexpr int PWR_BN = 128;
constexpr int PWR_FROW = 36;    // floats per row of the fp32 stage (32 + 4: conflict-free 16-byte writes from the accumulator layout)

__device__ __forceinline__ void bload16s(f32x4& dst, int voff, const v4i& rsrc, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" DLMCQ_NT : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void bstore16s(const f32x4& v, int voff, const v4i& rsrc, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" DLMCQ_NT "\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// CODES: the consumer's codes are emitted (false: the fp32 output alone - a network's last block)
// SWP (dual form): the sampled pair is the call's FIRST pair: the sum is (sampled) + (row by row), as conv_i8.hip adds them
template <int C, int NW, bool OUTF, bool CODES = true, int C2 = 0, bool SWP = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_pwr_i8_kernel(PwrArgs a, ConvEpi ep) {
  static_assert(OUTF || CODES, "nothing to produce");
  constexpr bool RES = C2 == 0;         // an fp32 shortcut tensor (the dual form adds its second reduction instead)
  constexpr int NA2 = C2 / 32;
  constexpr int WB2 = C2 * PWR_BN;
  constexpr int NPAR = RES ? 3 : 6;
  constexpr int BN = PWR_BN;
  constexpr int S = C / 64;             // 64-byte K steps
  constexpr int NA = C / 32;            // A fragments (16 bytes per lane each)
  constexpr int NP = BN / 32;           // passes = 32-channel accumulator blocks
  constexpr int WB = C * BN;            // weight bytes of the slice
  constexpr int PIECES = S * (BN / 16); // 1 KB DMA pieces of the slice
  constexpr int O = OUTF ? 4 : 0;
  constexpr int SROW = BN + 16;         // a staged row of codes
  constexpr int NST = CODES ? 4 : 0;    // code stores per block: 8 rows x 128 bytes each
  extern __shared__ __attribute__((aligned(1024))) int8_t pwr_lds[];
  int8_t* const wl = pwr_lds;                   // [S][BN rows][64 B], LDS slot p of row r = logical segment p ^ ((r >> 2) & 3)
  int8_t* const wl2 = pwr_lds + WB;             // the shortcut convolution's slice, same form
  int8_t* const par = pwr_lds + WB + WB2;       // s_in s_w | (shift - zp) SUM qw | bias (| the same of the shortcut convolution)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  int8_t* const stg = pwr_lds + WB + WB2 + NPAR * BN * 4 + wave * (32 * SROW + 32 * PWR_FROW * 4);    // this wave's code rows ...
  float* const fst = reinterpret_cast<float*>(stg + 32 * SROW);                               // ... and its fp32 block
  const int rsel = lane >> 3, q4 = lane & 7;    // the lane's place in the row-major forms: row (8 it + rsel), 16 bytes q4 of it

  const uint32_t b = blockIdx.x;
  const int slice = (int)((b >> 3) % (uint32_t)a.nslice);
  const int group = (int)((b >> 3) / (uint32_t)a.nslice) * 8 + (int)(b & 7u);
  const int ngroups = (int)(gridDim.x / (uint32_t)a.nslice);
  const int n0 = slice * BN;

  // ---- once per workgroup: constants and weights by LDS-DMA ----
  {
    const void* arrs[6] = {a.s_w, a.wsum, a.bias, a.s_w2, a.wsum2, a.bias2};
#pragma unroll
    for (int r = 0; r < NPAR; ++r) {
      if (!arrs[r]) {
        for (int c = tid; c < BN; c += NW * 64) reinterpret_cast<float*>(par)[r * BN + c] = 0.0f;
        continue;
      }
      for (int c = wave; c < BN / 64; c += NW)
        __builtin_amdgcn_global_load_lds((gptr_t)(static_cast<const int32_t*>(arrs[r]) + n0 + c * 64 + lane), (lptr_t)(par + (r * BN + c * 64) * 4), 4, 0, 0);
    }
    const int lrow = lane >> 2, pslot = lane & 3;
    for (int pc = wave; pc < PIECES; pc += NW) {
      const int s = pc / (BN / 16), r16 = pc - s * (BN / 16);
      const int row = r16 * 16 + lrow;               // LDS row; it holds channel 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3) of its 32-block (conv_i8.hip, SWAP)
      const int d = row & 31;
      const int k = n0 + (row & ~31) + 16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
      const int seg = pslot ^ ((row >> 2) & 3);
      __builtin_amdgcn_global_load_lds((gptr_t)(a.w + (int64_t)k * C + s * 64 + seg * 16), (lptr_t)(wl + s * (BN * 64) + r16 * 1024), 16, 0, 0);
    }
    for (int pc = wave; pc < (C2 / 64) * (BN / 16); pc += NW) {
      const int s = pc / (BN / 16), r16 = pc - s * (BN / 16);
      const int row = r16 * 16 + lrow;
      const int d = row & 31;
      const int k = n0 + (row & ~31) + 16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
      const int seg = pslot ^ ((row >> 2) & 3);
      __builtin_amdgcn_global_load_lds((gptr_t)(a.w2 + (int64_t)k * C2 + s * 64 + seg * 16), (lptr_t)(wl2 + s * (BN * 64) + r16 * 1024), 16, 0, 0);
    }
  }
  const float zpf = a.zp_in ? a.zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin = a.s_in[0];
  const uint32_t xorw = a.shift ? 0x80808080u : 0u;
  const EpiQuant eq(ep, true);                    // code(relu(v)) = max(code(v), code(0)): the byte conversion's own saturation
  const v4i r_x = make_rsrc(a.x, (uint32_t)((int64_t)a.M * C));
  const v4i r_c = make_rsrc(CODES ? ep.codes : nullptr, CODES ? (uint32_t)((int64_t)a.M * a.K) : 0u);
  const v4i r_r = make_rsrc(RES ? a.residual : nullptr, RES ? (uint32_t)((int64_t)a.M * a.K * 4) : 0u);
  const v4i r_x2 = make_rsrc(RES ? nullptr : a.x2, RES ? 0u : (uint32_t)((int64_t)(a.M / (a.P * a.Q)) * a.H2 * a.W2 * C2));
  const uint32_t xorw2 = (!RES && a.shift2) ? 0x80808080u : 0u;
  const v4i r_o = make_rsrc(OUTF ? a.out : nullptr, OUTF ? (uint32_t)((int64_t)a.M * a.K * 4) : 0u);
  const int so8 = __builtin_amdgcn_readfirstlane(8 * a.f_rowb);     // eight fp32 rows further, as a scalar offset
  const int fchunk = __builtin_amdgcn_readfirstlane(a.f_chunk);
  // byte offset of pass p's 32 channels from the slice's first: its 64-channel chunk, then the half of it
  auto poff = [&](int p) { return (p >> 1) * fchunk + (p & 1) * 128; };

  i32x4 areg[NA];
  i32x4 areg2[RES ? 1 : NA2];
  f32x4 res[2][4];
  auto request = [&](int blk) {       // this lane's fragments of block `blk`: bytes 32 f + 16 hsel .. + 15 of pixel 32 blk + l31
    const int vo = blk < a.nblk ? (blk * 32 + l31) * C + hsel * 16 : BUF_BIG;
    static_for<NA>([&](auto f) { bload16i<decltype(f)::value * 32>(areg[decltype(f)::value], vo, r_x); });
    if constexpr (!RES) {             // ... and of the shortcut convolution's input pixel (n, p * stride2, q * stride2)
      const uint32_t m = (uint32_t)(blk * 32 + l31);
      const uint32_t t = fdiv(m, a.qdiv);
      const int q = (int)(m - t * (uint32_t)a.Q);
      const uint32_t n = fdiv(t, a.pdiv);
      const int pp = (int)(t - n * (uint32_t)a.P);
      const int vo2 = blk < a.nblk ? (((int)n * a.H2 + pp * a.stride2) * a.W2 + q * a.stride2) * C2 + hsel * 16 : BUF_BIG;
      static_for<NA2>([&](auto f) { bload16i<(decltype(f)::value % 64) * 32>(areg2[decltype(f)::value], vo2 + (decltype(f)::value / 64) * 2048, r_x2); });
    }
  };
  // the shortcut rows of pass `p` of block `blk`: rows 8 it + rsel, bytes 16 q4 .. + 15 of the pass's 128
  auto fbase = [&](int blk) { return blk < a.nblk ? (blk * 32 + rsel) * a.f_rowb + slice * 2 * fchunk + 16 * q4 : BUF_BIG; };
  const int stride = ngroups * NW;
  int blk = group * NW + wave;
  request(blk);
  if constexpr (RES) {
    const int vb = fbase(blk);
#pragma unroll
    for (int it = 0; it < 4; ++it) bload16s(res[0][it], vb, r_r, it * so8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < BN) {      // (s_w, SUM qw) -> the epilogue's (s_in * s_w, (shift - zp) * SUM qw), once per channel
    float* pf = reinterpret_cast<float*>(par) + tid;
    int* pi = reinterpret_cast<int*>(par) + BN + tid;
    *pf = sin * *pf;
    *pi = (a.shift - zpi) * *pi;
    if constexpr (!RES) {
      const float zp2 = a.zp_in2 ? a.zp_in2[0] : 0.0f;
      pf[3 * BN] = a.s_in2[0] * pf[3 * BN];
      pi[3 * BN] = (a.shift2 - (int)__builtin_rintf(zp2)) * pi[3 * BN];
    }
  }
  __syncthreads();

  for (; blk < a.nblk; blk += stride) {
    // the fragments of this block and the shortcut rows of its pass 0 have landed; the previous block's last O output stores and
    // its 4 code stores may still be in flight (the first block: the prologue waited for everything)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(O + NST) : "memory");
#pragma unroll
    for (int f = 0; f < NA; ++f) {
      asm volatile("" : "+v"(areg[f]));     // the asm-loaded fragments are valid from here on
      areg[f] = i32x4{(int)(areg[f].x ^ xorw), (int)(areg[f].y ^ xorw), (int)(areg[f].z ^ xorw), (int)(areg[f].w ^ xorw)};   // uint8 -> int8, once for all passes
    }
    if constexpr (!RES) {
#pragma unroll
      for (int f = 0; f < NA2; ++f) {
        asm volatile("" : "+v"(areg2[f]));
        areg2[f] = i32x4{(int)(areg2[f].x ^ xorw2), (int)(areg2[f].y ^ xorw2), (int)(areg2[f].z ^ xorw2), (int)(areg2[f].w ^ xorw2)};
      }
    }
    const int vb = fbase(blk);                 // this block's fp32 rows
    const int vbn = fbase(blk + stride);       // the next block's
    static_for<NP>([&](auto p_c) {
      constexpr int p = decltype(p_c)::value;
      constexpr int cur = p & 1, nxt = cur ^ 1;
      // the next pass's shortcut rows (the last pass: pass 0 of the next block)
      if constexpr (RES) {
#pragma unroll
        for (int it = 0; it < 4; ++it) bload16s(res[nxt][it], (p + 1 < NP ? vb + poff(p + 1) : vbn), r_r, it * so8);
      }
      // the dual form: the shortcut convolution of these 32 channels first, dequantised into registers
      f32x2 ex[RES ? 1 : 8];
      if constexpr (!RES) {
        i32x16 acc2;
        const i32x4* cop = reinterpret_cast<const i32x4*>(par + (4 * BN + p * 32 + hsel * 16) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const i32x4 c4 = cop[q];
          acc2[4 * q] = c4.x;
          acc2[4 * q + 1] = c4.y;
          acc2[4 * q + 2] = c4.z;
          acc2[4 * q + 3] = c4.w;
        }
        static_for<NA2>([&](auto f_c) {
          constexpr int f = decltype(f_c)::value;
          constexpr int s = f >> 1, ks = f & 1;
          const int sg = ks * 2 + hsel;
          const int brow = p * 32 + l31;
          const i32x4 bf = *reinterpret_cast<const i32x4*>(wl2 + s * (BN * 64) + brow * 64 + ((sg ^ ((brow >> 2) & 3)) << 4));
          acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, areg2[f], acc2, 0, 0, 0);
        });
        const int cb2 = p * 32 + hsel * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (3 * BN + cb2 + 4 * q) * 4);
          const f32x4 bs = *reinterpret_cast<const f32x4*>(par + (5 * BN + cb2 + 4 * q) * 4);
          ex[RES ? 0 : 2 * q] = pk_fma(f32x2{(float)acc2[4 * q], (float)acc2[4 * q + 1]}, f32x2{mu.x, mu.y}, f32x2{bs.x, bs.y});
          ex[RES ? 0 : 2 * q + 1] = pk_fma(f32x2{(float)acc2[4 * q + 2], (float)acc2[4 * q + 3]}, f32x2{mu.z, mu.w}, f32x2{bs.z, bs.w});
        }
      }
      // the accumulators start from (shift - zp) * SUM qw of their channels (register i: channel 32 p + 16 hsel + i)
      i32x16 acc;
      {
        const i32x4* cop = reinterpret_cast<const i32x4*>(par + (BN + p * 32 + hsel * 16) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const i32x4 c4 = cop[q];
          acc[4 * q] = c4.x;
          acc[4 * q + 1] = c4.y;
          acc[4 * q + 2] = c4.z;
          acc[4 * q + 3] = c4.w;
        }
      }
      static_for<NA>([&](auto f_c) {
        constexpr int f = decltype(f_c)::value;
        constexpr int s = f >> 1, ks = f & 1;
        const int sg = ks * 2 + hsel;
        const int brow = p * 32 + l31;
        const i32x4 bf = *reinterpret_cast<const i32x4*>(wl + s * (BN * 64) + brow * 64 + ((sg ^ ((brow >> 2) & 3)) << 4));
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, areg[f], acc, 0, 0, 0);
      });
      if constexpr (p == NP - 1) request(blk + stride);     // (the fragments have been read for the last time)
      // ---- the pass's 32 x 32 values: dequantised on channel pairs in the accumulator layout, then row-major through the stage ----
      const int cb = p * 32 + hsel * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (cb + 4 * q) * 4);
        const f32x4 bs = *reinterpret_cast<const f32x4*>(par + (2 * BN + cb + 4 * q) * 4);
        f32x2 ya = pk_fma(f32x2{(float)acc[4 * q], (float)acc[4 * q + 1]}, f32x2{mu.x, mu.y}, f32x2{bs.x, bs.y});
        f32x2 yb = pk_fma(f32x2{(float)acc[4 * q + 2], (float)acc[4 * q + 3]}, f32x2{mu.z, mu.w}, f32x2{bs.z, bs.w});
        if constexpr (!RES) {
          ya = SWP ? ex[RES ? 0 : 2 * q] + ya : ya + ex[RES ? 0 : 2 * q];
          yb = SWP ? ex[RES ? 0 : 2 * q + 1] + yb : yb + ex[RES ? 0 : 2 * q + 1];
        }
        *reinterpret_cast<f32x4*>(fst + l31 * PWR_FROW + hsel * 16 + 4 * q) = f32x4{ya.x, ya.y, yb.x, yb.y};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave reads back only what it wrote itself)
      // this pass's shortcut rows: behind them in the queue are the previous pass's O stores, the 4 loads above and, in the last
      // pass, the NA fragment loads (pass 0: the loop top has waited)
      if constexpr (RES && p > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(O + 4 + (p == NP - 1 ? NA : 0)) : "memory");
      f32x4 v[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(fst + (it * 8 + rsel) * PWR_FROW + 4 * q4);
        if constexpr (RES) {
          asm volatile("" : "+v"(res[cur][it]));
          const f32x2 va = f32x2{y.x, y.y} + f32x2{res[cur][it].x, res[cur][it].y};
          const f32x2 vc = f32x2{y.z, y.w} + f32x2{res[cur][it].z, res[cur][it].w};
          v[it] = f32x4{va.x, va.y, vc.x, vc.y};
        } else {
          v[it] = y;
        }
        if constexpr (OUTF) {
          v[it] = relu4_nan(v[it]);
          bstore16s(v[it], vb + poff(p), r_o, it * so8);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the reads are done before the next pass overwrites the stage)
      if constexpr (CODES) {
        uint32_t wq[4];
        eq.code4n_plain(v, wq);
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<uint32_t*>(stg + (it * 8 + rsel) * SROW + p * 32 + 4 * q4) = wq[it];
      }
    });
    // row-major out of the code stage: 8 lanes x 16 bytes per row, 8 rows per instruction (LDS operations of one wave execute in order)
    if constexpr (CODES) {
      const int vc = blk < a.nblk ? (blk * 32 + rsel) * a.K + n0 + q4 * 16 : BUF_BIG;
#pragma unroll
      for (int it = 0; it < NST; ++it) {
        const i32x4 c16 = *reinterpret_cast<const i32x4*>(stg + (it * 8 + rsel) * SROW + q4 * 16);
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(c16), "v"(vc), "s"(r_c), "s"(it * 8 * a.K) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the requests past the end: nothing may be in flight into registers at s_endpgm)
}

// where the kernel applies: a 1 x 1 block end with an fp32 shortcut and ReLU that emits its consumer's plain codes
bool conv_pwr_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                      int32_t dilation, const ConvEpi& ep, const float* out, const ConvSeg2* seg2) {
  if (R != 1 || S != 1 || pad != 0 || dilation != 1 || !ep.relu || ep.w_off) return false;
  if (ep.codes ? !epi_plain(ep) : !out) return false;            // (other quantisers: the tiled kernel)
  const int64_t P = (H - 1) / stride + 1, Q = (W - 1) / stride + 1;
  const int64_t M = N * P * Q;
  if (seg2) {
    // the dual form (a stage's first block: one addend is the block's last 1x1 convolution - 256 channels, read row by row -, the
    // other the 1x1 / stride-s convolution on the shortcut - 512 channels, sampled; either may be the call's first pair): ResNet-50's
    // 28^2 -> 14^2 block, whose two weight slices ((256 + 512) x 128 bytes) fit the LDS together; fp32 output and codes
    const ConvGeom& g2 = seg2->g;
    if (ep.residual || !out || !ep.codes || g2.R != 1 || g2.S != 1 || g2.pad != 0 || g2.dil != 1 || g2.P != P || g2.Q != Q) return false;
    const bool main_dense = stride == 1 && C == 256 && g2.C == 512, seg_dense = g2.stride == 1 && g2.C == 256 && C == 512;
    if (!main_dense && !seg_dense) return false;
    if ((main_dense ? N * g2.H * g2.W : N * H * W) * 512 >= (int64_t)BUF_BIG) return false;
  } else {
    if (stride != 1 || !ep.residual || !(C == 256 || C == 512)) return false;
    if (M * C >= (int64_t)BUF_BIG) return false;
  }
  if (K % PWR_BN != 0 || K > 4096) return false;
  // one layout for the call's fp32 tensors: with both a shortcut and an output the two bits must agree
  if (ep.residual && out && ((ep.ctl & DLMCQ_FP32_IN_CHUNK_MAJOR) != 0) != ((ep.ctl & DLMCQ_FP32_OUT_CHUNK_MAJOR) != 0)) return false;
  if ((ep.codes && !aligned16(ep.codes)) || (ep.residual && !aligned16(ep.residual)) || (out && !aligned16(out))) return false;
  if (M < 4096 || M % 32 != 0) return false;
  if (M * K * 4 >= (int64_t)BUF_BIG) return false;     // 32-bit buffer offsets
  return true;
}

template <int C, int NW, bool OUTF, bool CODES = true, int C2 = 0, bool SWP = false>
static int pwr_go(const PwrArgs& a0, const ConvEpi& ep, hipStream_t st) {
  constexpr int LDS = (C + C2) * PWR_BN + (C2 ? 6 : 3) * PWR_BN * 4 + NW * (32 * (PWR_BN + 16) + 32 * PWR_FROW * 4);
  static_assert(LDS <= 160 * 1024, "LDS");
  PwrArgs a = a0;
  a.nslice = a.K / PWR_BN;
  static int cus = 0;      // (one device per process: dlmc/_native.py)
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  int ngroups = (cus / a.nslice) & ~7;                            // one workgroup per CU; groups of `nslice` workgroups on one XCD share their pixels
  if (ngroups < 8) ngroups = 8;
  const int maxg = ((a.nblk + NW - 1) / NW + 7) & ~7;
  if (ngroups > maxg) ngroups = maxg;
  auto kern = conv_pwr_i8_kernel<C, NW, OUTF, CODES, C2, SWP>;
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((uint32_t)(ngroups * a.nslice)), dim3(NW * 64), LDS, st, a, ep);
  return launch_status();
}

int conv_pwr_launch(const int8_t* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum, const float* in_scale,
                    const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int32_t stride,
                    int shift, const ConvEpi& ep, hipStream_t st, const ConvSeg2* seg2) {
  const int64_t P = (H - 1) / stride + 1, Q = (W - 1) / stride + 1;
  PwrArgs a{};
  a.x = x; a.w = w; a.s_w = w_scale; a.wsum = wsum; a.bias = bias; a.s_in = in_scale; a.zp_in = in_zero_point;
  a.residual = ep.residual; a.out = out;
  a.M = (int)(N * P * Q); a.K = (int)K; a.shift = shift;
  a.nblk = a.M / 32;
  const bool fcm = (ep.ctl & (ep.residual ? DLMCQ_FP32_IN_CHUNK_MAJOR : DLMCQ_FP32_OUT_CHUNK_MAJOR)) != 0;
  a.f_rowb = fcm ? 256 : a.K * 4;
  a.f_chunk = fcm ? a.M * 256 : 256;
#ifndef DLMCQ_PWR_NW256
#define DLMCQ_PWR_NW256 12
#endif
#ifndef DLMCQ_PWR_NW512
#define DLMCQ_PWR_NW512 8
#endif
  constexpr int N2 = DLMCQ_PWR_NW256, N5 = DLMCQ_PWR_NW512;
  if (seg2) {
    const ConvGeom& g2 = seg2->g;
    a.P = (int)P; a.Q = (int)Q;
    a.qdiv = make_fastdiv((uint32_t)Q);
    a.pdiv = make_fastdiv((uint32_t)P);
    if (C == 256) {      // the call's first pair is the row-by-row one
      a.x2 = seg2->x; a.w2 = seg2->w; a.s_w2 = seg2->s_w; a.wsum2 = seg2->wsum; a.bias2 = seg2->bias; a.s_in2 = seg2->s_in; a.zp_in2 = seg2->zp_in;
      a.shift2 = seg2->shift; a.H2 = g2.H; a.W2 = g2.W; a.stride2 = g2.stride;
      return pwr_go<256, 6, true, true, 512, false>(a, ep, st);
    }
    // ... or the sampled one: the roles change places (the sum's operand order is kept: first pair + second pair)
    a.x = seg2->x; a.w = seg2->w; a.s_w = seg2->s_w; a.wsum = seg2->wsum; a.bias = seg2->bias; a.s_in = seg2->s_in; a.zp_in = seg2->zp_in;
    a.shift = seg2->shift;
    a.x2 = x; a.w2 = w; a.s_w2 = w_scale; a.wsum2 = wsum; a.bias2 = bias; a.s_in2 = in_scale; a.zp_in2 = in_zero_point;
    a.shift2 = shift; a.H2 = (int)H; a.W2 = (int)W; a.stride2 = stride;
    return pwr_go<256, 6, true, true, 512, true>(a, ep, st);
  }
  if (!ep.codes) return C == 256 ? pwr_go<256, N2, true, false>(a, ep, st) : (C == 512 ? pwr_go<512, N5, true, false>(a, ep, st) : DLMCQ_EINVAL);
  if (C == 256) return out ? pwr_go<256, N2, true>(a, ep, st) : pwr_go<256, N2, false>(a, ep, st);
  if (C == 512) return out ? pwr_go<512, N5, true>(a, ep, st) : pwr_go<512, N5, false>(a, ep, st);
  return DLMCQ_EINVAL;
}

}  // namespace dlmcq
