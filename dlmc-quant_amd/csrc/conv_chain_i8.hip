// A residual block's LAST 1x1 convolution (+ shortcut + ReLU + the consumer's quantiser) and the NEXT block's FIRST 1x1
// convolution (+ ReLU + its consumer's quantiser) as ONE kernel.  (FSPTQuant/base.py:95-159 twice, with the `out += identity;
// relu` of the model between them; same arithmetic, same order as conv_i8.hip's kernel run twice: bit-identical.)
//
// Why: the block-end layer is an HBM stream (4 B shortcut in + 4 B fp32 out per element for a 64..256-deep reduction, MFMA
// 3-5 % busy) and the activation codes it writes are read back exactly once, by the 1x1 reduction that follows, whose own
// time is all operand traffic.  Here a workgroup owns <= 64 pixels for ALL K_d output channels of the block end, walks
// them in chunks of 64 channels, and feeds each finished chunk of codes - still in LDS - to the second reduction as one K
// step.  The wide code tensor is never read (and, inside a stage, never written); the second layer's MFMA work runs in
// the shadow of the first layer's shortcut stream.
//
// Structure (256 threads = 2 x 2 waves: wave (wr, wc) owns rows wr*32.. and the column half wc):
//   * A (the block-end input, <= 64 rows x C1 <= 256 B) is loaded ONCE into registers as MFMA fragments;
//   * per chunk n: W1[n] (64 x C1), W3[:, n] (KB x 64) and the chunk's per-channel constants arrive by LDS-DMA into a
//     double buffer, the chunk's shortcut tile by asm buffer loads into registers - all requested one chunk ahead; one
//     counted s_waitcnt + one barrier, GEMM 1 (C1/32 MFMAs), epilogue 1 in registers (dequantise, 4x4 DPP transpose
//     from the accumulator layout to rows of 4 consecutive channels, + shortcut, ReLU, 16-byte buffer stores, quantise),
//     codes -> LDS, barrier, GEMM 2 step (KB/32 MFMAs into the resident second accumulator);
//   * buffer instructions with the row guard folded into the offset (out-of-range lanes are dropped by the hardware), so
//     every wave issues the same number of vector-memory instructions and the vmcnt arithmetic is exact.
#include "conv_i8_common.h"

namespace dlmcq {

struct ChainArgs {
  // GEMM 1: block end.  x codes [M][C1], w1 [KD][C1] int8, per-channel scale / code sum / bias [KD]
  const int8_t* x;
  const int8_t* w1;
  const float* s_w1;
  const int32_t* wsum1;
  const float* bias1;
  const float* s_in1;
  const float* zp_in1;
  int shift1;
  const float* residual;   // fp32 [M][KD] (required unless the shortcut is a convolution: C2 > 0)
  // the shortcut as a second 1x1 convolution into the same tile (C2 > 0; the block's downsample): x2 codes [N][H2][W2][C2]
  // sampled at (p * stride2, q * stride2), w2 [KD][C2]
  const int8_t* x2;
  const int8_t* w2;
  const float* s_w2;
  const int32_t* wsum2;
  const float* bias2;
  const float* s_in2;
  const float* zp_in2;
  int shift2, P, Q, H2, W2, stride2;
  FastDiv qdiv, pdiv;
  float* out;              // fp32 [M][KD] or null
  uint8_t* codes;          // [M][KD] or null
  // GEMM 2: the next 1x1 reduction.  w3 [KB][KD] int8; its input quantiser is ep1's (scale, zero point)
  const int8_t* w3;
  const float* s_w3;
  const int32_t* wsum3;
  const float* bias3;
  uint8_t* codes2;         // [M][KB]
  int w3_row, w3_chunk;    // byte strides of w3: from output channel to output channel, from 64-column chunk to chunk (KRSC: KD, 64; chunk-major: 64, KB * 64)
  int M, KD, rows_per_tile;
  // layout of the two fp32 tensors - f_: the shortcut read, o_: the block output written - as fp32 elements from row to row / bytes
  // from 64-channel chunk to chunk.  Row-major [M][KD]: KD, 256.  Chunk-major [KD / 64][M][64] (DLMCQ_FP32_*_CHUNK_MAJOR): 64, M * 256
  int f_rowq, f_cstep, o_rowq, o_cstep;
  int lab;                     // lab builds: 1 = no shortcut loads (timing only)
  unsigned long long* trace;   // lab builds: 64 clock-stamp slots per workgroup (null: none)
};

template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));   // (every lane reads a live lane: no `old` value to initialise)
}
// 4 x 4 transpose across the four lanes of a quad: afterwards x_j of lane b is what x_b of lane j was
__device__ __forceinline__ void quad_transpose(float& x0, float& x1, float& x2, float& x3, bool b0, bool b1) {
  const float x0s = quad_dpp<0xB1>(x0), x1s = quad_dpp<0xB1>(x1), x2s = quad_dpp<0xB1>(x2), x3s = quad_dpp<0xB1>(x3);
  const float n0 = b0 ? x1s : x0, n1 = b0 ? x1 : x0s, n2 = b0 ? x3s : x2, n3 = b0 ? x3 : x2s;
  const float n0s = quad_dpp<0x4E>(n0), n1s = quad_dpp<0x4E>(n1), n2s = quad_dpp<0x4E>(n2), n3s = quad_dpp<0x4E>(n3);
  x0 = b1 ? n2s : n0;
  x2 = b1 ? n2 : n0s;
  x1 = b1 ? n3s : n1;
  x3 = b1 ? n3 : n1s;
}

// workgroups per CU the kernel is compiled for (registers), by instantiation
constexpr int CHAIN_WGS(int c1, int kb) { return c1 + kb <= 256 ? 3 : 2; }
constexpr int CH_BIG = 0x7fff0000;   // a byte offset beyond every buffer this kernel accepts (< 2^31 - 64 KiB)

// FL (round 4): what the first epilogue does, known at compile time - bit 0: ReLU, bit 1: the fp32 output is stored; -1: both read
// from the arguments at run time (the general form; the plan's launches are all ReLU ones).  A run-time flag is a uniform branch per
// group of 4 values - five per group with the quantiser's, each a bubble in a wave's issue and a basic-block boundary the scheduler
// cannot move work across.
template <int C1, int KB, int C2 = 0, int FL = -1>
__global__ __launch_bounds__(256, CHAIN_WGS(C1 + C2, KB)) void conv_chain_i8_kernel(ChainArgs a, ConvEpi ep1, ConvEpi ep2) {
  constexpr bool DUALH = C2 > 0;         // the shortcut is a second convolution (no fp32 shortcut tensor)
  const bool relu1 = FL < 0 ? ep1.relu != 0 : (FL & 1) != 0;
  const bool out1 = FL < 0 ? a.out != nullptr : (FL & 2) != 0;
  constexpr int S1 = C1 / 64;            // K steps of GEMM 1
  constexpr int S2 = C2 / 64;            // K steps of the shortcut convolution
  constexpr int U3 = KB / 64;            // 64-row units of a W3 chunk = accumulator slabs of GEMM 2 per wave (KB/2 columns)
  constexpr int WCH = (S1 + S2 + U3) * 4096;  // bytes of one chunk's weights
  constexpr int NPD = DUALH ? 2 : 1;     // constant-table DMAs per wave per chunk
  constexpr int PAR = NPD * 4 * 256;     // per-channel constants of one chunk ((scale, code sum, bias) per pair) x 64 columns
  // How the accumulator layout (lane = channel, register = row) becomes rows of 4 consecutive channels per lane (TF):
  //   0  two DPP exchange rounds inside each quad (8 v_mov_dpp + 8 v_cndmask per 4 values).  The quad structure then dictates which
  //      lane owns which row: 4 adjacent lanes = 4 rows, so a 16-lane group of a shortcut load / output store touches 4 rows x 64 B;
  //   1  a wave-private 1 KB LDS stage per group of 8 rows (4 ds_write_b32 + 1 ds_read_b128; LDS operations of one wave execute in
  //      order, so neither a barrier nor a wait separates the writes from the read), same lane -> row mapping as 0: a quarter fewer
  //      vector instructions and NO gain (round 3: every chain within +-1 %) - the kernel is not bound by their count;
  //   2  the same stage read back with 8 adjacent lanes = one row's 128 bytes: a 16-lane group of a load / store now touches 2 WHOLE
  //      128-byte lines.  That is what the fp32 streams respond to (round 2's timing build with 256-byte rows said so): same box, in
  //      the plan, 56^2 chains -3 .. -7 %, 28^2 -3 .. -5 %, 14^2 -2 %; the 28^2 convolution-shortcut chain (no shortcut loads, LDS
  //      nearly full) +3 %, so it keeps form 0.  Bit-identical in all forms.
#ifndef DLMCQ_CHAIN_LDS_T
#define DLMCQ_CHAIN_LDS_T ((C1 == 128 && C2 == 256) ? 0 : 2)
#endif
  constexpr int TF = DLMCQ_CHAIN_LDS_T;
  constexpr bool LDST = TF != 0;
#ifdef DLMCQ_LAB
  // lab builds of the shapes with LDS to spare carry a 16 KB landing area for the timing-only variant `lab & 0x800` (round 5: the
  // shortcut tile by LDS-DMA instead of loads into registers - does the stream itself move differently?  56^2 mid-stage -1.7 %, 56^2
  // stage end +3.5 %: no.  LABNOTES 14)
  constexpr int LABDMA = (C2 == 0 && C1 == 64) ? 16384 : 0;
#else
  constexpr int LABDMA = 0;
#endif
  __shared__ __attribute__((aligned(1024))) int8_t lds[2 * WCH + 2 * PAR + 4096 + 3 * KB * 4 + (LDST ? 4096 : 0) + LABDMA];
  int8_t* const par0 = lds + 2 * WCH;
  int8_t* const ctile = lds + 2 * WCH + 2 * PAR;
  int8_t* const par3 = ctile + 4096;     // GEMM 2's per-channel constants, ready-made: s_in * s_w[k] | (128 - zp) * SUM qw[k] | bias[k]
  float* const tstage = reinterpret_cast<float*>(par3 + 3 * KB * 4) + (threadIdx.x >> 6) * 256;   // this wave's 8 rows x 32 channels

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  // the lane's place in the row-major (transposed) form of a group of 8 rows x 32 channels: row `rsel`, channels 4 q4 .. 4 q4 + 3
  constexpr bool LDST2 = TF == 2;
  const int q4 = LDST2 ? (lane & 7) : (l31 >> 2), b4 = l31 & 3;
  const int rsel = LDST2 ? (lane >> 3) : (4 * hsel + b4);
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  const int64_t row0 = (int64_t)blockIdx.x * a.rows_per_tile;
  const int rows_here = (int)((a.M - row0) < a.rows_per_tile ? (a.M - row0) : a.rows_per_tile);
  const int NC = a.KD >> 6;
#ifndef DLMCQ_CHAIN_ROT
#define DLMCQ_CHAIN_ROT 0
#endif
  // -DDLMCQ_CHAIN_ROT=1 (A/B builds; round 5): workgroup b walks its chunks starting at chunk b mod NC - are all workgroups on the same
  // quarter of the channels at once?  GEMM 2's integer sum does not care about the order: bit-identical, and +-0 (LABNOTES 16)
  const int rot = DLMCQ_CHAIN_ROT ? (int)(blockIdx.x % (unsigned)NC) : 0;
  auto cn = [&](int n) { const int m = n + rot; return m >= NC ? m - NC : m; };
#ifdef DLMCQ_LAB
  unsigned long long* const tr = (a.trace && tid == 0) ? a.trace + 64 * (size_t)blockIdx.x : nullptr;
  int trn = 0;
#define CHAIN_STAMP() do { if (tr && trn < 56) tr[trn++] = __builtin_readcyclecounter(); } while (0)
#define CHAIN_FINE(k) do { if (tr && nseq == 1) tr[56 + (k)] = __builtin_readcyclecounter(); } while (0)   // chunk 1 in detail (tools/chain_trace.py)
#else
#define CHAIN_STAMP() do { } while (0)
#define CHAIN_FINE(k) do { } while (0)
#endif
  CHAIN_STAMP();   // 0: start
#ifdef DLMCQ_LAB
  // timing only (round 5): are the chip's workgroups in lockstep?  Workgroups of the FIRST round start late - 0x200: the second
  // workgroup of a CU (blockIdx 256 .. 511) by `lab >> 16` units of 64 clocks; 0x400: every workgroup by a hashed 0 .. 15 sixteenths
  // of that (later rounds inherit the offsets: a slot's next tile starts when its last one ends).  Measured: -1 .. -4 % at best (LABNOTES 14)
  if ((a.lab & 0x600) && blockIdx.x < 512) {
    const int unit = a.lab >> 16;
    const int d = (a.lab & 0x200) ? ((blockIdx.x >> 8) & 1) * unit : (int)(((blockIdx.x * 2654435761u) >> 28) * (uint32_t)unit) >> 4;
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(1);
  }
#endif

  // ---- The tile's input rows (rows beyond the tile read its last row: never stored) go to LDS by LDS-DMA, 16 rows x 64 B per
  // wave-instruction, into buffer 1 of the weight double buffer - in W1's (W2's) own unit layout and swizzle, so the fragment
  // reads below are GEMM 1's weight-fragment reads with another base.  (Round 3: the fragments used to be loaded straight into
  // registers, 16 bytes per lane from 32 different rows per instruction - the costliest shape a vector-memory instruction
  // can have here - and twice over, once per column wave; in the short chains that was a quarter of a tile's requests.) ----
  {
    const int lrw = tid >> 2;                                 // = wave * 16 + (lane >> 2): the row of a 64-row unit this lane fetches
    const int sgw = ((tid & 3) ^ ((lrw >> 2) & 3)) * 16;
    const int8_t* xp = a.x + (row0 + (lrw < rows_here ? lrw : rows_here - 1)) * C1 + sgw;
#pragma unroll
    for (int s = 0; s < S1; ++s)
      __builtin_amdgcn_global_load_lds((gptr_t)(xp + s * 64), (lptr_t)(lds + WCH + s * 4096 + (tid >> 6) * 1024), 16, 0, 0);
    if constexpr (DUALH) {
      const uint32_t m = (uint32_t)(row0 + (lrw < rows_here ? lrw : rows_here - 1));
      const uint32_t t = fdiv(m, a.qdiv), nn = fdiv(t, a.pdiv);
      const int q = (int)(m - t * (uint32_t)a.Q), pp = (int)(t - nn * (uint32_t)a.P);
      const int8_t* xp2 = a.x2 + (((int64_t)nn * a.H2 + pp * a.stride2) * a.W2 + q * a.stride2) * C2 + sgw;
#pragma unroll
      for (int s = 0; s < S2; ++s)
        __builtin_amdgcn_global_load_lds((gptr_t)(xp2 + s * 64), (lptr_t)(lds + WCH + (S1 + s) * 4096 + (tid >> 6) * 1024), 16, 0, 0);
    }
  }
  // GEMM 2's epilogue constants, once per workgroup (its accumulator is kept with the operands swapped - weights as A, the
  // code tile as B - so a lane owns 16 consecutive channels of one pixel and reads their constants as broadcast ds_read_b128)
  const float sin2 = ep1.q_scale[0];                       // GEMM 2's input scale IS the quantiser the codes were made with
  const float zpf2 = ep1.q_zp ? ep1.q_zp[0] : 0.0f;
  const int dz2 = 128 - (int)__builtin_rintf(zpf2);
  if (tid < KB) {
    reinterpret_cast<float*>(par3)[tid] = sin2 * a.s_w3[tid];
    reinterpret_cast<int*>(par3)[KB + tid] = dz2 * a.wsum3[tid];
    reinterpret_cast<float*>(par3)[2 * KB + tid] = a.bias3 ? a.bias3[tid] : 0.0f;
  }
  // the input tile has landed, and no compiler-known load is outstanding from here on (the counted waits below assume it)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  i32x4 af[S1][2];
  i32x4 af2[DUALH ? S2 : 1][2];
  {
    const int r = wr * 32 + l31;
    const int8_t* ab = lds + WCH + r * 64;
    const uint32_t xw = a.shift1 ? 0x80808080u : 0u;
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const i32x4 t = *reinterpret_cast<const i32x4*>(ab + s * 4096 + (((ks * 2 + hsel) ^ ((r >> 2) & 3)) << 4));
        af[s][ks] = i32x4{(int)(t.x ^ xw), (int)(t.y ^ xw), (int)(t.z ^ xw), (int)(t.w ^ xw)};
      }
    if constexpr (DUALH) {
      const uint32_t xw2 = a.shift2 ? 0x80808080u : 0u;
#pragma unroll
      for (int s = 0; s < S2; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const i32x4 t = *reinterpret_cast<const i32x4*>(ab + (S1 + s) * 4096 + (((ks * 2 + hsel) ^ ((r >> 2) & 3)) << 4));
          af2[s][ks] = i32x4{(int)(t.x ^ xw2), (int)(t.y ^ xw2), (int)(t.z ^ xw2), (int)(t.w ^ xw2)};
        }
    }
  }
  // (the fragments are in registers before this wave reaches chunk 0's barrier, behind which buffer 1 is requested for chunk 1)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  CHAIN_STAMP();   // 1: A fragments in registers

  // ---- addressing of the fp32 tile in the transposed (row-major) layout: group g -> row wr*32 + 8g + 4 hsel + b4 ----
  const uint32_t fbytes = (uint32_t)((int64_t)a.M * a.KD * 4);
  const v4i r_res = make_rsrc(a.residual, fbytes);
  const v4i r_out = make_rsrc(a.out ? (const void*)a.out : (const void*)a.residual, a.out ? fbytes : 0u);
  const v4i r_cod = make_rsrc(a.codes ? (const void*)a.codes : (const void*)a.residual, a.codes ? fbytes / 4 : 0u);
  int fo[4];     // byte offset of this lane's 16 bytes in chunk 0, per group (CH_BIG: row not in the tile)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int lr = wr * 32 + 8 * g + rsel;
    fo[g] = lr < rows_here ? (int)(((row0 + lr) * a.f_rowq + wc * 32 + q4 * 4) * 4) : CH_BIG;
#ifdef DLMCQ_LAB
    if (a.lab & 4) {   // timing only: the access pattern a 16 x 16 x 64 accumulator layout would have - 16 rows x 64 bytes per instruction
      const int lr2 = wr * 32 + 16 * (g >> 1) + 2 * (4 * hsel + b4) + (q4 >> 2);
      fo[g] = lr2 < rows_here ? (int)(((row0 + lr2) * a.KD + wc * 32 + 16 * (g & 1) + (q4 & 3) * 4) * 4) : CH_BIG;
    }
    if (a.lab & 8) {   // timing only: 4 rows x 256 bytes per instruction (a wave owning 16 rows x 64 channels)
      const int lr3 = wr * 32 + wc * 16 + 4 * g + (lane >> 4);
      fo[g] = lr3 < rows_here ? (int)(((row0 + lr3) * a.KD + (lane & 15) * 4) * 4) : CH_BIG;
    }
#endif
  }
  int cstep = a.f_cstep; // byte distance between a row's consecutive 64-channel chunks
#ifdef DLMCQ_LAB
  if (a.lab & 64) {      // timing only: the fp32 tensors in 64 x 64 blocks (every chunk of a tile one contiguous 16 KB)
    cstep = 64 * 256;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int lr = wr * 32 + 8 * g + rsel;
      fo[g] = lr < rows_here ? (int)(row0 * a.KD * 4 + lr * 256 + (wc * 32 + q4 * 4) * 4) : CH_BIG;
    }
  }
#endif
  // the output's offsets: the shortcut's, unless the two tensors differ in layout (one row-major, one chunk-major: a chain between a
  // kernel that does not know the chunk-major form and one that does).  The instantiation that sits at its register limit has no room
  // for a second set: chain_launch refuses mixed layouts for it.
  constexpr bool MIXED_OK = !DUALH && !(C1 == 128 && KB == 128);
  int fo2[MIXED_OK ? 4 : 1];
  const int cstep2 = a.o_cstep;
  if constexpr (MIXED_OK) {
    const bool same = a.o_rowq == a.f_rowq;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int lr = wr * 32 + 8 * g + rsel;
      fo2[g] = same ? fo[g] : (lr < rows_here ? (int)(((row0 + lr) * a.o_rowq + wc * 32 + q4 * 4) * 4) : CH_BIG);
    }
  }
  auto out_off = [&](int g, int n) {
    if constexpr (MIXED_OK) return fo2[g] + n * cstep2;
    else return fo[g] + n * cstep;
  };
  // transposition stage: row 4 hsel + j of a group holds this lane's register 4 g + j at column l31; a lane reads back row
  // 4 hsel + b4, columns 4 q4 .. + 3; 4-column slot c of row r sits at slot c ^ (r & 7)
  int tw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tw[j] = (4 * hsel + j) * 32 + (LDST2 ? l31 : ((l31 & 3) | ((l31 & ~3) ^ ((4 * hsel + j) * 4))));
  const int trd = LDST2 ? rsel * 32 + 4 * q4 : (4 * hsel + b4) * 32 + ((4 * q4) ^ ((4 * hsel + b4) * 4));   // (form 2: 8 lanes read one row = all 32 banks: no swizzle needed)
  const int nst = (out1 ? 4 : 0) + (a.codes ? 1 : 0);   // stores per wave per chunk

  // ---- DMA sources.  A wave-instruction lands 16 rows x 64 B; row r of a unit keeps its 16-byte segments XOR-swizzled ----
  const int lrow = lane >> 2, pslot = lane & 3;
  const int drow = wave * 16 + lrow;                       // row of a 64-row unit this lane fetches
  const int dseg = (pslot ^ ((drow >> 2) & 3)) * 16;
  const int8_t* w1p = a.w1 + (int64_t)drow * C1 + dseg;    // + chunk * 64 * C1 + s * 64
  // W3's rows are dealt to the LDS rows so that accumulator register i of GEMM 2 (weights as the A operand) is channel
  // 16 hsel + i of its 32-channel block: LDS row d of a block holds channel 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3)
  const int d3 = drow & 31;
  const int prow3 = (drow & 32) + 16 * ((d3 >> 2) & 1) + 4 * (d3 >> 3) + (d3 & 3);
  const int8_t* w3p = a.w3 + (int64_t)prow3 * a.w3_row + dseg;  // + unit * 64 * w3_row + chunk * w3_chunk
  const int cvo = drow < rows_here ? (int)((row0 + drow) * a.KD + dseg) : CH_BIG;   // this lane's 16 bytes of the code tile, chunk 0
  const void* const pars[4] = {a.s_w1, a.wsum1, a.bias1 ? (const void*)a.bias1 : (const void*)a.s_w1, a.s_w1};
  const int32_t* parp = static_cast<const int32_t*>(wave == 0 ? pars[0] : wave == 1 ? pars[1] : wave == 2 ? pars[2] : pars[3]) + lane;
  const int8_t* w2p = DUALH ? a.w2 + (int64_t)drow * C2 + dseg : nullptr;
  const void* const pars2[4] = {a.s_w2, a.wsum2, a.bias2 ? (const void*)a.bias2 : (const void*)a.s_w2, a.s_w2};
  const int32_t* parp2 = static_cast<const int32_t*>(wave == 0 ? pars2[0] : wave == 1 ? pars2[1] : wave == 2 ? pars2[2] : pars2[3]) + lane;

  f32x4 res[2][4];
#ifdef DLMCQ_LAB
  if (a.lab & 0x801)
    for (auto& r2 : res)
      for (auto& r : r2) r = f32x4{1.0f, 2.0f, 3.0f, 4.0f};
#endif
  auto request = [&](int nseq, auto par_c) {    // everything the nseq-th chunk of this workgroup's walk needs from memory
    constexpr int P = decltype(par_c)::value;
    const int n = cn(nseq);
    int8_t* wb = lds + P * WCH;
    bool wdma = true;
#ifdef DLMCQ_LAB
    // timing only: no weight DMA on odd chunks (16: what a tile of twice the pixels would request per output) / at all (32)
    wdma = !(((a.lab & 16) && (n & 1)) || (a.lab & 32));
#endif
#pragma unroll
    for (int s = 0; s < S1; ++s)
      if (wdma) __builtin_amdgcn_global_load_lds((gptr_t)(w1p + (int64_t)n * 64 * C1 + s * 64), (lptr_t)(wb + s * 4096 + wave * 1024), 16, 0, 0);
#pragma unroll
    for (int u = 0; u < U3; ++u)
      if (wdma) {
        const int8_t* src = w3p + (int64_t)u * 64 * a.w3_row + (int64_t)n * a.w3_chunk;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(wb + (S1 + S2 + u) * 4096 + wave * 1024), 16, 0, 0);
      }
    __builtin_amdgcn_global_load_lds((gptr_t)(parp + n * 64), (lptr_t)(par0 + P * PAR + wave * 256), 4, 0, 0);
    if constexpr (DUALH) {
#pragma unroll
      for (int s = 0; s < S2; ++s)
        if (wdma) __builtin_amdgcn_global_load_lds((gptr_t)(w2p + (int64_t)n * 64 * C2 + s * 64), (lptr_t)(wb + (S1 + s) * 4096 + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(parp2 + n * 64), (lptr_t)(par0 + P * PAR + 1024 + wave * 256), 4, 0, 0);
      return;      // no fp32 shortcut tile
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#ifdef DLMCQ_LAB
      if (a.lab & 1) continue;
      if constexpr (LABDMA > 0) {
        if (a.lab & 0x800) {    // timing only: the same 16 bytes per lane, by LDS-DMA (nt) into the landing area; `res` keeps stale values
          const int lr = wr * 32 + 8 * g + rsel;
          const float* src = a.residual + (row0 + (lr < rows_here ? lr : rows_here - 1)) * a.KD + n * 64 + wc * 32 + q4 * 4;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + sizeof(lds) - LABDMA + wave * 4096 + g * 1024), 16, 0, 2);
          continue;
        }
      }
#endif
      int lo = fo[g] + n * cstep;
#ifdef DLMCQ_LAB
      if (a.lab & 0x2000) lo &= 0xffff;
#endif
      bload16(res[P][g], lo, r_res);
    }
  };

  const float sin1 = a.s_in1[0];
  const float zpf1 = a.zp_in1 ? a.zp_in1[0] : 0.0f;
  const int dz1 = a.shift1 - (int)__builtin_rintf(zpf1);
  const float sin1b = DUALH ? a.s_in2[0] : 0.0f;
  const int dz1b = DUALH ? a.shift2 - (int)__builtin_rintf(a.zp_in2 ? a.zp_in2[0] : 0.0f) : 0;
  ConvEpi e1 = ep1;
  e1.codes = reinterpret_cast<uint8_t*>(uintptr_t(1));   // the quantiser is always needed (GEMM 2 reads its codes)
  // FL >= 0 also says: both quantisers are the plain unsigned-byte one (epi_plain_q; the launcher checks) - code4n_plain, whose
  // saturating pack is the ReLU as far as the codes are concerned
  const EpiQuant eq1(e1, FL >= 0);
  __builtin_assume(!eq1.sgn);                            // (the range is [0, 255]: chain_launch refuses anything else)

  i32x16 acc2[U3];
#pragma unroll
  for (int j = 0; j < U3; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[j][i] = 0;

  request(0, std::integral_constant<int, 0>{});

  auto chunk = [&](int nseq, auto par_c) {
    constexpr int P = decltype(par_c)::value;
    const int n = cn(nseq);
    // chunk n's requests have landed; the stores of chunk n-1 (younger) stay in flight.  Then the barrier: everyone's DMA pieces are
    // visible; everyone left GEMM 2 of chunk n-1.  Wait and barrier are ONE asm statement with a memory clobber, so no LDS access can
    // be scheduled between them or moved across the barrier (a bare __builtin_amdgcn_s_barrier() does not touch memory for the compiler)
    if (nseq == 0 || nst == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else if (nst == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else if (nst == 5) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
    CHAIN_STAMP();   // 2 + 3n: chunk n's operands are there
    if constexpr (!DUALH) asm volatile("" : "+v"(res[P][0]), "+v"(res[P][1]), "+v"(res[P][2]), "+v"(res[P][3]));
    if (nseq + 1 < NC) request(nseq + 1, std::integral_constant<int, 1 - P>{});

    const int8_t* wb = lds + P * WCH;
    const int8_t* pp = par0 + P * PAR + (wc * 32 + l31) * 4;
    i32x16 acc;
    // (round 5: the accumulators start from ZERO - the matrix instruction's inline constant, no register moves - and the column's constant
    //  (shift - zp) * SUM qw joins in fp32: |sum| <= 256 * 128 * 127 and |constant| <= 128 * 256 * 127, so both conversions and their sum are
    //  exact (< 2^24) and fma(float(sum) + float(constant), m, b) is fma(float(sum + constant), m, b) bit for bit; the chain runs on register
    //  PAIRS - v_pk_add_f32 / v_pk_fma_f32 -: 2 instructions per element where the integer form had 3)
    f32x2 extra2[DUALH ? 8 : 1];
    if constexpr (DUALH) {   // the shortcut convolution first: dequantised, it waits in registers for the block's own sum
      const float corr2f = (float)(dz1b * *reinterpret_cast<const int*>(pp + 1024 + 256));
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0;
      const int r = wc * 32 + l31;
#pragma unroll
      for (int s = 0; s < S2; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const i32x4 bf = *reinterpret_cast<const i32x4*>(wb + (S1 + s) * 4096 + r * 64 + (((ks * 2 + hsel) ^ ((r >> 2) & 3)) << 4));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af2[s][ks], bf, acc, 0, 0, 0);
        }
      const float mult2 = sin1b * *reinterpret_cast<const float*>(pp + 1024);
      const float bv2 = a.bias2 ? *reinterpret_cast<const float*>(pp + 1024 + 512) : 0.0f;
      const f32x2 c2{corr2f, corr2f}, m2{mult2, mult2}, b2{bv2, bv2};
#pragma unroll
      for (int i = 0; i < 8; ++i) extra2[i] = pk_fma(f32x2{(float)acc[2 * i], (float)acc[2 * i + 1]} + c2, m2, b2);
    }
    // ---- GEMM 1: rows wr*32.., columns n*64 + wc*32.. ----
    const float corrf = (float)(dz1 * *reinterpret_cast<const int*>(pp + 256));
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    {
      const int r = wc * 32 + l31;
#pragma unroll
      for (int s = 0; s < S1; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const i32x4 bf = *reinterpret_cast<const i32x4*>(wb + s * 4096 + r * 64 + (((ks * 2 + hsel) ^ ((r >> 2) & 3)) << 4));
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s][ks], bf, acc, 0, 0, 0);
        }
    }
#ifdef DLMCQ_LAB
    if (tr && nseq == 1) asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[15]));   // (GEMM 1 retired)
#endif
    CHAIN_FINE(0);
    // ---- epilogue 1 ----
    const float mult = sin1 * *reinterpret_cast<const float*>(pp);
    const float bv = a.bias1 ? *reinterpret_cast<const float*>(pp + 512) : 0.0f;
    float v[16];
    {
      const f32x2 c2{corrf, corrf}, m2{mult, mult}, b2{bv, bv};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x2 f = pk_fma(f32x2{(float)acc[2 * i], (float)acc[2 * i + 1]} + c2, m2, b2);
        if constexpr (DUALH) f = f + extra2[DUALH ? i : 0];     // `out += identity`, the identity being a convolution
        v[2 * i] = f.x;
        v[2 * i + 1] = f.y;
      }
    }
    // GP groups of 8 rows share ONE tie branch of the quantiser (code4n): with the flags above known at compile time that branch is the
    // only basic-block boundary left in the epilogue, and the scheduler can run a group's LDS round trip under its neighbour's arithmetic
#ifndef DLMCQ_CHAIN_GP
#define DLMCQ_CHAIN_GP 4
#endif
    // (the two instantiations that sit at the 168 registers of three workgroups per CU keep the branch per group: more spills otherwise)
    constexpr bool TIGHT = C2 == 0 && ((C1 == 64 && KB == 128) || (C1 == 128 && KB == 128));
    constexpr int GP = (FL < 0 || TIGHT) ? 1 : DLMCQ_CHAIN_GP;
#pragma unroll
    for (int g0 = 0; g0 < 4; g0 += GP) {
      f32x4 y[GP];
#pragma unroll
      for (int k = 0; k < GP; ++k) {
        const int g = g0 + k;
        if constexpr (LDST) {
#pragma unroll
          for (int j = 0; j < 4; ++j) tstage[tw[j]] = v[4 * g + j];
          // (the read takes what OTHER lanes of this wave wrote: LDS operations of a wave execute in order, and the compiler must
          //  keep them in this order - a memory clobber between the writes and the read, and after the read, costs no instruction)
          asm volatile("" ::: "memory");
          y[k] = *reinterpret_cast<const f32x4*>(tstage + trd);
          asm volatile("" ::: "memory");
        } else {
          quad_transpose(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3], b0, b1);
          y[k] = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        }
        if constexpr (!DUALH) y[k] = y[k] + res[P][g];
        if (relu1 && (FL < 0 || out1)) y[k] = relu4_nan(y[k]);      // (plain quantiser: only the fp32 output needs the rectified value)
#ifdef DLMCQ_LAB
        // timing only (round 5): 0x1000 - the fp32 stores land in a 64 KB window of the output (they stay in L2: what do the STORE INSTRUCTIONS cost
        // without their HBM traffic?); 0x2000 - the same for the shortcut loads (below)
        if (out1) bstore16(y[k], (a.lab & 0x1000) ? (out_off(g, n) & 0xffff) : out_off(g, n), r_out);
#else
        if (out1) bstore16(y[k], out_off(g, n), r_out);
#endif
      }
      uint32_t c[GP];
      if constexpr (FL >= 0) {
        eq1.code4n_plain(y, c);
      } else if constexpr (GP == 1) {
        c[0] = eq1.code4(y[0]);
      } else {
        bool un[GP];
        eq1.code4n(y, c, un);
      }
#pragma unroll
      for (int k = 0; k < GP; ++k) {
        const int R = wr * 32 + 8 * (g0 + k) + rsel;
        *reinterpret_cast<uint32_t*>(ctile + R * 64 + (((wc * 2 + (q4 >> 2)) ^ ((R >> 2) & 3)) << 4) + (q4 & 3) * 4) = c[k];
        CHAIN_FINE(1 + g0 + k);
      }
    }
    CHAIN_STAMP();   // 3 + 3n: GEMM 1 + epilogue 1 issued
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the 64 x 64 code tile is complete
    CHAIN_STAMP();   // 4 + 3n: past the mid-chunk barrier
    if (a.codes)     // stage ends: the tile goes out 64 contiguous bytes per pixel (slot s of row r holds segment s ^ (r >> 2 & 3))
      bstore16i(*reinterpret_cast<const i32x4*>(ctile + drow * 64 + pslot * 16), cvo == CH_BIG ? CH_BIG : cvo + n * 64, r_cod);
    // ---- GEMM 2: one K step (this chunk's 64 channels) into the resident accumulator ----
    {
      const int R = wr * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const i32x4 t = *reinterpret_cast<const i32x4*>(ctile + R * 64 + (((ks * 2 + hsel) ^ ((R >> 2) & 3)) << 4));
        const i32x4 a2 = i32x4{(int)(t.x ^ 0x80808080u), (int)(t.y ^ 0x80808080u), (int)(t.z ^ 0x80808080u), (int)(t.w ^ 0x80808080u)};
#pragma unroll
        for (int j = 0; j < U3; ++j) {
          const int kb = wc * (KB / 2) + j * 32 + l31;
          const i32x4 bf = *reinterpret_cast<const i32x4*>(wb + (S1 + S2) * 4096 + kb * 64 + (((ks * 2 + hsel) ^ ((kb >> 2) & 3)) << 4));
          acc2[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, a2, acc2[j], 0, 0, 0);
        }
      }
    }
  };
  for (int n = 0; n < NC; n += 2) {
    chunk(n, std::integral_constant<int, 0>{});
    if (n + 1 < NC) chunk(n + 1, std::integral_constant<int, 1>{});
  }

  CHAIN_STAMP();   // chunks done
  // ---- epilogue 2: the reduction layer's own dequantise, ReLU, its consumer's quantiser; codes only.  Lane (p = l31, hsel) holds
  // channels wc * KB/2 + 32 j + 16 hsel + 0..15 of pixel wr * 32 + p: no transposition, the ReLU folded into the quantiser's clamp
  // (code(relu(v)) = max(code(v), code(0))), 16 finished bytes per lane and block ----
  const EpiQuant eq2(ep2, ep2.relu != 0);
  const int lr2 = wr * 32 + l31;
#pragma unroll
  for (int j = 0; j < U3; ++j) {
    const int cb = wc * (KB / 2) + j * 32 + hsel * 16;
    f32x4 y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 mu = *reinterpret_cast<const f32x4*>(par3 + (cb + 4 * q) * 4);
      const i32x4 co = *reinterpret_cast<const i32x4*>(par3 + (KB + cb + 4 * q) * 4);
      const f32x4 bs = *reinterpret_cast<const f32x4*>(par3 + (2 * KB + cb + 4 * q) * 4);
      y[q] = f32x4{dequant1(acc2[j][4 * q] + co.x, mu.x, bs.x), dequant1(acc2[j][4 * q + 1] + co.y, mu.y, bs.y),
                   dequant1(acc2[j][4 * q + 2] + co.z, mu.z, bs.z), dequant1(acc2[j][4 * q + 3] + co.w, mu.w, bs.w)};
    }
    uint32_t wq[4];
    if constexpr (FL >= 0) {
      eq2.code4n_plain(y, wq);
    } else {
      bool uq[4];
      eq2.code4n(y, wq, uq);
    }
    if (lr2 < rows_here)
      *reinterpret_cast<i32x4*>(ep2.codes + (row0 + lr2) * KB + cb) = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
  }
#ifdef DLMCQ_LAB
  if (tr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CHAIN_STAMP();   // end
    tr[63] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_ID
  }
#endif
}

}  // namespace dlmcq

using namespace dlmcq;

#ifdef DLMCQ_LAB
static unsigned long long* g_chain_trace = nullptr;   // set by dlmcq_x_chain_trace (tools/chain_trace.py)
extern "C" void dlmcq_x_chain_trace(void* buf) { g_chain_trace = static_cast<unsigned long long*>(buf); }
static int g_chain_lab = 0;
extern "C" void dlmcq_x_chain_lab(int flags) { g_chain_lab = flags; }
#endif

static int chain_launch(ChainArgs& a, int64_t M, int64_t C, int64_t K, int64_t C2, int64_t K2, int32_t relu, void* codes,
                        const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g,
                        int32_t relu2, void* codes2, const float* q2_scale, const float* q2_zero_point, int32_t q2_lo, int32_t q2_hi,
                        int32_t q2_form, float q2_ste_g, int32_t rows_per_tile, dlmcq_stream_t stream) {
  if (q_lo != 0 || q_hi != 255) return DLMCQ_EINVAL;   // GEMM 2 reads the codes as uint8 (shift 128)
  const bool w3cm = (q2_form & DLMCQ_W2_CHUNK_MAJOR) != 0, ocm = (q2_form & DLMCQ_FP32_OUT_CHUNK_MAJOR) != 0;
  // (a call without a shortcut tensor - the convolution-shortcut form - or without an output has ONE fp32 tensor: its layout is the call's)
  const bool icm = (C2 == 0 && a.residual) ? (q2_form & DLMCQ_FP32_IN_CHUNK_MAJOR) != 0 : ocm;
  const bool ocm2 = a.out ? ocm : icm;
  q2_form &= ~(DLMCQ_W2_CHUNK_MAJOR | DLMCQ_FP32_IN_CHUNK_MAJOR | DLMCQ_FP32_OUT_CHUNK_MAJOR);
  if (icm != ocm2 && C2 == 0 && C == 128 && K2 == 128) return DLMCQ_EINVAL;   // (no registers for two offset sets there: conv_chain_i8_kernel)
  a.w3_row = w3cm ? 64 : (int)K;
  a.w3_chunk = w3cm ? (int)K2 * 64 : 64;
  ConvEpi ep1{}, ep2{};
  if (q2_lo > q2_hi || q2_lo < -128 || q2_hi > 255 || q2_hi - q2_lo > 255 || q_form < DLMCQ_FORM_EMULATE ||
      q_form > DLMCQ_FORM_SYMMETRIC || !epi_set_form(ep2, q2_form, q2_lo, q2_hi))
    return DLMCQ_EINVAL;
  if (M * K * 4 > (int64_t)CH_BIG) return DLMCQ_ERANGE;   // 32-bit buffer offsets
  a.M = (int)M; a.KD = (int)K;
  a.f_rowq = icm ? 64 : (int)K;
  a.f_cstep = icm ? (int)M * 256 : 256;
  a.o_rowq = ocm2 ? 64 : (int)K;
  a.o_cstep = ocm2 ? (int)M * 256 : 256;
#ifdef DLMCQ_LAB
  a.trace = g_chain_trace;
  a.lab = g_chain_lab;
#endif
  // Tile height: 64 rows.  A chunk costs a workgroup the same time at 32 .. 64 rows (round 5, tools/chain_ab.py --rows: a 14^2 launch takes
  // 238 / 218 / 241 / 249 / 258 / 323 us at 64 / 56 / 49 / 48 / 40 / 32 rows, every other shape is fastest at 64), so shorter tiles only pay
  // where they keep a nearly empty last round from costing a whole tile's life: ResNet-50's 14^2 chains at batch 512 - 1 568 tiles on the
  // 512 slots of the two-workgroup instantiations, 3.06 rounds - run 3.5 rounds of 56-row tiles 8 % faster.  The rule: few rounds, the last
  // one less than an eighth full.
  if (rows_per_tile <= 0) {
    const int64_t t64 = (M + 63) / 64, slots = 256 * CHAIN_WGS((int)(C + C2), (int)K2);
    const int64_t full = t64 / slots, left = t64 % slots;
    rows_per_tile = (full >= 2 && full <= 4 && left > 0 && left * 8 < slots) ? 56 : 64;
  }
  a.rows_per_tile = rows_per_tile;
  if (a.rows_per_tile > 64) return DLMCQ_EINVAL;
  ep1.relu = relu != 0; ep1.q_scale = q_scale; ep1.q_zp = q_zero_point; ep1.q_lo = (float)q_lo; ep1.q_hi = (float)q_hi;
  ep1.q_g = q_ste_g; ep1.q_form = q_form; ep1.codes = static_cast<uint8_t*>(codes);
  ep2.relu = relu2 != 0; ep2.q_scale = q2_scale; ep2.q_zp = q2_zero_point; ep2.q_lo = (float)q2_lo; ep2.q_hi = (float)q2_hi;
  ep2.q_g = q2_ste_g; ep2.codes = static_cast<uint8_t*>(codes2);
  const int64_t tiles = (M + a.rows_per_tile - 1) / a.rows_per_tile;
  if (tiles >= (1ll << 31)) return DLMCQ_ERANGE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((uint32_t)tiles), block(256);
  size_t dyn = 0;
#ifdef DLMCQ_LAB
  dyn = (g_chain_lab & 128 ? 40960 : 0) + (g_chain_lab & 256 ? 81920 : 0);   // timing only: unused dynamic LDS = fewer workgroups per CU
#endif
  // the plan's launches end their first layer with a ReLU: those get the instantiation that knows its flags at compile time
  const int fl = (relu && epi_plain_q(ep1) && epi_plain_q(ep2)) ? (a.out ? 3 : 1) : -1;
#define DLMCQ_CHAIN_GO(...)                                                                                          \
  do {                                                                                                               \
    if (fl == 3) hipLaunchKernelGGL((conv_chain_i8_kernel<__VA_ARGS__, 3>), grid, block, dyn, st, a, ep1, ep2);       \
    else if (fl == 1) hipLaunchKernelGGL((conv_chain_i8_kernel<__VA_ARGS__, 1>), grid, block, dyn, st, a, ep1, ep2);  \
    else hipLaunchKernelGGL((conv_chain_i8_kernel<__VA_ARGS__, -1>), grid, block, dyn, st, a, ep1, ep2);              \
  } while (0)
  if (C2 == 0) {
    if (C == 64 && K2 == 64) DLMCQ_CHAIN_GO(64, 64, 0);
    else if (C == 64 && K2 == 128) DLMCQ_CHAIN_GO(64, 128, 0);
    else if (C == 128 && K2 == 128) DLMCQ_CHAIN_GO(128, 128, 0);
    else if (C == 128 && K2 == 256) DLMCQ_CHAIN_GO(128, 256, 0);
    else if (C == 256 && K2 == 256) DLMCQ_CHAIN_GO(256, 256, 0);
    else return DLMCQ_EINVAL;
  } else {
    if (C == 64 && C2 == 64 && K2 == 64) DLMCQ_CHAIN_GO(64, 64, 64);
    else if (C == 128 && C2 == 256 && K2 == 128) DLMCQ_CHAIN_GO(128, 128, 256);
    else return DLMCQ_EINVAL;
  }
#undef DLMCQ_CHAIN_GO
  return launch_status();
}

extern "C" int dlmcq_conv2d_i8_nhwc_chain(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                          const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t M,
                                          int64_t C, int64_t K, int32_t x_is_unsigned, const float* residual, int32_t relu,
                                          void* codes, const float* q_scale, const float* q_zero_point, int32_t q_lo,
                                          int32_t q_hi, int32_t q_form, float q_ste_g, const int8_t* w2, const float* bias2,
                                          const int32_t* wsum2, const float* w_scale2, int64_t K2, int32_t relu2, void* codes2,
                                          const float* q2_scale, const float* q2_zero_point, int32_t q2_lo, int32_t q2_hi,
                                          int32_t q2_form, float q2_ste_g, int32_t rows_per_tile, dlmcq_stream_t stream) {
  if (M < 0 || C < 1 || K < 1 || K2 < 1 || K % 64 != 0) return DLMCQ_EINVAL;
  if (M == 0) return DLMCQ_OK;
  if (!x || !w || !wsum || !in_scale || !w_scale || !residual || !w2 || !wsum2 || !w_scale2 || !codes2 || !q_scale || !q2_scale)
    return DLMCQ_EINVAL;
  if (!aligned16(x) || !aligned16(w) || !aligned16(w2) || !aligned16(residual) || (out && !aligned16(out)) ||
      (codes && !aligned16(codes)) || !aligned16(codes2))
    return DLMCQ_EALIGN;
  ChainArgs a{};
  a.x = static_cast<const int8_t*>(x); a.w1 = w; a.s_w1 = w_scale; a.wsum1 = wsum; a.bias1 = bias;
  a.s_in1 = in_scale; a.zp_in1 = in_zero_point; a.shift1 = x_is_unsigned ? 128 : 0;
  a.residual = residual; a.out = out; a.codes = static_cast<uint8_t*>(codes);
  a.w3 = w2; a.s_w3 = w_scale2; a.wsum3 = wsum2; a.bias3 = bias2; a.codes2 = static_cast<uint8_t*>(codes2);
  return chain_launch(a, M, C, K, 0, K2, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g, relu2, codes2, q2_scale,
                      q2_zero_point, q2_lo, q2_hi, q2_form, q2_ste_g, rows_per_tile, stream);
}

extern "C" int dlmcq_conv2d_i8_nhwc_dual_chain(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                               const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                                               int64_t H, int64_t W, int64_t C, int64_t K, int32_t x_is_unsigned, const void* x2,
                                               const int8_t* w2, const float* bias2, const int32_t* wsum2, const float* in_scale2,
                                               const float* in_zero_point2, const float* w_scale2, int64_t H2, int64_t W2,
                                               int64_t C2, int32_t stride2, int32_t x2_is_unsigned, int32_t relu, void* codes,
                                               const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                                               int32_t q_form, float q_ste_g, const int8_t* w3, const float* bias3,
                                               const int32_t* wsum3, const float* w_scale3, int64_t K3, int32_t relu3, void* codes3,
                                               const float* q3_scale, const float* q3_zero_point, int32_t q3_lo, int32_t q3_hi,
                                               int32_t q3_form, float q3_ste_g, int32_t rows_per_tile, dlmcq_stream_t stream) {
  if (N < 0 || H < 1 || W < 1 || C < 1 || K < 1 || K3 < 1 || K % 64 != 0 || H2 < 1 || W2 < 1 || C2 < 1 || stride2 < 1)
    return DLMCQ_EINVAL;
  if ((H2 - 1) / stride2 + 1 != H || (W2 - 1) / stride2 + 1 != W) return DLMCQ_EINVAL;   // both pairs give [N, H, W, K]
  const int64_t M = N * H * W;
  if (M == 0) return DLMCQ_OK;
  if (!x || !w || !wsum || !in_scale || !w_scale || !x2 || !w2 || !wsum2 || !in_scale2 || !w_scale2 || !w3 || !wsum3 || !w_scale3 ||
      !codes3 || !q_scale || !q3_scale)
    return DLMCQ_EINVAL;
  if (!aligned16(x) || !aligned16(w) || !aligned16(x2) || !aligned16(w2) || !aligned16(w3) || (out && !aligned16(out)) ||
      (codes && !aligned16(codes)) || !aligned16(codes3))
    return DLMCQ_EALIGN;
  if (M >= (1ll << 31) || N * H2 * W2 * C2 >= (1ll << 40)) return DLMCQ_ERANGE;
  ChainArgs a{};
  a.x = static_cast<const int8_t*>(x); a.w1 = w; a.s_w1 = w_scale; a.wsum1 = wsum; a.bias1 = bias;
  a.s_in1 = in_scale; a.zp_in1 = in_zero_point; a.shift1 = x_is_unsigned ? 128 : 0;
  a.x2 = static_cast<const int8_t*>(x2); a.w2 = w2; a.s_w2 = w_scale2; a.wsum2 = wsum2; a.bias2 = bias2;
  a.s_in2 = in_scale2; a.zp_in2 = in_zero_point2; a.shift2 = x2_is_unsigned ? 128 : 0;
  a.P = (int)H; a.Q = (int)W; a.H2 = (int)H2; a.W2 = (int)W2; a.stride2 = stride2;
  a.qdiv = make_fastdiv((uint32_t)W);
  a.pdiv = make_fastdiv((uint32_t)H);
  a.residual = nullptr; a.out = out; a.codes = static_cast<uint8_t*>(codes);
  a.w3 = w3; a.s_w3 = w_scale3; a.wsum3 = wsum3; a.bias3 = bias3; a.codes2 = static_cast<uint8_t*>(codes3);
  return chain_launch(a, M, C, K, C2, K3, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g, relu3, codes3, q3_scale,
                      q3_zero_point, q3_lo, q3_hi, q3_form, q3_ste_g, rows_per_tile, stream);
}
