// A MobileOne / MobileNet unit - depthwise 3x3 (stride 1, pad 1) + ReLU + quantiser, then pointwise 1x1 + ReLU + the consumer's
// quantiser - as ONE kernel (modules/conv.py:13-19 twice, the first with `groups`; QBase forward modules/base.py:96-102 between
// them; asymmetric per-channel weights ops.py:129-136 on both layers: BASELINE configs[4]).
//
// Why (round 3's plan profile of MobileOne-S1 W4A8, batch 1024): the depthwise kernel runs at 0.24-0.31 of the HBM roofline -
// it is bound by its own ~24 vector instructions per element -, the pointwise layer behind it at 0.5-0.9 POP/s - its K loop has
// 3-8 steps, a tile's life is latency -, and between them the wide code tensor is written and read back (2 of the unit's ~4
// bytes per element).  Here the depthwise output never leaves the CU: a workgroup owns 64 consecutive positions of the LINEAR
// FRAME (csrc/conv3x3_i8.hip: image n as (H + 1) x (W + 1) positions with shared zero-point borders, so the nine taps are nine
// fixed shifts of one sequence), walks the C channels in chunks of 64, and per chunk
//   * takes the chunk's halo tile (64 + 2 Wp + 2 positions x 64 B, LDS-DMA, requested one chunk ahead) and the depthwise
//     constants of its 64 channels (a 2 KB table, dlmcq_dwpw_pack_table) from LDS,
//   * evaluates the depthwise layer exactly as conv_dw3_i8_kernel does (byte transposition, v_dot4_i32_i8, the fp32 chain, the
//     quantiser with the ReLU folded in): thread (position, 16 channels) -> 16 finished code bytes into the LDS code tile,
//   * and multiplies the 64 x 64 code tile with the pointwise layer's weight chunk (K x 64 B, LDS-DMA during the depthwise
//     arithmetic) on v_mfma_i32_32x32x32_i8 into accumulators that stay in registers for the whole tile (weights as A, pixels
//     as B: a lane owns 16 consecutive output channels of one pixel - conv_i8.hip's swapped epilogue).
// The matrix work of one workgroup runs in the shadow of the vector work of the others on the CU.  Same integers, same fp32
// chains, same quantisers as the two launches: bit-identical (tests/test_gpu_mobileone.py).
#include "conv_i8_common.h"

namespace dlmcq {

struct DwPwArgs {
  const int8_t* x;           // depthwise input codes [N][H][W][C]
  const uint32_t* table;     // dlmcq_dwpw_pack_table: [C / 64][64 records][8 dwords]
  int dw_asym, dw_bias, dw_relu, x_signed;
  // pointwise layer: w [K][C] int8, per-channel scale / code sum / bias / offset [K]
  const int8_t* w;
  const float* s_w;
  const int32_t* wsum;
  const float* bias;
  const float* w_off;        // null: symmetric weights
  const float* s_in;         // the pointwise layer's input scale (the depthwise output quantiser's dequantising scale)
  const float* zp_in;
  int N, H, W, C, K;
  int Wp, FS, hp;            // W + 1, (H + 1) (W + 1), halo pieces per chunk
  uint32_t MQ;               // N FS frame positions
  const float* zp_x;         // the depthwise input's zero point (null: 0): the border code
  FastDiv fsdiv, wpdiv;
  uint8_t* codes;            // [N][H][W][K]
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int DWPW_TP = 64;      // frame positions per workgroup
constexpr int DWPW_WGS(int kb) { return kb <= 192 ? 3 : 2; }

template <int KB, int HPW>
__global__ __launch_bounds__(256, DWPW_WGS(KB)) void conv_dwpw_i8_kernel(DwPwArgs a, ConvEpi ep1, ConvEpi ep2) {
  constexpr int U3 = KB / 64;            // 64-row units of the pointwise weight chunk = DMA pieces per wave per chunk
  constexpr int NB = KB / 64;            // 32-channel accumulator blocks per wave (wave (wr, wc): pixels wr*32.., channels wc*KB/2..)
  constexpr int WB = KB * 64;            // bytes of one weight chunk
  constexpr int HALO = HPW * 4 * 1024;   // bytes of one halo buffer
  constexpr int SROW = KB + 16;          // staged output row (conflict-free 16-byte accesses)
  constexpr int OPER = WB + 2 * HALO + 2 * 2048 + 4096;
  constexpr int STAGE = DWPW_TP * SROW;
  constexpr int LDS_BYTES = OPER < STAGE ? STAGE : OPER;
  constexpr int PAR_BYTES = 4 * KB * 4;
  __shared__ __attribute__((aligned(1024))) int8_t lds[LDS_BYTES + PAR_BYTES + 3 * DWPW_TP * 4];
  int8_t* const wbuf = lds;
  int8_t* const halo = lds + WB;
  int8_t* const tab = halo + 2 * HALO;
  int8_t* const ctile = tab + 2 * 2048;
  int8_t* const par = lds + LDS_BYTES;     // s_in s_w | (128 - zp) SUM qw | bias | s_in o_w   (per output channel)
  int* const psum = reinterpret_cast<int*>(par + PAR_BYTES);      // per position: SUM of its depthwise output codes
  int* const prow = psum + DWPW_TP;                                // per position: output pixel index, -1 for a junk position

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const uint32_t q0 = (uint32_t)blockIdx.x * DWPW_TP;
  const int nchunks = a.C >> 6;

  // ---- the pointwise layer's per-channel constants: LDS-DMA, requested first ----
  {
    const void* arrs[4] = {a.s_w, a.wsum, a.bias, a.w_off};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!arrs[r]) continue;
#pragma unroll
      for (int c = 0; c < KB / 64; ++c) {
        if (((r * (KB / 64) + c) & 3) != wave) continue;
        __builtin_amdgcn_global_load_lds((gptr_t)(static_cast<const int32_t*>(arrs[r]) + c * 64 + lane), (lptr_t)(par + (r * KB + c * 64) * 4), 4, 0, 0);
      }
    }
  }
  // ---- halo DMA: piece i of this wave covers halo positions (i * 4 + wave) * 16 .. + 15; LDS slot s of position p holds the logical
  // 16-byte segment s ^ ((p >> 2) & 3) (swizzle on the source side, undone by the readers) ----
  const int lrow = lane >> 2, pslot = lane & 3;
  const int zpi_x = (int)(a.zp_x ? a.zp_x[0] : 0.0f);                     // (integral: every int8 layer checks it once after calibration)
  const int8_t* const padline = g_pad_table.b + ((zpi_x & 0xff) << 6);
  const int8_t* hsrc[HPW];
  int hinc[HPW], hpc[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    int pc = i * 4 + wave;
    pc = pc < a.hp ? pc : a.hp - 1;                                        // (surplus pieces re-load the last one: same bytes, same place)
    hpc[i] = pc;
    const int p = pc * 16 + lrow;
    const uint32_t f = q0 + (uint32_t)p;
    const uint32_t n = fdiv(f, a.fsdiv);
    const uint32_t rem = f - n * (uint32_t)a.FS;
    const uint32_t fy = fdiv(rem, a.wpdiv);
    const uint32_t fx = rem - fy * (uint32_t)a.Wp;
    const bool in = n < (uint32_t)a.N && fy >= 1u && fx >= 1u;
    const int seg = pslot ^ ((p >> 2) & 3);
    hsrc[i] = in ? a.x + ((int64_t)((n * (uint32_t)a.H + fy - 1u) * (uint32_t)a.W + fx - 1u)) * a.C + seg * 16 : padline;
    hinc[i] = in ? 64 : 0;
  }
  auto issue_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)hsrc[i], (lptr_t)(halo + buf * HALO + hpc[i] * 1024), 16, 0, 0);
      hsrc[i] += hinc[i];
    }
  };
  // the depthwise table of a chunk: 2 KB = two pieces; waves 2 and 3 re-load what waves 0 and 1 load (every wave issues the same
  // number of vector-memory instructions: the counted wait in front of the matrix step relies on it)
  const uint32_t* tsrc = a.table + (wave & 1) * 256 + lane * 4;
  auto issue_tab = [&](int buf) {
    __builtin_amdgcn_global_load_lds((gptr_t)tsrc, (lptr_t)(tab + buf * 2048 + (wave & 1) * 1024), 16, 0, 0);
    tsrc += 512;
  };
  // ---- weight DMA: a wave-instruction lands 16 rows x 64 B of a 64-row unit; LDS row d of a 32-row block holds channel
  // 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3), which makes a lane's 16 accumulator registers 16 consecutive channels ----
  const int drow = wave * 16 + lrow;
  const int dseg = (pslot ^ ((drow >> 2) & 3)) * 16;
  const int d3 = drow & 31;
  const int prow3 = (drow & 32) + 16 * ((d3 >> 2) & 1) + 4 * (d3 >> 3) + (d3 & 3);
  const int8_t* wsrc = a.w + (int64_t)prow3 * a.C + dseg;
  auto issue_w = [&]() {
#pragma unroll
    for (int u = 0; u < U3; ++u)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (int64_t)u * 64 * a.C), (lptr_t)(wbuf + u * 4096 + wave * 1024), 16, 0, 0);
    wsrc += 64;
  };

  issue_halo(0);
  issue_tab(0);
  const float sin2 = a.s_in[0];
  const float zpf2 = a.zp_in ? a.zp_in[0] : 0.0f;
  const int zpi2 = (int)__builtin_rintf(zpf2);
  // ---- per position: where its output pixel lives (frame positions with x = W or y = H, or beyond the batch, are junk) ----
  if (tid < DWPW_TP) {
    const uint32_t f = q0 + (uint32_t)tid;
    const uint32_t n = fdiv(f, a.fsdiv);
    const uint32_t rem = f - n * (uint32_t)a.FS;
    const uint32_t fy = fdiv(rem, a.wpdiv);
    const uint32_t fx = rem - fy * (uint32_t)a.Wp;
    prow[tid] = (n < (uint32_t)a.N && fy < (uint32_t)a.H && fx < (uint32_t)a.W) ? (int)((n * (uint32_t)a.H + fy) * (uint32_t)a.W + fx) : -1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // the pointwise constants have landed: (s_w, SUM qw, o_w) -> (s_in s_w, (128 - zp) SUM qw, s_in o_w) in place, once per channel
  for (int k = tid; k < KB; k += 256) {
    float* pf = reinterpret_cast<float*>(par) + k;
    int* pi = reinterpret_cast<int*>(par) + KB + k;
    *pf = sin2 * *pf;
    *pi = (128 - zpi2) * *pi;
    if (a.w_off) pf[3 * KB] = sin2 * pf[3 * KB];
  }

  // ---- the depthwise half: thread (position dp, 16 channels c16 of the chunk) ----
  const int dp = tid >> 2, c16 = tid & 3;
  const uint32_t xw = a.x_signed ? 0u : 0x80808080u;
  const int dz = (a.x_signed ? 0 : 128) - zpi_x;
  const EpiQuant eq1(ep1, ep1.relu != 0);      // codes only: the ReLU is folded into the quantiser's clamp (as conv_dw3_i8_kernel)
  int hoff[9];                                  // byte offsets of the nine taps' 16-byte segments in a halo buffer
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int pp = dp + r * a.Wp + s;
      hoff[r * 3 + s] = pp * 64 + ((c16 ^ ((pp >> 2) & 3)) << 4);
    }
  const int coff = dp * 64 + ((c16 ^ ((dp >> 2) & 3)) << 4);      // this thread's 16 bytes of the code tile
  int csum = 0;                                                    // SUM of this thread's emitted codes over all chunks

  i32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0;

  for (int n = 0; n < nchunks; ++n) {
    const int b = n & 1;
    // chunk n's halo tile and table (requested a chunk ago) have landed; everyone has left the matrix step of chunk n - 1
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue_w();                                   // needed behind the depthwise arithmetic: a single buffer is enough
    if (n + 1 < nchunks) {
      issue_halo(1 - b);
      issue_tab(1 - b);
    }
    const int8_t* const hb = halo + b * HALO;
    const int8_t* const tb = tab + b * 2048;
    u32x4 t9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t9[k] = *reinterpret_cast<const u32x4*>(hb + hoff[k]) ^ xw;
    uint32_t codes[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {                                   // 4 channels at a time (csrc/conv_dw_i8.hip, same arithmetic)
      uint32_t T[4], U[4], V[4];
      {
        const uint32_t l01 = __builtin_amdgcn_perm(t9[1][d], t9[0][d], 0x05010400u), h01 = __builtin_amdgcn_perm(t9[1][d], t9[0][d], 0x07030602u);
        const uint32_t l23 = __builtin_amdgcn_perm(t9[3][d], t9[2][d], 0x05010400u), h23 = __builtin_amdgcn_perm(t9[3][d], t9[2][d], 0x07030602u);
        T[0] = __builtin_amdgcn_perm(l23, l01, 0x05040100u);
        T[1] = __builtin_amdgcn_perm(l23, l01, 0x07060302u);
        T[2] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
        T[3] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);
      }
      {
        const uint32_t l01 = __builtin_amdgcn_perm(t9[5][d], t9[4][d], 0x05010400u), h01 = __builtin_amdgcn_perm(t9[5][d], t9[4][d], 0x07030602u);
        const uint32_t l23 = __builtin_amdgcn_perm(t9[7][d], t9[6][d], 0x05010400u), h23 = __builtin_amdgcn_perm(t9[7][d], t9[6][d], 0x07030602u);
        U[0] = __builtin_amdgcn_perm(l23, l01, 0x05040100u);
        U[1] = __builtin_amdgcn_perm(l23, l01, 0x07060302u);
        U[2] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
        U[3] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) V[j] = (t9[8][d] >> (8 * j)) & 0xffu;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // record of channel c16 * 16 + d * 4 + j of the chunk: slot (position in the thread's 16 channels) * 4 + c16
        const int8_t* rec = tb + ((d * 4 + j) * 4 + c16) * 32;
        const u32x4 tw = *reinterpret_cast<const u32x4*>(rec);
        const f32x4 tp = *reinterpret_cast<const f32x4*>(rec + 16);
        int s1 = __builtin_amdgcn_sdot4((int)T[j], (int)tw.x, (int)tw.w, false);
        s1 = __builtin_amdgcn_sdot4((int)U[j], (int)tw.y, s1, false);
        s1 = __builtin_amdgcn_sdot4((int)V[j], (int)tw.z, s1, false);
        float r = (float)s1 * tp.x;                                  // S1 = SUM (q - zp) * qw, exact
        if (a.dw_asym) {
          int s0 = __builtin_amdgcn_sdot4((int)T[j], 0x01010101, 9 * dz, false);
          s0 = __builtin_amdgcn_sdot4((int)U[j], 0x01010101, s0, false);
          s0 = __builtin_amdgcn_sdot4((int)V[j], 0x01010101, s0, false);
          r = r + (float)s0 * tp.y;                                  // S0 = SUM (q - zp)
        }
        if (a.dw_bias) r = r + tp.z;
        v[j] = r;
      }
      codes[d] = eq1.code4(v);
      csum = (int)__builtin_amdgcn_udot4(codes[d], 0x01010101u, (uint32_t)csum, false);
    }
    *reinterpret_cast<u32x4*>(ctile + coff) = u32x4{codes[0], codes[1], codes[2], codes[3]};
    // the code tile is complete and this chunk's weights (older than the halo / table pieces requested behind them) have landed
    if (n + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(HPW + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- the matrix step: this chunk's 64 channels into the resident accumulators ----
    {
      const int R = wr * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const i32x4 t = *reinterpret_cast<const i32x4*>(ctile + R * 64 + (((ks * 2 + hsel) ^ ((R >> 2) & 3)) << 4));
        const i32x4 a2 = i32x4{(int)(t.x ^ 0x80808080u), (int)(t.y ^ 0x80808080u), (int)(t.z ^ 0x80808080u), (int)(t.w ^ 0x80808080u)};
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int kb = wc * (KB / 2) + j * 32 + l31;
          const i32x4 bf = *reinterpret_cast<const i32x4*>(wbuf + kb * 64 + (((ks * 2 + hsel) ^ ((kb >> 2) & 3)) << 4));
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, a2, acc[j], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: the pointwise layer's dequantise (+ the asymmetric term), ReLU, its consumer's quantiser; codes only.  Lane
  // (p = l31, hsel) of wave (wr, wc) holds channels wc * KB/2 + 32 j + 16 hsel + 0 .. 15 of position wr * 32 + p ----
  csum += __shfl_xor(csum, 1, 64);
  csum += __shfl_xor(csum, 2, 64);
  if (c16 == 0) psum[dp] = csum;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the sums are there; everyone is done with the operand buffers (the stage re-uses them)
  const EpiQuant eq2(ep2, ep2.relu != 0);
  const int lr = wr * 32 + l31;
  // SUM x' / s_in of this lane's pixel = SUM (q - zp) over ALL C channels (padded channels hold the code of 0 like any other)
  const float s0f = (float)(psum[lr] - zpi2 * a.C);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int cb = wc * (KB / 2) + j * 32 + hsel * 16;
    f32x4 y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (cb + 4 * q) * 4);
      const i32x4 co = *reinterpret_cast<const i32x4*>(par + (KB + cb + 4 * q) * 4);
      const f32x4 bs = a.bias ? *reinterpret_cast<const f32x4*>(par + (2 * KB + cb + 4 * q) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      y[q] = f32x4{dequant1(acc[j][4 * q] + co.x, mu.x, bs.x), dequant1(acc[j][4 * q + 1] + co.y, mu.y, bs.y),
                   dequant1(acc[j][4 * q + 2] + co.z, mu.z, bs.z), dequant1(acc[j][4 * q + 3] + co.w, mu.w, bs.w)};
      if (a.w_off) {
        const f32x4 wo = *reinterpret_cast<const f32x4*>(par + (3 * KB + cb + 4 * q) * 4);
        y[q] = f32x4{y[q].x + s0f * wo.x, y[q].y + s0f * wo.y, y[q].z + s0f * wo.z, y[q].w + s0f * wo.w};
      }
    }
    uint32_t wq[4];
    bool uq[4];
    eq2.code4n(y, wq, uq);
    *reinterpret_cast<i32x4*>(lds + lr * SROW + cb) = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (a row's two column halves come from two waves)
  constexpr int LPR = KB / 16;                    // lanes per staged row
  constexpr int RPP = 256 / LPR;                  // whole rows per pass of the workgroup (192-wide: 21, four threads idle)
  const int srow = tid / LPR, sseg = tid % LPR;
#pragma unroll
  for (int it = 0; it < (DWPW_TP + RPP - 1) / RPP; ++it) {
    const int r = it * RPP + srow;
    if (srow < RPP && r < DWPW_TP) {
      const int pix = prow[r];
      if (pix >= 0) {
        const i32x4 c16v = *reinterpret_cast<const i32x4*>(lds + r * SROW + sseg * 16);
        __builtin_nontemporal_store(c16v, reinterpret_cast<i32x4*>(a.codes + (int64_t)pix * a.K + sseg * 16));
      }
    }
  }
}

// The depthwise layer's per-channel constants in the layout the fused kernel's chunks take by LDS-DMA: per 64-channel chunk 64
// records of 8 dwords {taps 0-3, taps 4-7, tap 8 (weights as signed bytes), (shift - zp) SUM w | s_in s_w, s_in o_w, bias, 0}, the
// record of channel c at slot (c & 15) * 4 + ((c >> 4) & 3) (neighbouring lanes - neighbouring channel groups - read neighbouring
// records).  The same values conv_dw3_i8_kernel puts into its LDS table.
__global__ __launch_bounds__(DLMCQ_BLOCK) void dwpw_pack_table_kernel(const int8_t* __restrict__ w, const float* __restrict__ bias,
                                                                     const float* __restrict__ s_in, const float* __restrict__ zp_in,
                                                                     const float* __restrict__ s_w, const float* __restrict__ o_w, int C,
                                                                     int x_signed, uint32_t* __restrict__ table) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sin = s_in[0], zp = zp_in ? zp_in[0] : 0.0f;
  const int dz = (x_signed ? 0 : 128) - (int)zp;
  uint32_t pk[3] = {0u, 0u, 0u};
  int sum = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int wv = w[k * C + c];
    sum += wv;
    pk[k >> 2] |= (uint32_t)(wv & 0xff) << (8 * (k & 3));
  }
  uint32_t* rec = table + (size_t)(c >> 6) * 512 + (((c & 15) * 4 + ((c >> 4) & 3)) * 8);
  rec[0] = pk[0];
  rec[1] = pk[1];
  rec[2] = pk[2];
  rec[3] = (uint32_t)(dz * sum);
  rec[4] = __float_as_uint(sin * s_w[c]);
  rec[5] = __float_as_uint(o_w ? sin * o_w[c] : 0.0f);
  rec[6] = __float_as_uint(bias ? bias[c] : 0.0f);
  rec[7] = 0u;
}

}  // namespace dlmcq

using namespace dlmcq;

extern "C" int dlmcq_dwpw_pack_table(const int8_t* w, const float* bias, const float* in_scale, const float* in_zero_point,
                                     const float* w_scale, const float* w_offset, int64_t C, int32_t x_is_unsigned, void* table,
                                     dlmcq_stream_t stream) {
  if (C < 64 || (C & 63)) return DLMCQ_EINVAL;
  if (!w || !in_scale || !w_scale || !table) return DLMCQ_EINVAL;
  if (!aligned16(table)) return DLMCQ_EALIGN;
  hipLaunchKernelGGL(dwpw_pack_table_kernel, dim3((uint32_t)((C + DLMCQ_BLOCK - 1) / DLMCQ_BLOCK)), dim3(DLMCQ_BLOCK), 0,
                     reinterpret_cast<hipStream_t>(stream), w, bias, in_scale, in_zero_point, w_scale, w_offset, (int)C, x_is_unsigned ? 0 : 1,
                     static_cast<uint32_t*>(table));
  return launch_status();
}

extern "C" int dlmcq_conv2d_dwpw_i8_nhwc(const void* x, const void* dw_table, int32_t dw_asym, int32_t dw_bias, int32_t dw_relu,
                                         const float* in_zero_point, int64_t N, int64_t H, int64_t W, int64_t C, int32_t x_is_unsigned,
                                         const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form,
                                         float q_ste_g, const int8_t* w, const float* bias, const int32_t* wsum,
                                         const float* pw_in_scale, const float* w_scale, const float* w_offset, int64_t K,
                                         int32_t relu, void* codes, const float* q2_scale, const float* q2_zero_point, int32_t q2_lo,
                                         int32_t q2_hi, int32_t q2_form, float q2_ste_g, dlmcq_stream_t stream) {
  if (N < 0 || H < 1 || W < 1 || C < 64 || (C & 63) || (K != 128 && K != 192 && K != 512)) return DLMCQ_EINVAL;
  if (N == 0) return DLMCQ_OK;
  if (!x || !dw_table || !q_scale || !w || !wsum || !pw_in_scale || !w_scale || !codes || !q2_scale) return DLMCQ_EINVAL;
  if (q_lo != 0 || q_hi != 255) return DLMCQ_EINVAL;      // the matrix step reads the depthwise codes as unsigned bytes (shift 128)
  if (!aligned16(x) || !aligned16(dw_table) || !aligned16(w) || !aligned16(codes)) return DLMCQ_EALIGN;
  ConvEpi ep1{}, ep2{};
  if (q2_lo > q2_hi || q2_lo < -128 || q2_hi > 255 || q2_hi - q2_lo > 255 || q_form < DLMCQ_FORM_EMULATE || q_form > DLMCQ_FORM_SYMMETRIC ||
      !epi_set_form(ep2, q2_form, q2_lo, q2_hi))
    return DLMCQ_EINVAL;
  DwPwArgs a{};
  a.Wp = (int)W + 1;
  a.FS = (int)((H + 1) * (W + 1));
  const int64_t mq = N * a.FS;
  if (mq + 4096 >= (1ll << 31) || N * H * W * (C > K ? C : K) >= (1ll << 40)) return DLMCQ_ERANGE;
  a.hp = (DWPW_TP + 2 * a.Wp + 2 + 15) / 16;
  if (a.hp > 12) return DLMCQ_EINVAL;                      // images wider than 61 pixels: two launches (the plan does that)
  a.x = static_cast<const int8_t*>(x);
  a.table = static_cast<const uint32_t*>(dw_table);
  a.dw_asym = dw_asym != 0; a.dw_bias = dw_bias != 0; a.dw_relu = dw_relu != 0; a.x_signed = x_is_unsigned ? 0 : 1;
  a.w = w; a.s_w = w_scale; a.wsum = wsum; a.bias = bias; a.w_off = w_offset; a.s_in = pw_in_scale; a.zp_in = q_zero_point;
  a.N = (int)N; a.H = (int)H; a.W = (int)W; a.C = (int)C; a.K = (int)K;
  a.MQ = (uint32_t)mq;
  a.fsdiv = make_fastdiv((uint32_t)a.FS);
  a.wpdiv = make_fastdiv((uint32_t)a.Wp);
  a.codes = static_cast<uint8_t*>(codes);
  a.zp_x = in_zero_point;
  ep1.relu = dw_relu != 0; ep1.q_scale = q_scale; ep1.q_zp = q_zero_point; ep1.q_lo = (float)q_lo; ep1.q_hi = (float)q_hi; ep1.q_g = q_ste_g;
  ep1.q_form = q_form; ep1.codes = reinterpret_cast<uint8_t*>(uintptr_t(1));     // (the quantiser is always needed: the matrix step reads its codes)
  ep2.relu = relu != 0; ep2.q_scale = q2_scale; ep2.q_zp = q2_zero_point; ep2.q_lo = (float)q2_lo; ep2.q_hi = (float)q2_hi; ep2.q_g = q2_ste_g;
  ep2.codes = a.codes;
  const int64_t tiles = (mq + DWPW_TP - 1) / DWPW_TP;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((uint32_t)tiles), block(256);
  const int hpw = (a.hp + 3) / 4;     // halo pieces per wave
#define DLMCQ_DWPW_GO(KB_)                                                                                  \
  do {                                                                                                      \
    if (hpw <= 2) hipLaunchKernelGGL((conv_dwpw_i8_kernel<KB_, 2>), grid, block, 0, st, a, ep1, ep2);        \
    else hipLaunchKernelGGL((conv_dwpw_i8_kernel<KB_, 3>), grid, block, 0, st, a, ep1, ep2);                 \
  } while (0)
  if (K == 128) DLMCQ_DWPW_GO(128);
  else if (K == 192) DLMCQ_DWPW_GO(192);
  else DLMCQ_DWPW_GO(512);
#undef DLMCQ_DWPW_GO
  return launch_status();
}
