// Depthwise 3 x 3 (stride 1, padding 1) on activation codes with the multiply-adds on the MATRIX cores (modules/conv.py:13-19 with
// `groups`, QBase forward modules/base.py:96-102 behind it, asymmetric per-channel weights ops.py:129-136: MobileOne-S1's 17
// stride-1 depthwise layers, BASELINE configs[4]).
//
// Why (round 4 counters, tools/pmc_pw.sh and the plan-wide pass in LABNOTES 11): conv_dw3p2_i8_kernel is bound by its own vector
// instructions - 21-27 per output element at one per 4 clocks and SIMD: byte transposes, six v_dot4 per element (three for SUM
// q w, three for SUM q of the asymmetric term), conversions, the fp32 chain, the quantiser - while the matrix pipes idle.  A
// depthwise layer is a GEMM with a DIAGONAL weight matrix: wasteful in multiply-adds (31 of 32 are zeros), but an int8 MFMA does
// 32 768 of them in 32 clocks, so even at 1/32 efficiency it beats six v_dot4 per element by 2.4 x, and it needs NO byte
// transposition: the operand fragment is 16 consecutive channels of one pixel, as they lie in memory.  One
// v_mfma_i32_32x32x32_i8 takes, for 32 positions x 16 channels,
//     B (32 k x 32 positions)   k = 16 h + c:  channel c of tap t_h at the position   (two taps per instruction)
//     A (32 rows x 32 k)        row m < 16:  w[t_h][channel(m)] at k = 16 h + channel(m)      -> S1 = SUM (q - zp) qw
//                               row 16 + m:  1 at the same place                               -> S0 = SUM (q - zp)
// so five instructions (nine taps in pairs) give a lane BOTH sums of 8 channels of one position, already in the accumulator
// registers; the zero-point terms ((shift - zp) SUM w, 9 (shift - zp)) are the accumulators' start values (the C operand of the
// first instruction: no vector instruction).  What is left for the vector unit: 2 conversions, the fp32 chain on pairs and the
// plain quantiser (EpiQuant::code4n_plain) - 8 instructions per element in the blocks (10.75 with unsigned input codes, whose
// fragments are re-centred), 16 measured over the whole kernel with the frame arithmetic of requests and stores.
//
// Structure: the linear frame of conv3x3_i8.hip / conv_dwpw_i8.hip (image n as (H + 1) x (W + 1) positions with shared
// zero-point borders: the nine taps are nine fixed shifts of one sequence).  A workgroup owns ONE 64-channel chunk - its four
// waves one 16-channel segment each, weights and constants in registers for the whole launch - and walks tiles of 192 positions:
// halo tiles by LDS-DMA (two buffers: the next tile is requested before this tile's arithmetic; counted waits), two blocks'
// MFMA chains interleaved (five dependent instructions per block otherwise), fragments by ds_read_b128 at constant offsets,
// codes through an LDS stage so that a position's 64 bytes leave together.  Same integers, same fp32 chain, same quantiser as
// conv_dw3p2_i8_kernel: bit-identical (tests/test_gpu_mobileone.py).
#include "conv_i8_common.h"

namespace dlmcq {

struct DwmArgs {
  const int8_t* x;           // input codes [N][H][W][C]
  const int8_t* w;           // [9][C] int8 (tap-major, as the other depthwise kernels take them)
  const float* s_w;          // [C]
  const float* o_w;          // [C] or null (symmetric weights)
  const float* bias;         // [C] or null
  const float* s_in;
  const float* zp_in;        // null: 0
  int N, H, W, C, x_signed;
  int Wp, FS, hp;            // W + 1, (H + 1) (W + 1), halo pieces of 16 positions per tile
  uint32_t MQ;               // N FS frame positions
  int ntiles, nchunks;
  FastDiv fsdiv, wpdiv;
  uint8_t* codes;            // [N][H][W][C]
};

constexpr int DWM_TP = 192;      // output positions per tile (128: +2.6 % time - the requests' and stores' fixed costs; 256 at two workgroups per CU: +6 %)

// XS: the input codes are signed bytes (int8 codes, or an unsigned quantiser's codes handed over as `code - 128`): no re-centring
// of the fragments (4 vector instructions per MFMA otherwise)
template <int HPW, bool XS>
__global__ __launch_bounds__(256, 3) void conv_dwm_i8_kernel(DwmArgs a, ConvEpi ep) {
  constexpr int HALO = HPW * 4 * 1024;                 // bytes of one halo buffer (16 positions x 64 B per piece)
  constexpr int NHB = 2;                               // halo buffers: the halo of tile t + NHB - 1 is requested while tile t is worked on (three: no faster)
  __shared__ __attribute__((aligned(1024))) int8_t lds[NHB * HALO + DWM_TP * 64];
  int8_t* const stage = lds + NHB * HALO;              // [DWM_TP positions][64 B]: this chunk's codes

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const uint32_t b = blockIdx.x;
  const int chunk = (int)((b >> 3) % (uint32_t)a.nchunks);
  const int group = (int)((b >> 3) / (uint32_t)a.nchunks) * 8 + (int)(b & 7u);
  const int ngroups = (int)(gridDim.x / (uint32_t)a.nchunks);
  const int cbase = chunk * 64 + wave * 16;             // this wave's 16 channels

  const int zpi = (int)(a.zp_in ? a.zp_in[0] : 0.0f);   // (integral: every int8 layer checks it once after calibration)
  const int dz = (a.x_signed ? 0 : 128) - zpi;
  const float sin = a.s_in[0];
  const int8_t* const padline = g_pad_table.b + ((zpi & 0xff) << 6);

  // ---- once per workgroup: this lane's weights as A fragments, its accumulator start values, its fp32 constants ----
  // A fragment of tap pair k: lane (m = l31, hsel) holds A[m][16 hsel .. + 15], one non-zero byte at its row's channel
  const int mrow = l31 & 15;                                          // row within its kind (S1 rows 0-15, S0 rows 16-31)
  const int mch = 8 * ((mrow >> 2) & 1) + (mrow & 3) + 4 * (mrow >> 3);   // the channel (of the wave's 16) that row stands for
  i32x4 wfrag[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int t = 2 * k + hsel;
    int v = 0;
    if (t < 9) v = l31 < 16 ? (int)a.w[t * a.C + cbase + mch] : 1;
    const uint32_t word = (uint32_t)(v & 0xff) << (8 * (mch & 3));
    wfrag[k] = i32x4{(mch >> 2) == 0 ? (int)word : 0, (mch >> 2) == 1 ? (int)word : 0, (mch >> 2) == 2 ? (int)word : 0,
                     (mch >> 2) == 3 ? (int)word : 0};
  }
  // output layout: lane (position n = l31, hsel), register i < 8: S1 of channel 8 hsel + i, register 8 + i: S0 of the same channel
  i32x16 init;
  f32x2 m2[4], mo2[4], b2[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cbase + 8 * hsel + i;
    int sum = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) sum += (int)a.w[t * a.C + c];
    init[i] = dz * sum;
    init[8 + i] = 9 * dz;
    m2[i >> 1][i & 1] = sin * a.s_w[c];
    mo2[i >> 1][i & 1] = a.o_w ? sin * a.o_w[c] : 0.0f;
    b2[i >> 1][i & 1] = a.bias ? a.bias[c] : 0.0f;
  }
  const EpiQuant eq(ep, true);                 // codes only: the ReLU is folded into the quantiser's clamp (as conv_dw3p2_i8_kernel)
  const uint32_t xw = XS ? 0u : 0x80808080u;

  // ---- halo DMA: piece i of this wave covers halo positions (i * 4 + wave) * 16 .. + 15 of the tile; LDS slot s of position p holds
  // the logical 16-byte segment s ^ ((p >> 2) & 3) (swizzle on the source side, undone by the readers) ----
  const int lrow = lane >> 2, pslot = lane & 3;
  // A piece = 16 consecutive frame positions starting at a multiple of 16: its first position is decomposed on the SCALAR unit
  // (wave-uniform), a lane's own position is that plus `lrow` with at most one row wrap (W + 1 >= 15): ~14 vector instructions per
  // piece instead of two divisions' ~25 (they were a seventh of the kernel's instructions)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_halo = [&](int tile, int buf) {
    const uint32_t q0 = (uint32_t)tile * DWM_TP;
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      int pc = i * 4 + wave_u;
      pc = pc < a.hp ? pc : a.hp - 1;                                        // (surplus pieces re-load the last one: same bytes, same place)
      const uint32_t f0 = q0 + (uint32_t)pc * 16u;                           // uniform
      const uint32_t n0 = fdiv(f0, a.fsdiv);
      const uint32_t rem0 = f0 - n0 * (uint32_t)a.FS;
      const uint32_t fy0 = fdiv(rem0, a.wpdiv);
      const uint32_t fx0 = rem0 - fy0 * (uint32_t)a.Wp;
      const int pix0 = (int)((n0 * (uint32_t)a.H + fy0 - 1u) * (uint32_t)a.W + fx0 - 1u);      // (meaningless for a border position: not used there)
      // this lane: position f0 + lrow
      uint32_t fx = fx0 + (uint32_t)lrow;
      const bool wrap = fx >= (uint32_t)a.Wp;
      fx = wrap ? fx - (uint32_t)a.Wp : fx;
      const uint32_t fy = fy0 + (wrap ? 1u : 0u);                            // fy == H + 1: the next image's border row
      const bool in = n0 < (uint32_t)a.N && fy >= 1u && fy <= (uint32_t)a.H && fx >= 1u;
      const int pix = pix0 + lrow - (wrap ? 1 : 0);                          // (a row further, W + 1 positions on: one pixel less than lrow says)
      const int p = pc * 16 + lrow;
      const int seg = pslot ^ ((p >> 2) & 3);
      const int8_t* src = in ? a.x + (int64_t)pix * a.C + chunk * 64 + seg * 16 : padline;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + buf * HALO + pc * 1024), 16, 0, 0);
    }
  };
  // byte offset of this lane's fragment for tap pair k in block 0 of a halo buffer (lane (n, hsel) reads tap 2 k + hsel of position n;
  // the fifth pair's second half multiplies zeros: it re-reads tap 8).  A block further is 32 positions = 2 KB further and the
  // swizzle term ((position >> 2) & 3) does not change with it: blocks and buffers are constant offsets of ONE address per pair.
  int foff[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int t = (2 * k + hsel) < 9 ? 2 * k + hsel : 8;
    const int pp = (t / 3) * a.Wp + (t % 3) + l31;
    foff[k] = pp * 64 + ((wave ^ ((pp >> 2) & 3)) << 4);
  }
  const v4i r_c = make_rsrc(a.codes, (uint32_t)((int64_t)a.N * a.H * a.W * a.C));
  int tile = group;
  int buf = 0;
  // prologue: the first NHB - 1 tiles' halos (a wave issues HPW pieces per tile whether the tile exists or not - the last one again
  // - so that the counted wait below always means the same thing)
  auto request = [&](int t, int bf) { issue_halo(t < a.ntiles ? t : a.ntiles - 1, bf); };
#pragma unroll
  for (int d = 0; d < NHB - 1; ++d) request(tile + d * ngroups, d);
  constexpr int NSTO = DWM_TP / 64;     // 16-byte stores a thread issues per tile (round 5: the wait below is derived from it - it used to say "+ 2"
                                        // for the 128-position tiles of an earlier build and was stricter than meant at 192: ADVICE r4)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NHB - 2) * HPW) : "memory");      // the first tile's halo (the loop's wait counts stores that do not exist yet)
  for (; tile < a.ntiles; tile += ngroups, buf = (buf + 1 == NHB ? 0 : buf + 1)) {
    // this tile's halo has landed; the halos of the next NHB - 2 tiles and the previous tile's NSTO stores may still be in flight
    // (behind the prologue there are no stores yet: the wait is then stricter than it has to be)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NHB - 2) * HPW + NSTO) : "memory");
    request(tile + (NHB - 1) * ngroups, buf == 0 ? NHB - 1 : buf - 1);
    const int8_t* const hb = lds + buf * HALO;
    // two blocks of 32 positions at a time: their accumulation chains are independent, so one block's MFMA issues while the
    // other's is in flight (five DEPENDENT instructions per block otherwise: their latency, not their 32 clocks, was the time)
#pragma unroll
    for (int bp = 0; bp < DWM_TP / 64; ++bp) {
      i32x16 acc[2];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const i32x4 t = *reinterpret_cast<const i32x4*>(hb + foff[k] + (2 * bp + e) * 2048);
          const i32x4 bf = XS ? t : i32x4{(int)(t.x ^ xw), (int)(t.y ^ xw), (int)(t.z ^ xw), (int)(t.w ^ xw)};
          acc[e] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wfrag[k], bf, k == 0 ? init : acc[e], 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // the fp32 chain of conv_dw3p2_i8_kernel on pairs: r = S1 * m; r = r + S0 * mo; r = r + b
        f32x4 v[2];
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          f32x2 r = f32x2{(float)acc[e][2 * jp], (float)acc[e][2 * jp + 1]} * m2[jp];
          r = r + f32x2{(float)acc[e][8 + 2 * jp], (float)acc[e][9 + 2 * jp]} * mo2[jp];
          r = r + b2[jp];
          v[jp >> 1][2 * (jp & 1)] = r.x;
          v[jp >> 1][2 * (jp & 1) + 1] = r.y;
        }
        uint32_t wq[2];
        eq.code4n_plain(v, wq);
        // (16-byte segment s of position p sits in slot s ^ ((p >> 2) & 3), like the halo tiles: unswizzled, the 32 positions of a
        //  block write through 8 banks - SQ_LDS_BANK_CONFLICT was half of the kernel's LDS cycles)
        {
          const int ps = (2 * bp + e) * 32 + l31;
          *reinterpret_cast<uint2*>(stage + ps * 64 + ((wave ^ ((ps >> 2) & 3)) << 4) + hsel * 8) = uint2{wq[0], wq[1]};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the tile's codes are staged (a position's 64 bytes come from four waves)
    // row-major out of the stage: 4 threads x 16 bytes per position; frame positions with x = W or y = H, or beyond the batch, are
    // junk: their store goes to an offset beyond the tensor (every thread issues its NSTO stores: the counted wait above)
    const uint32_t q0 = (uint32_t)tile * DWM_TP;
#pragma unroll
    for (int it = 0; it < DWM_TP / 64; ++it) {
      const int sseg = tid & 3;
      const uint32_t f0 = q0 + (uint32_t)(it * 64 + wave_u * 16);            // uniform: this wave's 16 positions of the pass
      const uint32_t n0 = fdiv(f0, a.fsdiv);
      const uint32_t rem0 = f0 - n0 * (uint32_t)a.FS;
      const uint32_t fy0 = fdiv(rem0, a.wpdiv);
      const uint32_t fx0 = rem0 - fy0 * (uint32_t)a.Wp;
      const int pix0 = (int)((n0 * (uint32_t)a.H + fy0) * (uint32_t)a.W + fx0);
      uint32_t fx = fx0 + (uint32_t)lrow;
      const bool wrap = fx >= (uint32_t)a.Wp;
      fx = wrap ? fx - (uint32_t)a.Wp : fx;
      // (a group that starts in an image's last frame row - junk for outputs - wraps into row 0 of the NEXT image, which is not:
      //  the frame has H + 1 rows per image, the tensor H)
      const bool next = wrap && fy0 == (uint32_t)a.H;
      const uint32_t fy = next ? 0u : fy0 + (wrap ? 1u : 0u);
      const uint32_t nn = n0 + (next ? 1u : 0u);
      const bool ok = nn < (uint32_t)a.N && fy < (uint32_t)a.H && fx < (uint32_t)a.W;
      const int pix = pix0 + lrow - (wrap ? 1 : 0) - (next ? a.W : 0);
      const int ps = it * 64 + wave_u * 16 + lrow;
      const i32x4 c16 = *reinterpret_cast<const i32x4*>(stage + ps * 64 + ((sseg ^ ((ps >> 2) & 3)) << 4));
      bstore16i(c16, ok ? pix * a.C + chunk * 64 + sseg * 16 : BUF_BIG, r_c);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// where the kernel applies: 3 x 3 / stride 1 / padding 1, codes only with the plain quantiser, whole 64-channel chunks, a halo tile
// of at most 16 pieces
bool conv_dwm_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t R, int64_t S, int32_t stride, int32_t pad, const ConvEpi& ep,
                      const float* out, const void* x) {
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || out || !epi_plain(ep)) return false;
  if (C < 64 || (C & 63) || !aligned16(x) || !aligned16(ep.codes)) return false;
  if ((DWM_TP + 2 * (W + 1) + 2 + 15) / 16 > 20 || W < 14) return false;        // (a piece of 16 positions wraps at most one frame row)
  if (N * (H + 1) * (W + 1) + 4096 >= (1ll << 31) || N * H * W * C >= (int64_t)BUF_BIG) return false;     // 32-bit positions and buffer offsets
  return N * H * W >= 4096;                                     // (weights and constants are set up once per workgroup)
}

int conv_dwm_launch(const int8_t* x, const int8_t* w, const float* bias, const float* in_scale, const float* in_zero_point,
                    const float* w_scale, const float* w_offset, int64_t N, int64_t H, int64_t W, int64_t C, int x_signed,
                    const ConvEpi& ep, hipStream_t st) {
  DwmArgs a{};
  a.x = x; a.w = w; a.s_w = w_scale; a.o_w = w_offset; a.bias = bias; a.s_in = in_scale; a.zp_in = in_zero_point;
  a.N = (int)N; a.H = (int)H; a.W = (int)W; a.C = (int)C; a.x_signed = x_signed;
  a.Wp = (int)W + 1;
  a.FS = (int)((H + 1) * (W + 1));
  a.MQ = (uint32_t)(N * a.FS);
  a.hp = (DWM_TP + 2 * a.Wp + 2 + 15) / 16;
  a.ntiles = (int)(((int64_t)a.MQ + DWM_TP - 1) / DWM_TP);
  a.nchunks = (int)(C / 64);
  a.fsdiv = make_fastdiv((uint32_t)a.FS);
  a.wpdiv = make_fastdiv((uint32_t)a.Wp);
  a.codes = ep.codes;
  static int cus = 0;      // (one device per process: dlmc/_native.py)
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  const int hpw = (a.hp + 3) / 4;
  // workgroups: three per CU (registers) in groups of `nchunks` that share their positions on one XCD
  int ngroups = ((cus * 3) / a.nchunks) & ~7;
  if (ngroups < 8) ngroups = 8;
  const int maxg = (a.ntiles + 7) & ~7;
  if (ngroups > maxg) ngroups = maxg;
  const dim3 grid((uint32_t)(ngroups * a.nchunks)), block(256);
#define DLMCQ_DWM_GO(HP_)                                                                         \
  do {                                                                                            \
    if (x_signed) hipLaunchKernelGGL((conv_dwm_i8_kernel<HP_, true>), grid, block, 0, st, a, ep);  \
    else hipLaunchKernelGGL((conv_dwm_i8_kernel<HP_, false>), grid, block, 0, st, a, ep);          \
  } while (0)
  if (hpw <= 4) DLMCQ_DWM_GO(4);
  else DLMCQ_DWM_GO(5);
#undef DLMCQ_DWM_GO
  return launch_status();
}

}  // namespace dlmcq
