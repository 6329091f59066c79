// 3x3 / stride 1 / pad 1 fused int8-dequant convolution for layers that emit only their consumer's codes (every 3x3 of a
// ResNet bottleneck, every stride-1 layer of RepVGG): the kernel conv_i8.hip's generic implicit GEMM spends its time
// pushing operand bytes through the CU's vector-memory path (16 KB per 128 x 128 x 64 step = the whole 64 B/clock of the path
// for the step's 256 clocks of matrix work), nine times over for the activations - every tap re-fetches the tile.
// Here (modules/conv.py:13-19 is the reference's F.conv2d call):
//
//   * THE FRAME.  Image n is laid out, conceptually, as (H + 1) rows of (W + 1) positions: one border row above and one
//     border column left of the pixels; the border below image n is the border above image n + 1, the border right of a row
//     is the border left of the next row (both hold "x' = 0", i.e. the zero point's code, so they can be shared).  Frame
//     position f = n FS + fy Wp + fx (Wp = W + 1, FS = (H + 1) Wp) holds input pixel (n, fy - 1, fx - 1) or the border.
//     Output pixel (n, y, x) is "row" q = n FS + y Wp + x of the GEMM, and its tap (r, s) reads frame position
//     q + r Wp + s: the convolution is a 1-D stencil over the linear frame, nine FIXED shifts of one sequence.
//     Rows q with x = W or y = H are junk (3.5 % of the rows at 56^2, 6.8 % at 28^2, 12.9 % at 14^2, 23 % at 7^2): they are
//     multiplied and never stored.
//   * THE HALO TILE.  A workgroup owns TM consecutive rows q (they may span image rows and images) for BN output
//     channels.  Per 64-channel chunk of the reduction it stages frame positions [q0, q0 + TM + 2 Wp + 2) x 64 bytes ONCE
//     in LDS (LDS-DMA, 16 positions per wave-instruction, the source address of a lane decided once per tile; border
//     positions read a constant line) and the nine taps read nine shifted ds_read_b128 fragments of it: activation
//     traffic through the vector-memory path / ~7.  The next chunk's tile lands in a second buffer meanwhile.
//   * WEIGHTS ride a 3-slot ring, one (tap, chunk) slab of BN x 64 bytes per K step, two steps in flight, ONE raw barrier
//     and one counted s_waitcnt vmcnt per step (as conv_i8.hip; LDS-DMA with the XOR swizzle on the source side).
//     A 256-row tile halves the slab bytes per MAC of the 128-row kernel.
//   * 8 waves: 2 x 2 MFMA blocks each (v_mfma_i32_32x32x32_i8, weights as A, pixels as B: a lane's 16 accumulator
//     registers of a block are 16 consecutive channels of one pixel - conv_i8.hip's swapped epilogue, ~10 vector
//     instructions per output element); two workgroups per CU.
//   Vector-memory bytes per K step: 8 KB of weights + 1/9 of a ~20 KB tile for 512 clocks of matrix work per CU: ~20 B/clock.
#include "conv_i8_common.h"

namespace dlmcq {

struct HaloGeom {
  int N, H, W, C, K;
  int Wp, FS;            // W + 1, (H + 1) (W + 1)
  uint32_t MQ;           // N FS: rows of the linear frame space
  int nblk_n;
  int hp;                // halo pieces (16 frame positions each) a chunk's tile needs: ceil((TM + 2 Wp + 2) / 16)
  FastDiv fsdiv, wpdiv;
};

template <int BN, int TM, int HPW>
__global__ __launch_bounds__(512, (HPW <= 3 && TM == 256 ? 4 : 2)) void conv3x3_halo_i8_kernel(
    const int8_t* __restrict__ x, const int8_t* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ wsum,
    const float* __restrict__ s_in, const float* __restrict__ zp_in, const float* __restrict__ s_w, HaloGeom g, int shift, ConvEpi ep) {
  constexpr int NBUF = 3;
  constexpr int SLAB = BN * 64;             // one (tap, chunk) of the weights
  constexpr int WC = BN / 64;               // waves across the channels (64 each)
  constexpr int WPX = 8 / WC;               // waves across the pixels
  constexpr int PW = TM / WPX;              // pixels per wave
  constexpr int PB = PW / 32;               // 32-pixel MFMA blocks per wave
  constexpr int HP = HPW * 8;               // halo pieces allocated per buffer
  constexpr int HALO = HP * 1024;
  constexpr int BPW = SLAB / 1024;          // weight pieces per step (one per wave for BN = 128; the first BPW waves otherwise)
  constexpr int RING = NBUF * SLAB;
  constexpr int SROW = BN + 16;             // staged code row (conflict-free 16-byte accesses)
  constexpr int STAGE = TM * SROW;
  constexpr int OPER = RING + 2 * HALO;
  constexpr int LDS_BYTES = OPER < STAGE ? STAGE : OPER;
  constexpr int PAR_BYTES = 3 * BN * 4;
  static_assert(PB >= 1 && BPW <= 8 && (BN == 64 || BN == 128), "tile shape");
  __shared__ __attribute__((aligned(1024))) int8_t lds[LDS_BYTES + PAR_BYTES];
  int8_t* const ring = lds;
  int8_t* const halo = lds + RING;
  int8_t* const par = lds + LDS_BYTES;

  // XCD-aware tile order (as conv_i8.hip): consecutive tiles - the column blocks of one row block - on one XCD
  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t qd = nwg >> 3, rm = nwg & 7u;
  const uint32_t tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
  const int bn = tile % g.nblk_n, bm = tile / g.nblk_n;
  const uint32_t q0 = (uint32_t)bm * TM;
  const int n0 = bn * BN;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const int wc = wave % WC, wp = wave / WC;

  // ---- per-channel constants of the epilogue: LDS-DMA into a table behind everything, requested first ----
  {
    const void* arrs[3] = {s_w, wsum, bias};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!arrs[a]) continue;
#pragma unroll
      for (int c = 0; c < BN / 64; ++c) {
        if (((a * (BN / 64) + c) & 7) != wave) continue;
        __builtin_amdgcn_global_load_lds((gptr_t)(static_cast<const int32_t*>(arrs[a]) + n0 + c * 64 + lane),
                                         (lptr_t)(par + (a * BN + c * 64) * 4), 4, 0, 0);
      }
    }
  }
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin_early = s_in[0];
  const uint32_t xorw = shift ? 0x80808080u : 0u;
  const int8_t* const padline = g_pad_table.b + ((zpi & 0xff) << 6);     // stored UNshifted: the xor happens on read

  // ---- halo DMA: piece i of this wave covers halo positions (i * 8 + wave) * 16 .. + 15; LDS slot s of position p holds the
  // logical 16-byte segment s ^ ((p >> 2) & 3) (swizzle on the source side, undone by the fragment reads) ----
  const int lrow = lane >> 2, pslot = lane & 3;
  const int8_t* hsrc[HPW];
  int hinc[HPW];
  int hpc[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    int pc = i * 8 + wave;
    pc = pc < g.hp ? pc : g.hp - 1;                                        // (surplus pieces re-load the last one: same bytes, same place)
    hpc[i] = pc;
    const int p = pc * 16 + lrow;
    const uint32_t f = q0 + (uint32_t)p;
    const uint32_t n = fdiv(f, g.fsdiv);
    const uint32_t rem = f - n * (uint32_t)g.FS;
    const uint32_t fy = fdiv(rem, g.wpdiv);
    const uint32_t fx = rem - fy * (uint32_t)g.Wp;
    const bool in = n < (uint32_t)g.N && fy >= 1u && fx >= 1u;
    const int seg = pslot ^ ((p >> 2) & 3);
    hsrc[i] = in ? x + ((int64_t)((n * (uint32_t)g.H + fy - 1u) * (uint32_t)g.W + fx - 1u)) * g.C + seg * 16 : padline;
    hinc[i] = in ? 64 : 0;
  }
  auto issue_halo = [&](auto i_c, int buf) {
    constexpr int i = decltype(i_c)::value;
    __builtin_amdgcn_global_load_lds((gptr_t)hsrc[i], (lptr_t)(halo + buf * HALO + hpc[i] * 1024), 16, 0, 0);
    hsrc[i] += hinc[i];
  };

  // ---- weight DMA: wave `wave` < BPW moves slab rows wave * 16 .. + 15 of every step; slab row d of a 32-row block holds channel
  // 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3), which makes a lane's 16 accumulator registers 16 consecutive channels ----
  const bool wload = BPW == 8 || wave < BPW;
  const int8_t* wsrc;
  {
    const int drow = (wave % BPW) * 16 + lrow, d = drow & 31;
    const int k = n0 + (drow & ~31) + 16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
    wsrc = w + (int64_t)k * (9 * g.C) + (pslot ^ ((drow >> 2) & 3)) * 16;
  }
  const int nchunks = g.C >> 6;
  const int w_tap = g.C;                       // next tap, same chunk
  const int w_chunk = 64 - 8 * g.C;            // tap 8 of chunk c -> tap 0 of chunk c + 1
  auto issue_w = [&](auto slot_c, int inc) {
    constexpr int SL = decltype(slot_c)::value;
    if (wload) __builtin_amdgcn_global_load_lds((gptr_t)wsrc, (lptr_t)(ring + SL * SLAB + (wave % BPW) * 1024), 16, 0, 0);
    wsrc += inc;
  };

  // ---- fragment addresses ----
  // weights (MFMA A): rows wc * 64 + jc * 32 + l31 of the slab; the two K halves of a step are slots (ks * 2 + hsel) ^ swz
  int woff[2];
#pragma unroll
  for (int jc = 0; jc < 2; ++jc) {
    const int d = wc * 64 + jc * 32 + l31;
    woff[jc] = d * 64 + ((hsel ^ ((d >> 2) & 3)) << 4);
  }
  const int pbase = wp * PW + l31;             // this lane's pixel row of block 0 (halo position of tap (0, 0))
  const int tap_r1 = g.Wp, tap_r2 = 2 * g.Wp;

  i32x16 acc[2][PB];
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int jp = 0; jp < PB; ++jp)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[jc][jp][i] = 0;

  // ---- prologue: chunk 0's tile, the first two slabs ----
  static_for<HPW>([&](auto i) { issue_halo(i, 0); });
  issue_w(std::integral_constant<int, 0>{}, w_tap);
  issue_w(std::integral_constant<int, 1>{}, w_tap);

  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    const int8_t* const hb = halo + (c & 1) * HALO;
    static_for<9>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      constexpr int U = t % NBUF;
      // this step's slab (and, at t = 0, this chunk's tile) must have landed; what the previous step issued stays in flight:
      // its slab (unless this is the very last step) and, for 1 <= t <= HPW with another chunk to come, one halo piece
      constexpr bool HPREV = t >= 1 && t <= HPW;
      auto wait = [&](auto nb_c) {
        constexpr int nb = decltype(nb_c)::value;
        if (t == 8 && !more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (HPREV && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nb + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nb) : "memory");
      };
      if (BPW == 8 || wave < BPW) wait(std::integral_constant<int, 1>{});
      else wait(std::integral_constant<int, 0>{});
      __builtin_amdgcn_s_barrier();
      if (t == 0 && c == 0 && tid < BN) {
        // the per-channel constants have landed: (s_w, SUM qw) -> (s_in s_w, (shift - zp) SUM qw) in place, once per channel
        float* pf = reinterpret_cast<float*>(par) + tid;
        int* pi = reinterpret_cast<int*>(par) + BN + tid;
        *pf = sin_early * *pf;
        *pi = (shift - zpi) * *pi;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      // the slab two steps ahead goes into the slot step - 1 released; the next chunk's tile into the other halo buffer
      if (t < 7 || more) issue_w(std::integral_constant<int, (U + 2) % NBUF>{}, t == 6 ? w_chunk : w_tap);
      if constexpr (t < HPW) {
        if (more) issue_halo(t_c, (c + 1) & 1);
      }
      constexpr int r = t / 3, s = t % 3;
      const int p0 = pbase + (r == 0 ? 0 : (r == 1 ? tap_r1 : tap_r2)) + s;
      const int8_t* const sb = ring + U * SLAB;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        i32x4 wf[2], pf[PB];
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) wf[jc] = *reinterpret_cast<const i32x4*>(sb + (woff[jc] ^ (ks << 5)));
#pragma unroll
        for (int jp = 0; jp < PB; ++jp) {
          const int p = p0 + jp * 32;
          const i32x4 v = *reinterpret_cast<const i32x4*>(hb + p * 64 + (((ks * 2 + hsel) ^ ((p >> 2) & 3)) << 4));
          pf[jp] = i32x4{(int)(v.x ^ xorw), (int)(v.y ^ xorw), (int)(v.z ^ xorw), (int)(v.w ^ xorw)};
        }
#pragma unroll
        for (int jc = 0; jc < 2; ++jc)
#pragma unroll
          for (int jp = 0; jp < PB; ++jp) acc[jc][jp] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[jc], pf[jp], acc[jc][jp], 0, 0, 0);
      }
    });
  }

  // ---- epilogue: lane (p = l31, h = hsel), register i of block (jc, jp) = channel n0 + wc * 64 + jc * 32 + 16 h + i of row
  // q0 + wp * PW + jp * 32 + p.  Dequantise, quantise for the consumer, stage the code tile, store whole rows. ----
  __builtin_amdgcn_s_barrier();                   // every wave is done with the operand buffers (the code tile is staged there)
  const EpiQuant eq(ep, ep.relu != 0);            // code(relu(v)) = max(code(v), code(0))
  int8_t* const stg = lds;
#pragma unroll
  for (int jc = 0; jc < 2; ++jc) {
    const int cb = wc * 64 + jc * 32 + hsel * 16;
#pragma unroll
    for (int jp = 0; jp < PB; ++jp) {
      f32x4 y[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (cb + 4 * q) * 4);
        const i32x4 co = *reinterpret_cast<const i32x4*>(par + (BN + cb + 4 * q) * 4);
        const f32x4 bs = bias ? *reinterpret_cast<const f32x4*>(par + (2 * BN + cb + 4 * q) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        y[q] = f32x4{dequant1(acc[jc][jp][4 * q] + co.x, mu.x, bs.x), dequant1(acc[jc][jp][4 * q + 1] + co.y, mu.y, bs.y),
                     dequant1(acc[jc][jp][4 * q + 2] + co.z, mu.z, bs.z), dequant1(acc[jc][jp][4 * q + 3] + co.w, mu.w, bs.w)};
      }
      uint32_t wq[4];
      bool uq[4];
      eq.code4n(y, wq, uq);
      *reinterpret_cast<i32x4*>(stg + (wp * PW + jp * 32 + l31) * SROW + cb) = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
    }
  }
  __syncthreads();                                // a row's BN bytes come from WC waves
  constexpr int LPR = BN / 16;                    // lanes per row
  constexpr int RPP = 512 / LPR;                  // rows per pass
  const int srow = tid / LPR, sseg = tid % LPR;
#pragma unroll
  for (int it = 0; it < TM / RPP; ++it) {
    const int rr = it * RPP + srow;
    const uint32_t q = q0 + (uint32_t)rr;
    const uint32_t n = fdiv(q, g.fsdiv);
    const uint32_t rem = q - n * (uint32_t)g.FS;
    const uint32_t yy = fdiv(rem, g.wpdiv);
    const uint32_t xx = rem - yy * (uint32_t)g.Wp;
    const i32x4 c16 = *reinterpret_cast<const i32x4*>(stg + rr * SROW + sseg * 16);
    if (q < g.MQ && yy < (uint32_t)g.H && xx < (uint32_t)g.W)
      __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(ep.codes + ((int64_t)((n * (uint32_t)g.H + yy) * (uint32_t)g.W + xx)) * g.K + n0 +
                                                                sseg * 16));
  }
}

// Whether the halo kernel takes this layer (conv_launch asks before it picks a generic tile), and the launch.
bool conv3x3_halo_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                          int32_t dilation, const ConvEpi& ep, const float* out, bool dual) {
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || dilation != 1 || dual || out || ep.residual || ep.w_off || !ep.codes) return false;
  if (C % 64 != 0 || K % 128 != 0 || !aligned16(ep.codes)) return false;
  if (W + 1 > 120 || N * (H + 1) * (W + 1) + 1024 >= (1ll << 31) || N * H * W * C >= (1ll << 31)) return false;
  return true;
}

int conv3x3_halo_launch(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                        const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                        int shift, const ConvEpi& ep, hipStream_t st) {
  constexpr int TM = 256;
  HaloGeom g;
  g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K;
  g.Wp = (int)W + 1;
  g.FS = (int)((H + 1) * (W + 1));
  g.MQ = (uint32_t)(N * g.FS);
  g.nblk_n = (int)(K / 128);
  g.hp = (TM + 2 * g.Wp + 2 + 15) / 16;
  g.fsdiv = make_fastdiv((uint32_t)g.FS);
  g.wpdiv = make_fastdiv((uint32_t)g.Wp);
  const int64_t nblk_m = ((int64_t)g.MQ + TM - 1) / TM;
  const int64_t nwg = nblk_m * g.nblk_n;
  if (nwg >= (1ll << 31)) return DLMCQ_ERANGE;
#define DLMCQ_HALO_ARGS dim3((uint32_t)nwg), dim3(512), 0, st, x, w, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep
  if (g.hp <= 24) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<128, TM, 3>), DLMCQ_HALO_ARGS);
  else if (g.hp <= 32) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<128, TM, 4>), DLMCQ_HALO_ARGS);
  else return DLMCQ_EINVAL;
#undef DLMCQ_HALO_ARGS
  return launch_status();
}

}  // namespace dlmcq
