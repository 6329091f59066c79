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
//     and one counted s_waitcnt vmcnt per step (as conv_i8.hip; LDS-DMA with the XOR swizzle on the source side); the
//     loop is software-pipelined by half a step, so that barrier, fragment reads and DMA requests issue between MFMAs.
//     A 256-row tile halves the slab bytes per MAC of the 128-row kernel.
//   * v_mfma_i32_32x32x32_i8 with the weights as A and the pixels as B: a lane's 16 accumulator registers of a block are 16
//     consecutive channels of one pixel - conv_i8.hip's swapped epilogue, ~10 vector instructions per output element.
//     Wave tiling (template): 4 waves of 64 pixels x 128 channels (2 x 4 blocks: 6 fragment reads, 16 MFMAs per K half-step;
//     two workgroups = two INDEPENDENT waves per SIMD), or 8 waves of 64 x 64 (2 x 2 blocks; round 3's first version:
//     its fragment reads, xors and address arithmetic per MFMA are twice as many and its two waves per SIMD run in lockstep).
//   Vector-memory bytes per K step: 8 KB of weights + 1/9 of a ~20 KB tile for 512 clocks of matrix work per CU: ~20 B/clock.
//
//   * STRIDE 2 (S2; a ResNet stage's first 3x3, RepVGG's stage openers).  Split the input into its four PHASE images (even / odd
//     rows x even / odd columns): output (y, x) reads input rows 2y-1, 2y, 2y+1, i.e. phase row y-1 of the odd rows, y of the even
//     rows, y of the odd rows - in every phase image the convolution is a stride-1 stencil again, with shifts of 0 or 1 rows and 0
//     or 1 columns.  The frame is that of the OUTPUT ((P + 1) x (Q + 1), border above and left); tap (r, s) reads phase
//     (r != 1, s != 1) at frame position q + (r > 0) Wp + (s > 0).  A chunk has four phase tiles of TM + Wp + 2 positions; they
//     stream through a ring of THREE tile buffers the way the weight slabs do: the nine K steps run phase by phase
//     (taps 4 | 3 5 | 1 7 | 0 2 6 8) and the tile three ahead is requested as soon as a buffer's last tap has read it.
#include "conv_i8_common.h"

namespace dlmcq {

struct HaloGeom {
  int N, H, W, C, K;     // H, W: the OUTPUT's (= the input's for stride 1)
  int Hin, Win;          // the input's
  int Wp, FS;            // W + 1, (H + 1) (W + 1)
  uint32_t MQ;           // N FS: rows of the linear frame space
  int nblk_n;
  int hp;                // halo pieces (16 frame positions each) a chunk's tile needs: ceil((TM + 2 Wp + 2) / 16)
  FastDiv fsdiv, wpdiv;
};

// LAB (lab library only; results are garbage except for 0 and 1, only the time means something): 1 = clock stamps into `trace`;
// 2 = no weight DMA, 3 = no halo DMA, 4 = no MFMAs, 6 = no xor of the pixel fragments, 7 = no quantising epilogue,
// 8 = codes stored straight from the accumulator layout (32-byte pieces, no staging through LDS)
// K step t of a chunk: which tap's weights (halo_tap), which halo tile (halo_ph: always 0 for stride 1), which shift of it
// (halo_rs rows, halo_cs columns).  Stride 1: taps in order, shifts (r, s).  Stride 2: phase by phase - taps 4 | 3 5 | 1 7 | 0 2 6 8 -
// tap (r, s) reading phase (r != 1, s != 1) at shift (r > 0, s > 0).
constexpr int halo_tap(bool s2, int t) { return !s2 ? t : (t == 0 ? 4 : t == 1 ? 3 : t == 2 ? 5 : t == 3 ? 1 : t == 4 ? 7 : t == 5 ? 0 : t == 6 ? 2 : t == 7 ? 6 : 8); }
constexpr int halo_ph(bool s2, int t) { return !s2 ? 0 : (t == 0 ? 0 : t <= 2 ? 1 : t <= 4 ? 2 : 3); }
constexpr int halo_rs(bool s2, int t) { return !s2 ? t / 3 : (halo_tap(true, t) / 3 > 0 ? 1 : 0); }
constexpr int halo_cs(bool s2, int t) { return !s2 ? t % 3 : (halo_tap(true, t) % 3 > 0 ? 1 : 0); }

// NHB = halo buffers: 2, or 1 for layers with ONE 64-channel chunk (C = 64: nothing to prefetch; the LDS saved buys a third
// workgroup per CU, whose K loop covers the others' quantising epilogues - at 9 K steps per tile those are most of a tile's life).
// WPE = waves per SIMD the registers are budgeted for.  XS = the activation codes are uint8 and every pixel fragment is re-centred
// (^ 0x80) on its way to the matrix cores; producers that emit `code - 128` (DLMCQ_EMIT_SHIFT128) spare this kernel 16 of its ~45
// vector instructions per K step.
// PLAIN (the launcher has checked epi_plain): the consumer's quantiser is the plain unsigned-byte one - the epilogue on channel pairs with
// EpiQuant::code4n_plain (conv_epilogue.h; 6.75 instead of ~9 vector instructions per element), the same bytes.
template <int BN, int TM, int NW, int WC, int HPW, int NHB, int WPE, bool XS, int LAB = 0, bool S2 = false, int HPA = HPW * NW, bool PLAIN = false>
__global__ __launch_bounds__(NW * 64, WPE) void conv3x3_halo_i8_kernel(
    const int8_t* __restrict__ x, const int8_t* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ wsum,
    const float* __restrict__ s_in, const float* __restrict__ zp_in, const float* __restrict__ s_w, HaloGeom g, int shift, ConvEpi ep,
    unsigned long long* __restrict__ trace) {
  constexpr bool STAMP = LAB == 1;
  // LAB 9 / 10 / 11 (timing only, round 4): the workgroups that take a CU's SECOND slot in the first round start half / a quarter / a
  // whole tile life late, so that the two workgroups of a CU are out of phase (one in its matrix-bound K loop, one in its vector-bound
  // epilogue) - what a phase-shifted pair is worth before anything is built for it
  if (LAB >= 9 && LAB <= 11 && blockIdx.x >= 256 && blockIdx.x < 512) {
    for (int i = 0; i < (LAB == 9 ? 4 : LAB == 10 ? 2 : 8); ++i) __builtin_amdgcn_s_sleep(127);
  }
  // LAB 12 / 13 (timing only): four workgroups per CU (the 64-channel layers) - the first round's slot s (blockIdx / 256) starts s x 8 k / s x 4 k clocks late
  if ((LAB == 12 || LAB == 13) && blockIdx.x < 1024) {
    for (int i = 0; i < (int)((blockIdx.x >> 8) & 3u); ++i) {
      if (LAB == 12) __builtin_amdgcn_s_sleep(127);
      else __builtin_amdgcn_s_sleep(63);
    }
  }
  const unsigned long long t_start = STAMP ? __builtin_readcyclecounter() : 0ull;
  constexpr int NT = NW * 64;               // threads
  constexpr int NBUF = 3;
  constexpr int SLAB = BN * 64;             // one (tap, chunk) of the weights
  constexpr int WPX = NW / WC;              // waves across the pixels
  constexpr int PW = TM / WPX;              // pixels per wave
  constexpr int CW = BN / WC;               // channels per wave
  constexpr int PB = PW / 32, CB = CW / 32; // 32 x 32 MFMA blocks per wave
  constexpr int HP = HPA;                   // halo pieces allocated per buffer (HPW per wave are requested; surplus ones re-load the last)
  constexpr int HALO = HP * 1024;
  constexpr int NBW = SLAB / 1024 / NW;     // weight pieces per wave per step
  constexpr int RING = NBUF * SLAB;
  constexpr int SROW = BN + 16;             // staged code row (conflict-free 16-byte accesses)
  constexpr int STAGE = TM * SROW;
  constexpr int OPER = RING + NHB * HALO;
  constexpr int LDS_BYTES = OPER < STAGE ? STAGE : OPER;
  constexpr int PAR_BYTES = 3 * BN * 4;
  static_assert(PB >= 1 && CB >= 1 && NBW >= 1 && NBW * NW * 1024 == SLAB && HPW <= 8 && BN % 64 == 0 && (S2 ? NHB == 3 : (NHB == 1 || NHB == 2)) && HPA <= HPW * NW, "tile shape");
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
  // STAMP: wave 0 of every workgroup writes start / loop entered / loop left / end (+ the constant 100 MHz clock at start and end) at trace[64 * 8 + 6 * blockIdx.x],
  // and wave 0 of the workgroup in the middle of the grid the phase boundaries of its first 64 steps at trace[step * 8 + k]
  unsigned long long* const wgt = (STAMP && tid == 0) ? trace + 64 * 8 + 6 * (size_t)blockIdx.x : nullptr;
  unsigned long long* const stp = (STAMP && tid == 0 && blockIdx.x == gridDim.x / 2) ? trace : nullptr;
  if (STAMP && wgt) {
    wgt[0] = t_start;
    wgt[4] = __builtin_amdgcn_s_memrealtime();              // the 100 MHz constant clock: shader clock = d(memtime) / d(memrealtime) x 100 MHz
  }
  int stepno = 0;
  auto stamp = [&](int k) {
    if (STAMP && stp && stepno < 64) stp[stepno * 8 + k] = __builtin_readcyclecounter();
  };

  // ---- per-channel constants of the epilogue: LDS-DMA into a table behind everything, requested first ----
  {
    const void* arrs[3] = {s_w, wsum, bias};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!arrs[a]) continue;
#pragma unroll
      for (int c = 0; c < BN / 64; ++c) {
        if (((a * (BN / 64) + c) % NW) != wave) continue;
        __builtin_amdgcn_global_load_lds((gptr_t)(static_cast<const int32_t*>(arrs[a]) + n0 + c * 64 + lane),
                                         (lptr_t)(par + (a * BN + c * 64) * 4), 4, 0, 0);
      }
    }
  }
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin_early = s_in[0];
  const uint32_t xorw = 0x80808080u;
  const int8_t* const padline = g_pad_table.b + ((zpi & 0xff) << 6);     // stored UNshifted: the xor happens on read

  // ---- halo DMA: piece i of this wave covers halo positions (i * NW + wave) * 16 .. + 15; LDS slot s of position p holds the
  // logical 16-byte segment s ^ ((p >> 2) & 3) (swizzle on the source side, undone by the fragment reads) ----
  const int lrow = lane >> 2, pslot = lane & 3;
  const int8_t* hsrc[HPW];
  int hinc[HPW];     // stride 1: bytes to the next chunk (64, or 0 at a border); stride 2: all ones for a pixel, 0 for a border (a mask)
  int hpc[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    int pc = i * NW + wave;
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
    if constexpr (S2) {   // phase (0, 0)'s pixel of this frame position: input (2 (fy - 1), 2 (fx - 1)); the other phases are + (a Win + b) C
      hsrc[i] = in ? x + ((int64_t)((n * (uint32_t)g.Hin + 2u * (fy - 1u)) * (uint32_t)g.Win + 2u * (fx - 1u))) * g.C + seg * 16 : padline;
      hinc[i] = in ? -1 : 0;
    } else {
      hsrc[i] = in ? x + ((int64_t)((n * (uint32_t)g.H + fy - 1u) * (uint32_t)g.W + fx - 1u)) * g.C + seg * 16 : padline;
      hinc[i] = in ? 64 : 0;
    }
  }
  auto issue_halo = [&](auto i_c, int buf) {
    constexpr int i = decltype(i_c)::value;
    if (LAB != 3) __builtin_amdgcn_global_load_lds((gptr_t)hsrc[i], (lptr_t)(halo + buf * HALO + hpc[i] * 1024), 16, 0, 0);
    hsrc[i] += hinc[i];
  };
  // stride 2: the whole tile of phase `ph` (0: even rows / even columns, 1: even / odd, 2: odd / even, 3: odd / odd) of chunk `c`
  const int ph_row = g.Win * g.C, ph_col = g.C;
  auto issue_tile = [&](int ph, int c, int buf) {
    const int off = (ph >> 1) * ph_row + (ph & 1) * ph_col + c * 64;
    static_for<HPW>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      if (LAB != 3) __builtin_amdgcn_global_load_lds((gptr_t)(hsrc[i] + (off & hinc[i])), (lptr_t)(halo + buf * HALO + hpc[i] * 1024), 16, 0, 0);
    });
  };

  // ---- weight DMA: piece j of this wave moves slab rows (j * NW + wave) * 16 .. + 15 of every step; slab row d of a 32-row block
  // holds channel 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3), which makes a lane's 16 accumulator registers 16 consecutive channels ----
  const int8_t* wsrc[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int drow = (j * NW + wave) * 16 + lrow, d = drow & 31;
    const int k = n0 + (drow & ~31) + 16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
    wsrc[j] = w + (int64_t)k * (9 * g.C) + halo_tap(S2, 0) * g.C + (pslot ^ ((drow >> 2) & 3)) * 16;
  }
  const int nchunks = g.C >> 6;
  // wsrc always points at the next slab to request; after the slab of K step u (tap TAP[u]) it moves to TAP[u + 1]'s, or to
  // TAP[0] of the next chunk
  auto issue_w = [&](auto slot_c, auto u_c) {
    constexpr int SL = decltype(slot_c)::value, u = decltype(u_c)::value;
    constexpr int dtap = u == 8 ? halo_tap(S2, 0) - halo_tap(S2, 8) : halo_tap(S2, (u + 1) % 9) - halo_tap(S2, u);
    const int inc = dtap * g.C + (u == 8 ? 64 : 0);
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      if (LAB != 2) __builtin_amdgcn_global_load_lds((gptr_t)wsrc[j], (lptr_t)(ring + SL * SLAB + (j * NW + wave) * 1024), 16, 0, 0);
      wsrc[j] += inc;
    }
  };

  // ---- fragment addresses ----
  // weights (MFMA A): rows wc * CW + jc * 32 + l31 of the slab (the swizzle term (d >> 2) & 3 does not depend on jc); the two
  // K halves of a step are slots (ks * 2 + hsel) ^ swz
  int woff[2];
  {
    const int d = wc * CW + l31;
    woff[0] = d * 64 + ((hsel ^ ((d >> 2) & 3)) << 4);
    woff[1] = woff[0] ^ 32;
  }
  const int pbase = wp * PW + l31;             // this lane's pixel row of block 0 (halo position of tap (0, 0))
  const int tap_r1 = g.Wp, tap_r2 = 2 * g.Wp;

  i32x16 acc[CB][PB];
#pragma unroll
  for (int jc = 0; jc < CB; ++jc)
#pragma unroll
    for (int jp = 0; jp < PB; ++jp)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[jc][jp][i] = 0;

  // ---- the K loop, software-pipelined by half a step.  Fragment set A holds K half 0 of a step, set B half 1.  Step s:
  //   phase 0:  read B(s);                                   8 MFMAs on A(s)
  //   phase 1:  wait for slab s + 1, barrier, read A(s + 1), request slab s + 3 (+ a piece of the next chunk's tile);
  //             8 MFMAs on B(s)
  // so every LDS read, DMA request and barrier is issued between MFMAs that are already fed. ----
  i32x4 wfA[CB], pfA[PB], wfB[CB], pfB[PB];
  auto read_frags = [&](auto u_c, auto t_c, auto ks_c, const int8_t* hbuf, i32x4 (&wf)[CB], i32x4 (&pf)[PB]) {
    constexpr int U = decltype(u_c)::value, t = decltype(t_c)::value, ks = decltype(ks_c)::value;
    const int p0 = pbase + (halo_rs(S2, t) == 0 ? 0 : (halo_rs(S2, t) == 1 ? tap_r1 : tap_r2)) + halo_cs(S2, t);
    const int pa = (p0 * 64 + ((hsel ^ ((p0 >> 2) & 3)) << 4)) ^ (ks << 5);     // ((p + 32 jp) >> 2) & 3 = (p >> 2) & 3
    const int8_t* const sb = ring + U * SLAB;
#pragma unroll
    for (int jc = 0; jc < CB; ++jc) wf[jc] = *reinterpret_cast<const i32x4*>(sb + woff[ks] + jc * 2048);
#pragma unroll
    for (int jp = 0; jp < PB; ++jp) pf[jp] = *reinterpret_cast<const i32x4*>(hbuf + pa + jp * 2048);
  };
  auto multiply = [&](const i32x4 (&wf)[CB], i32x4 (&pf)[PB]) {
    if (XS && LAB != 6) {
#pragma unroll
      for (int jp = 0; jp < PB; ++jp)
        pf[jp] = i32x4{(int)(pf[jp].x ^ xorw), (int)(pf[jp].y ^ xorw), (int)(pf[jp].z ^ xorw), (int)(pf[jp].w ^ xorw)};
    }
#pragma unroll
    for (int jc = 0; jc < CB; ++jc)
#pragma unroll
      for (int jp = 0; jp < PB; ++jp) {
        if (LAB == 4) asm volatile("" ::"v"(wf[jc]), "v"(pf[jp]));
        else acc[jc][jp] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[jc], pf[jp], acc[jc][jp], 0, 0, 0);
      }
  };
  constexpr int nbw = (LAB == 2 ? 0 : NBW);

  // prologue: chunk 0's tile (stride 2: its first three phase tiles) and the first three slabs; the first wait leaves slabs 1 and 2 in flight
  if constexpr (S2) {
    issue_tile(0, 0, 0);
    issue_tile(1, 0, 1);
    issue_tile(2, 0, 2);
  } else {
    static_for<HPW>([&](auto i) { issue_halo(i, 0); });
  }
  issue_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  issue_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  issue_w(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
  if (STAMP && wgt) wgt[1] = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * nbw) : "memory");
  if (tid < BN) {
    // the per-channel constants have landed: (s_w, SUM qw) -> (s_in s_w, (shift - zp) SUM qw) in place, once per channel
    float* pf = reinterpret_cast<float*>(par) + tid;
    int* pi = reinterpret_cast<int*>(par) + BN + tid;
    *pf = sin_early * *pf;
    *pi = (shift - zpi) * *pi;
  }
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, halo, wfA, pfA);

  for (int c = 0; c < (NHB == 1 ? 1 : nchunks); ++c) {
    const bool more = NHB != 1 && c + 1 < nchunks;
    // stride 1: the chunk's one tile, double-buffered; stride 2: phase tile p of chunk c lives in buffer (c + p) % 3
    const int b0 = S2 ? c % 3 : (NHB == 2 ? (c & 1) : 0);
    auto tile_of = [&](int ph, bool next_chunk) -> const int8_t* {
      if constexpr (S2) {
        int b = b0 + ph + (next_chunk ? 1 : 0);
        b = b >= 6 ? b - 6 : (b >= 3 ? b - 3 : b);
        return halo + b * HALO;
      } else {
        return halo + (NHB == 2 ? ((next_chunk ? c + 1 : c) & 1) * HALO : 0);
      }
    };
    static_for<9>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      constexpr int U = t % NBUF;
      stamp(0);
      // phase 0
      read_frags(std::integral_constant<int, U>{}, t_c, std::integral_constant<int, 1>{}, tile_of(halo_ph(S2, t), false), wfB, pfB);
      multiply(wfA, pfA);
      stamp(1);
      // phase 1
      if (t < 8 || more) {
        // slab s + 1 (requested three steps ago) must have landed; younger and allowed to stay in flight: slab s + 2 (unless the
        // loop ends before it) and whatever halo pieces step s - 1 requested in front of it.  Stride 1: one piece per wave in steps
        // 0 .. HPW - 1 of a chunk with a successor (never counted at t = 8: the next step reads the new tile, so everything but the
        // youngest slab must be there).  Stride 2: a whole tile (HPW pieces) in steps 0 (always: this chunk's fourth phase), 2, 4 and
        // 8 (the next chunk's first three, if there is one).  A tile is read two or more steps after its request, i.e. behind a
        // wait that has retired it.  This wave's reads of slab s are complete (lgkmcnt) before the barrier lets anyone overwrite it.
        constexpr int hw = (LAB == 3 ? 0 : (S2 ? HPW : 1));
        if constexpr (S2) {
          if (t >= 7 && !more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else if (t == 1 || (t == 0 && c > 0) || ((t == 3 || t == 5) && more))
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(nbw + hw) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(nbw) : "memory");
        } else {
          constexpr bool HPREV = t >= 1 && t <= HPW && t <= 7;
          if (t >= 7 && !more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else if (HPREV && more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(nbw + hw) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(nbw) : "memory");
        }
        stamp(2);
        read_frags(std::integral_constant<int, (U + 1) % NBUF>{}, std::integral_constant<int, (t + 1) % 9>{}, std::integral_constant<int, 0>{},
                   tile_of(halo_ph(S2, (t + 1) % 9), t == 8), wfA, pfA);
        // requests: halo pieces first (see the wait above), then slab s + 3 into the slot of slab s
        if constexpr (S2) {
          // the buffer whose last tap has just been read takes the tile three ahead: T(c, 3) at step 0, T(c + 1, 0 / 1 / 2) at 2 / 4 / 8
          if constexpr (t == 0) issue_tile(3, c, b0);
          if constexpr (t == 2 || t == 4 || t == 8) {
            if (more) issue_tile(t == 2 ? 0 : (t == 4 ? 1 : 2), c + 1, (b0 + (t == 2 ? 1 : (t == 4 ? 2 : 0))) % 3);
          }
        } else if constexpr (t < HPW && NHB == 2) {
          if (more) issue_halo(t_c, (c + 1) & 1);
        }
        if (t < 6 || more) issue_w(std::integral_constant<int, U>{}, std::integral_constant<int, (t + 3) % 9>{});
        stamp(3);
      }
      multiply(wfB, pfB);
      if (STAMP) {
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[CB - 1][PB - 1][15]));
        stamp(4);
        ++stepno;
      }
    });
  }
  if (STAMP && wgt) wgt[2] = __builtin_readcyclecounter();

  // ---- epilogue: lane (p = l31, h = hsel), register i of block (jc, jp) = channel n0 + wc * CW + jc * 32 + 16 h + i of row
  // q0 + wp * PW + jp * 32 + p.  Dequantise, quantise for the consumer, stage the code tile, store whole rows. ----
  asm volatile("s_barrier" ::: "memory");         // every wave is done with the operand buffers (the code tile is staged there)
  const EpiQuant eq(ep, ep.relu != 0);            // code(relu(v)) = max(code(v), code(0))
  int8_t* const stg = lds;
  auto row_addr = [&](int rr, bool& ok) -> uint8_t* {   // tile row -> the pixel's K code bytes (ok = it is a pixel of the batch)
    const uint32_t q = q0 + (uint32_t)rr;
    const uint32_t n = fdiv(q, g.fsdiv);
    const uint32_t rem = q - n * (uint32_t)g.FS;
    const uint32_t yy = fdiv(rem, g.wpdiv);
    const uint32_t xx = rem - yy * (uint32_t)g.Wp;
    ok = q < g.MQ && yy < (uint32_t)g.H && xx < (uint32_t)g.W;
    return ep.codes + ((int64_t)((n * (uint32_t)g.H + yy) * (uint32_t)g.W + xx)) * g.K + n0;
  };
#pragma unroll
  for (int jp = 0; jp < PB; ++jp) {
    bool ok8 = false;
    uint8_t* dst8 = nullptr;
    if (LAB == 8) dst8 = row_addr(wp * PW + jp * 32 + l31, ok8);
#pragma unroll
    for (int jc = 0; jc < CB; ++jc) {
      const int cb = wc * CW + jc * 32 + hsel * 16;
      f32x4 y[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (cb + 4 * q) * 4);
        const i32x4 co = *reinterpret_cast<const i32x4*>(par + (BN + cb + 4 * q) * 4);
        const f32x4 bs = bias ? *reinterpret_cast<const f32x4*>(par + (2 * BN + cb + 4 * q) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (PLAIN) {
          const f32x2 ya = pk_fma(f32x2{(float)(acc[jc][jp][4 * q] + co.x), (float)(acc[jc][jp][4 * q + 1] + co.y)}, f32x2{mu.x, mu.y}, f32x2{bs.x, bs.y});
          const f32x2 yb = pk_fma(f32x2{(float)(acc[jc][jp][4 * q + 2] + co.z), (float)(acc[jc][jp][4 * q + 3] + co.w)}, f32x2{mu.z, mu.w}, f32x2{bs.z, bs.w});
          y[q] = f32x4{ya.x, ya.y, yb.x, yb.y};
        } else {
          y[q] = f32x4{dequant1(acc[jc][jp][4 * q] + co.x, mu.x, bs.x), dequant1(acc[jc][jp][4 * q + 1] + co.y, mu.y, bs.y),
                       dequant1(acc[jc][jp][4 * q + 2] + co.z, mu.z, bs.z), dequant1(acc[jc][jp][4 * q + 3] + co.w, mu.w, bs.w)};
        }
      }
      uint32_t wq[4];
      bool uq[4];
      if (LAB == 7) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[q] = __builtin_bit_cast(uint32_t, y[q].x + y[q].y + y[q].z + y[q].w);
      } else if constexpr (PLAIN) eq.code4n_plain(y, wq);
      else eq.code4n(y, wq, uq);
      const i32x4 c16 = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
      if (LAB == 8) {
        if (ok8) __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(dst8 + cb));
      } else *reinterpret_cast<i32x4*>(stg + (wp * PW + jp * 32 + l31) * SROW + cb) = c16;
    }
  }
  if (LAB != 8) {
    constexpr int LPR = BN / 16;                    // lanes per row
    if constexpr (WC == 1) {                        // a wave's rows are its own: no block barrier, just the LDS counter
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int RPP = 64 / LPR;
      const int srow = lane / LPR, sseg = lane % LPR;
#pragma unroll
      for (int it = 0; it < PW / RPP; ++it) {
        const int rr = wp * PW + it * RPP + srow;
        bool ok;
        uint8_t* const dst = row_addr(rr, ok);
        const i32x4 c16 = *reinterpret_cast<const i32x4*>(stg + rr * SROW + sseg * 16);
        if (ok) __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(dst + sseg * 16));
      }
    } else {
      __syncthreads();                              // a row's BN bytes come from WC waves
      constexpr int RPP = NT / LPR;                 // rows per pass
      const int srow = tid / LPR, sseg = tid % LPR;
#pragma unroll
      for (int it = 0; it < TM / RPP; ++it) {
        const int rr = it * RPP + srow;
        bool ok;
        uint8_t* const dst = row_addr(rr, ok);
        const i32x4 c16 = *reinterpret_cast<const i32x4*>(stg + rr * SROW + sseg * 16);
        if (ok) __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(dst + sseg * 16));
      }
    }
  }
  if (STAMP && wgt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wgt[3] = __builtin_readcyclecounter();
    wgt[5] = __builtin_amdgcn_s_memrealtime();
  }
}

// Whether the halo kernel takes this layer (conv_launch asks before it picks a generic tile), and the launch.
bool conv3x3_halo_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                          int32_t dilation, const ConvEpi& ep, const float* out, bool dual) {
  if (R != 3 || S != 3 || (stride != 1 && stride != 2) || pad != 1 || dilation != 1 || dual || out || ep.residual || ep.w_off || !ep.codes)
    return false;
  if (C % 64 != 0 || K % 64 != 0 || !aligned16(ep.codes)) return false;
  if (stride == 2 && ((H | W) & 1)) return false;                      // (odd sizes: the generic kernel)
  const int64_t P = H / stride, Q = W / stride;                        // pad 1, 3 x 3: P = H for stride 1, H / 2 for even H at stride 2
  if (stride == 2 && Q + 1 > 62) return false;                         // a phase tile of 256 + Wp + 2 positions in 20 pieces
  // measured (plan profiles, batch 512): 256 -> 256 at 28^2 -> 14^2 110 -> 86 us, 512 -> 512 at 14^2 -> 7^2 88 -> 75, 128 -> 128 at
  // 56^2 -> 28^2 122 -> 115, 64 -> 64 at 112^2 -> 56^2 210 -> 197; a single chunk feeding 128 channels (64 -> 128 at 56^2 -> 28^2) is
  // all prologue: 80 -> 87, left to the generic kernel
  if (stride == 2 && C == 64 && K != 64) return false;
  if (Q + 1 > 120 || N * (P + 1) * (Q + 1) + 1024 >= (1ll << 31) || N * H * W * C >= (1ll << 31)) return false;
  return true;
}

int conv3x3_halo_launch(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                        const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                        int32_t stride, int shift, const ConvEpi& ep, hipStream_t st, int lab, void* lab_trace) {
  constexpr int TM = 256;
  const int bn = K % 128 == 0 ? 128 : 64;
  const bool s2 = stride == 2;
  const bool plain = epi_plain(ep);
  HaloGeom g;
  g.N = (int)N; g.Hin = (int)H; g.Win = (int)W; g.H = (int)(H / stride); g.W = (int)(W / stride); g.C = (int)C; g.K = (int)K;
  g.Wp = g.W + 1;
  g.FS = (g.H + 1) * (g.W + 1);
  g.MQ = (uint32_t)(N * g.FS);
  g.nblk_n = (int)(K / bn);
  g.hp = (TM + (s2 ? 1 : 2) * g.Wp + 2 + 15) / 16;
  g.fsdiv = make_fastdiv((uint32_t)g.FS);
  g.wpdiv = make_fastdiv((uint32_t)g.Wp);
  const int64_t nblk_m = ((int64_t)g.MQ + TM - 1) / TM;
  const int64_t nwg = nblk_m * g.nblk_n;
  if (nwg >= (1ll << 31)) return DLMCQ_ERANGE;
  unsigned long long* const trace = static_cast<unsigned long long*>(lab_trace);
#define DLMCQ_HALO_ARGS(NW) dim3((uint32_t)nwg), dim3(NW * 64), 0, st, x, w, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, trace
#ifdef DLMCQ_LAB
  if (lab) {     // lab: variant = lab % 100 on the product tiling, + 100 for the 8-wave tiling (100 = its product code); stride 1 only
    const int v = lab % 100;
    if (s2 || g.hp > 24 || (v == 1 && !trace)) return DLMCQ_EINVAL;
    if (lab >= 100) {
      if (bn != 128) return DLMCQ_EINVAL;
      switch (v) {
#define DLMCQ_HALO_LAB(V) case V: hipLaunchKernelGGL((conv3x3_halo_i8_kernel<128, TM, 8, 2, 3, 2, 4, true, V>), DLMCQ_HALO_ARGS(8)); break
        DLMCQ_HALO_LAB(0); DLMCQ_HALO_LAB(1); DLMCQ_HALO_LAB(4);
#undef DLMCQ_HALO_LAB
        default: return DLMCQ_EINVAL;
      }
    } else if (bn == 128) {
      switch (v) {
#define DLMCQ_HALO_LAB(V) case V: hipLaunchKernelGGL((conv3x3_halo_i8_kernel<128, TM, 4, 1, 6, 2, 2, true, V>), DLMCQ_HALO_ARGS(4)); break
        DLMCQ_HALO_LAB(1); DLMCQ_HALO_LAB(2); DLMCQ_HALO_LAB(3); DLMCQ_HALO_LAB(4); DLMCQ_HALO_LAB(6); DLMCQ_HALO_LAB(7); DLMCQ_HALO_LAB(8);
        DLMCQ_HALO_LAB(9); DLMCQ_HALO_LAB(10); DLMCQ_HALO_LAB(11);
#undef DLMCQ_HALO_LAB
        default: return DLMCQ_EINVAL;
      }
    } else {
      if (C != 64) return DLMCQ_EINVAL;
      switch (v) {
#define DLMCQ_HALO_LAB(V) case V: hipLaunchKernelGGL((conv3x3_halo_i8_kernel<64, TM, 4, 1, 6, 1, 4, true, V>), DLMCQ_HALO_ARGS(4)); break
        DLMCQ_HALO_LAB(1); DLMCQ_HALO_LAB(2); DLMCQ_HALO_LAB(3); DLMCQ_HALO_LAB(4); DLMCQ_HALO_LAB(6); DLMCQ_HALO_LAB(7); DLMCQ_HALO_LAB(8);
        DLMCQ_HALO_LAB(12); DLMCQ_HALO_LAB(13);
#undef DLMCQ_HALO_LAB
        default: return DLMCQ_EINVAL;
      }
    }
    return launch_status();
  }
#else
  if (lab) return DLMCQ_EINVAL;
#endif
#define DLMCQ_HALO_GO(NW, ...)                                                                            \
  do {                                                                                                   \
    if (shift) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<__VA_ARGS__>), DLMCQ_HALO_ARGS(NW));           \
    else hipLaunchKernelGGL((conv3x3_halo_i8_kernel<__VA_ARGS__>), DLMCQ_HALO_ARGS(NW));                 \
  } while (0)
  if (s2) {
    // stride 2: three phase-tile buffers of 18 pieces (Wp <= 30: two workgroups per CU at 128 channels) or 20 (Wp <= 62)
    if (g.hp > 20) return DLMCQ_EINVAL;
    const bool big = g.hp > 18;
#define DLMCQ_HALO_S2(BN_, HPA_, WPE_)                                                                                                         \
  do {                                                                                                                                        \
    if (shift && plain) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, 5, 3, WPE_, true, 0, true, HPA_, true>), DLMCQ_HALO_ARGS(4));  \
    else if (shift) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, 5, 3, WPE_, true, 0, true, HPA_>), DLMCQ_HALO_ARGS(4));            \
    else if (plain) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, 5, 3, WPE_, false, 0, true, HPA_, true>), DLMCQ_HALO_ARGS(4));     \
    else hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, 5, 3, WPE_, false, 0, true, HPA_>), DLMCQ_HALO_ARGS(4));                      \
  } while (0)
    if (bn == 128) {
      if (!big) DLMCQ_HALO_S2(128, 18, 2);
      else DLMCQ_HALO_S2(128, 20, 1);      // (87 KB of LDS: one workgroup per CU)
    } else {
      if (!big) DLMCQ_HALO_S2(64, 18, 2);
      else DLMCQ_HALO_S2(64, 20, 2);
    }
#undef DLMCQ_HALO_S2
    return launch_status();
  }
  if (g.hp > 32) return DLMCQ_EINVAL;
  const bool wide = g.hp > 24;        // images wider than 57 pixels: 8 halo pieces per wave, one workgroup per CU
#undef DLMCQ_HALO_GO
#define DLMCQ_HALO_GO(BN_, HPW_, NHB_, WPE_)                                                                                                              \
  do {                                                                                                                                                   \
    if (shift && plain) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, HPW_, NHB_, WPE_, true, 0, false, 4 * HPW_, true>), DLMCQ_HALO_ARGS(4));  \
    else if (shift) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, HPW_, NHB_, WPE_, true>), DLMCQ_HALO_ARGS(4));                                \
    else if (plain) hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, HPW_, NHB_, WPE_, false, 0, false, 4 * HPW_, true>), DLMCQ_HALO_ARGS(4));     \
    else hipLaunchKernelGGL((conv3x3_halo_i8_kernel<BN_, TM, 4, 1, HPW_, NHB_, WPE_, false>), DLMCQ_HALO_ARGS(4));                                          \
  } while (0)
  if (bn == 128) {
    if (!wide) DLMCQ_HALO_GO(128, 6, 2, 2);
    else DLMCQ_HALO_GO(128, 8, 2, 1);
  } else if (C == 64) {
    if (!wide) DLMCQ_HALO_GO(64, 6, 1, 4);
    else DLMCQ_HALO_GO(64, 8, 1, 3);
  } else {
    if (!wide) DLMCQ_HALO_GO(64, 6, 2, 2);
    else DLMCQ_HALO_GO(64, 8, 2, 2);
  }
#undef DLMCQ_HALO_GO
#undef DLMCQ_HALO_ARGS
  return launch_status();
}

}  // namespace dlmcq
