// The halo-tile 3x3 kernel (conv3x3_i8.hip), PERSISTENT and SOFTWARE-PIPELINED ACROSS TILES (round 5): tile t's quantising epilogue is issued
// in the shadow of tile t + 1's K loop.  Same linear frame, same halo tile, same weight ring, same arithmetic (modules/conv.py:13-19 on the
// operands of FSPTQuant/base.py:108-109,149-152, the model's ReLU and the consumer's quantiser behind it) - the same bytes as
// conv3x3_halo_i8_kernel<..., PLAIN> and as the tiled kernel.
//
// Why: conv3x3_halo_i8_kernel runs two workgroups per CU whose phases coincide - both in their matrix-bound K loops, then both in their
// vector-bound epilogues (dequantise, quantise, stage, store: a quarter of a tile's life) while the matrix pipes idle; inside a wave the two
// kinds of work alternate, and a phase shift between the workgroups recovered 3 - 5 % (LABNOTES 11).  The matrix pipe and the vector
// pipe of a SIMD are separate: an MFMA holds the SIMD's issue for 8 of its 32 clocks, a handful of vector instructions fits behind each.
// So: ONE workgroup of EIGHT waves per CU - two per SIMD, each 64 pixels x 64 channels of the tile - each wave with TWO accumulator sets:
// `acc`, which the K loop of the current tile multiplies into, and `accE`, the finished sums of the tile before.  (The first build had four
// waves of 64 x 128, one per SIMD with 400 registers: bit-identical and 20 - 35 % SLOWER than the kernel it was to replace - with one wave
// per SIMD nothing covers an LDS round trip, a DMA issue or a barrier, and the compiler's schedule hid a third of the epilogue at best:
// compile-time ablations, tools/halo_pipe_lab.sh, LABNOTES 15.  Two waves per SIMD cover each other as the two workgroups of the plain
// kernel do, and there is no epilogue PHASE left for them to fall into together.)  The chunk loop stays a run-time loop of nine unrolled
// steps (a first version unrolled all 9 C / 64 steps to have every accumulator register index a compile-time constant: 110 KB of code for a
// 64 KB instruction cache - an EMPTY skeleton of it took longer than the kernel it was to replace); the epilogue's share of a chunk - 32 /
// (C / 64) quads, i.e. 8 / (C / 64) accumulator blocks - is copied out of accE into a fixed work buffer at the chunk's start (a uniform
// switch on the chunk index selects the source registers), so the quads index compile-time registers again.  Each quad's fast path (4
// values: dequantise on pairs, code4_plain_fast) is issued before a half-step's eight MFMAs, its rare exact redo and its store into the
// wave's code stage after them; its constants are read from LDS a half-step ahead.  A tile boundary is: the previous tile's staged rows
// out (eight 16-byte stores per lane), `accE = acc`, on to the next tile - whose first slabs and halo tile have been requested by the last
// steps of this one: the weight ring and the halo double buffer run through tile boundaries as through chunk boundaries.
//
// Takes: stride 1, C in {128, 256, 512}, K a multiple of 128 up to 512, the plain quantiser (epi_plain), rows up to 62 pixels wide, at
// least two tiles per CU.  Everything else: conv3x3_halo_i8_kernel.
#include "conv_i8_common.h"

namespace dlmcq {

struct PipeGeom {
  int N, H, W, C, K;
  int Wp, FS;            // W + 1, (H + 1) (W + 1)
  uint32_t MQ;           // N FS: rows of the linear frame space
  int nblk_n, hp;        // column blocks of 128 channels; halo pieces (16 frame positions each) a chunk's tile needs
  uint32_t ntiles;
  FastDiv fsdiv, wpdiv, nbdiv;
  int lab;               // lab builds, timing only (results are garbage): 1 = no weight DMA, 2 = no halo DMA, 4 = no epilogue quads, 8 = no barriers
};

// Timing-only ablations are COMPILE-TIME (-DDLMCQ_PIPE_ABL=bits builds an A/B library: tools/halo_pipe_lab.sh): run-time flags in this
// kernel's loop cost the lab build 500 - 700 spilled registers, and its times meant nothing.
#ifndef DLMCQ_PIPE_ABL
#define DLMCQ_PIPE_ABL 0
#endif
#define PIPE_LAB(bit) ((DLMCQ_PIPE_ABL) & (bit))

constexpr int PIPE_KMAX = 512;     // output channels the constant table in LDS holds

// -DDLMCQ_PIPE_STAMP (A/B library only, tools/halo_pipe_lab.sh): wave 0 of workgroup 0 records the shader clock at every K step of its first two
// tiles, and the shader clock + the 100 MHz constant clock around its whole life (their quotient = the clock the chip holds under this kernel)
#ifdef DLMCQ_PIPE_STAMP
__device__ unsigned long long g_pipe_stamps[256];
__device__ unsigned long long g_pipe_wg[1024];       // per workgroup: start, end in 100 MHz ticks (s_memrealtime: one clock for the whole chip)
#define PIPE_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 250) g_pipe_stamps[(i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PIPE_STAMP(i) do { } while (0)
#endif

template <int NCH, bool XS>
__global__ __launch_bounds__(512, 2) void conv3x3_pipe_i8_kernel(
    const int8_t* __restrict__ x, const int8_t* __restrict__ w, const float* __restrict__ bias, const int32_t* __restrict__ wsum,
    const float* __restrict__ s_in, const float* __restrict__ zp_in, const float* __restrict__ s_w, PipeGeom g, int shift, ConvEpi ep) {
  constexpr int BN = 128, TM = 256, NW = 8, WC = 2, PW = 64, CW = BN / WC, PB = 2, CB = CW / 32, HPW = 3, HP = HPW * NW;
  constexpr int NBUF = 3;                     // the weight ring: a slab per K step, requested three steps ahead (slot = tap % 3 in every chunk)
  constexpr int SLAB = BN * 64, NBW = SLAB / 1024 / NW, RING = NBUF * SLAB, HALO = HP * 1024;
  constexpr int QPC = 4 * CB * PB / NCH;      // quads (4 channels of a pixel) of the previous tile's epilogue a chunk carries, per wave
  constexpr int SROW = BN + 16, STAGE = TM * SROW, PARB = 3 * PIPE_KMAX * 4;
  constexpr int NST = TM / (NW * 8);          // row stores per lane at a tile boundary (8 rows of 128 bytes per wave-instruction)
  static_assert(NCH % 2 == 0 && NCH <= 8 && NBW == 1 && QPC >= 1 && QPC <= 8, "tile shape");
  __shared__ __attribute__((aligned(1024))) int8_t lds[RING + 2 * HALO + STAGE + PARB];
  int8_t* const ring = lds;
  int8_t* const halo = lds + RING;
  int8_t* const stg = halo + 2 * HALO;
  int8_t* const par = stg + STAGE;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const int wp = wave & 3, wc = wave >> 2;        // pixel group, channel half (waves w and w + 4 - one SIMD's two - share their pixels)
  // every compiler-known global load first (scalars of the layer and of the consumer's quantiser)
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin_early = s_in[0];
  const EpiQuant eq(ep, ep.relu != 0);            // code(relu(v)) = max(code(v), code(0)); the plain quantiser's pack saturates at 0
  const uint32_t xorw = 0x80808080u;
  const int8_t* const padline = g_pad_table.b + ((zpi & 0xff) << 6);

  // ---- per-channel constants of ALL K output channels, once per workgroup: (s_w, SUM qw, bias) by LDS-DMA, 64 channels per wave-instruction ----
  {
    const int kc = g.K >> 6;
    for (int c = wave; c < 3 * kc; c += NW) {
      const int a = c / kc, seg = c - a * kc;
      const int32_t* const src = a == 0 ? reinterpret_cast<const int32_t*>(s_w) : (a == 1 || !bias) ? (a == 1 ? wsum : reinterpret_cast<const int32_t*>(s_w))
                                                                                                  : reinterpret_cast<const int32_t*>(bias);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + seg * 64 + lane), (lptr_t)(par + (a * PIPE_KMAX + seg * 64) * 4), 4, 0, 0);
    }
  }

  // ---- tiles: virtual block vb = blockIdx.x + round * gridDim.x (gridDim.x a multiple of 8: a workgroup's tiles stay on its XCD); the XCD-aware
  // order of conv3x3_i8.hip - consecutive tiles, i.e. the column blocks of one row block, on one XCD ----
  const uint32_t G = gridDim.x;
  auto tile_at = [&](uint32_t vb, uint32_t& q0, int& n0) {
    const uint32_t xcd = vb & 7u, slot = vb >> 3;
    const uint32_t qd = g.ntiles >> 3, rm = g.ntiles & 7u;
    const uint32_t tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
    const uint32_t bm = fdiv(tile, g.nbdiv);
    q0 = bm * TM;
    n0 = (int)(tile - bm * (uint32_t)g.nblk_n) * BN;
  };

  // ---- halo DMA: piece i of this wave covers halo positions (i * NW + wave) * 16 .. + 15 (conv3x3_i8.hip) ----
  const int lrow = lane >> 2, pslot = lane & 3;
  const int8_t* hsrc[HPW];
  int hinc[HPW], hpc[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) {
    const int pc = i * NW + wave;
    hpc[i] = pc < g.hp ? pc : g.hp - 1;
  }
  auto halo_sources = [&](uint32_t q0) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      const int p = hpc[i] * 16 + lrow;
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
  };
  auto issue_halo = [&](auto i_c, int buf) {
    constexpr int i = decltype(i_c)::value;
    if (!PIPE_LAB(2)) __builtin_amdgcn_global_load_lds((gptr_t)hsrc[i], (lptr_t)(halo + buf * HALO + hpc[i] * 1024), 16, 0, 0);
    hsrc[i] += hinc[i];
  };

  // ---- weight DMA: piece j of this wave moves slab rows (j * NW + wave) * 16 .. + 15 of every step (conv3x3_i8.hip) ----
  const int8_t* wsrc[NBW];
  int wrow[NBW];         // the output channel (within a column block) a lane's slab row holds
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int drow = (j * NW + wave) * 16 + lrow, d = drow & 31;
    wrow[j] = (drow & ~31) + 16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
  }
  auto weight_sources = [&](int n0) {
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int drow = (j * NW + wave) * 16 + lrow;
      wsrc[j] = w + (int64_t)(n0 + wrow[j]) * (9 * g.C) + (pslot ^ ((drow >> 2) & 3)) * 16;
    }
  };
  auto issue_w = [&](auto slot_c, auto u_c) {
    constexpr int SL = decltype(slot_c)::value, u = decltype(u_c)::value;
    const int inc = u == 8 ? 64 - 8 * g.C : g.C;         // to the next tap's slab, or to tap 0 of the next chunk
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      if (!PIPE_LAB(1)) __builtin_amdgcn_global_load_lds((gptr_t)wsrc[j], (lptr_t)(ring + SL * SLAB + (j * NW + wave) * 1024), 16, 0, 0);
      wsrc[j] += inc;
    }
  };

  // ---- fragment addresses: a wave's 64 pixels x 64 channels ----
  int woff[2];
  {
    const int d = wc * CW + l31;
    woff[0] = d * 64 + ((hsel ^ ((d >> 2) & 3)) << 4);
    woff[1] = woff[0] ^ 32;
  }
  const int pbase = wp * PW + l31;
  const int tap_r1 = g.Wp, tap_r2 = 2 * g.Wp;

  i32x16 acc[CB][PB], accE[CB][PB];
#pragma unroll
  for (int jc = 0; jc < CB; ++jc)
#pragma unroll
    for (int jp = 0; jp < PB; ++jp)
#pragma unroll
      for (int i = 0; i < 16; ++i) accE[jc][jp][i] = 0;

  i32x4 wfA[CB], pfA[PB], wfB[CB], pfB[PB];
  auto read_frags = [&](auto u_c, auto t_c, auto ks_c, const int8_t* hbuf, i32x4 (&wf)[CB], i32x4 (&pf)[PB]) {
    constexpr int U = decltype(u_c)::value, t = decltype(t_c)::value, ks = decltype(ks_c)::value;
    const int p0 = pbase + (t / 3 == 0 ? 0 : (t / 3 == 1 ? tap_r1 : tap_r2)) + t % 3;
    const int pa = (p0 * 64 + ((hsel ^ ((p0 >> 2) & 3)) << 4)) ^ (ks << 5);
    const int8_t* const sb = ring + U * SLAB;
    if (PIPE_LAB(16)) {                  // timing only: no fragment reads (the MFMAs run on whatever the registers hold)
#pragma unroll
      for (int jc = 0; jc < CB; ++jc) asm volatile("" : "+v"(wf[jc]));
#pragma unroll
      for (int jp = 0; jp < PB; ++jp) asm volatile("" : "+v"(pf[jp]));
      return;
    }
#pragma unroll
    for (int jc = 0; jc < CB; ++jc) wf[jc] = *reinterpret_cast<const i32x4*>(sb + woff[ks] + jc * 2048);
#pragma unroll
    for (int jp = 0; jp < PB; ++jp) pf[jp] = *reinterpret_cast<const i32x4*>(hbuf + pa + jp * 2048);
  };
  auto multiply = [&](auto first_c, const i32x4 (&wf)[CB], i32x4 (&pf)[PB]) {
    constexpr bool FIRST = decltype(first_c)::value;       // a tile's first half-step starts its sums from zero (no clearing pass)
    if (XS) {
#pragma unroll
      for (int jp = 0; jp < PB; ++jp)
        pf[jp] = i32x4{(int)(pf[jp].x ^ xorw), (int)(pf[jp].y ^ xorw), (int)(pf[jp].z ^ xorw), (int)(pf[jp].w ^ xorw)};
    }
#pragma unroll
    for (int jc = 0; jc < CB; ++jc)
#pragma unroll
      for (int jp = 0; jp < PB; ++jp) {
        if (PIPE_LAB(32)) {              // timing only: no MFMAs
          asm volatile("" ::"v"(wf[jc]), "v"(pf[jp]));
          continue;
        }
        if constexpr (FIRST) {
          const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          acc[jc][jp] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[jc], pf[jp], zero, 0, 0, 0);
        } else {
          acc[jc][jp] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[jc], pf[jp], acc[jc][jp], 0, 0, 0);
        }
      }
  };

  // The order of a half-step's instructions, told to the scheduler (sched_group_barrier: 0x8 MFMA, 0x100 LDS read, 0x20 vector-memory read,
  // 0x2 vector ALU): behind each of the eight MFMAs two of the LDS reads (the other fragment set, the next quad's constants), one of the
  // phase's DMA requests and four vector instructions of the quad - what fits into the 24 issue clocks an MFMA leaves of its 32.  With one
  // wave per SIMD whatever is NOT between two MFMAs is time the matrix pipe idles.
  auto interleave = [&]() {
#ifndef DLMCQ_PIPE_NO_SGB
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
#endif
  };

  // ---- the epilogue of the tile whose sums are in accE, a quad (4 channels of one pixel) at a time.  Quad j: block (jc, jp) = (j / 4 % CB,
  // j / 4 / CB), channels cb + 4 q .. + 3 with cb = jc * 32 + 16 hsel, of pixel row jp * 32 + l31 of this wave.  conv3x3_i8.hip's PLAIN
  // epilogue, the same operations in the same order. ----
  const int8_t* parE = par;                       // the constants of accE's column block: par + n0E * 4
  int8_t* const stgw = stg + (wp * PW + l31) * SROW + wc * CW + hsel * 16;
  // A quad's twelve constants are read from LDS ONE HALF-STEP AHEAD of the quad (with that half-step's fragment reads, in front of its
  // MFMAs): with one wave per SIMD nothing covers an LDS round trip in the middle of a half-step - the first build of this kernel waited
  // for its constants where it used them and ran 20 % SLOWER than the kernel it replaces
  f32x4 cmu[2], cbs[2];
  i32x4 cco[2];
  f32x4 qy;                                       // the quad in flight across a half-step's MFMAs
  uint32_t qw = 0;
  uint64_t qb = 0;
  i32x4 wk[QPC];                                  // the accE registers of the chunk's quads: quad (c * QPC + j) = block (jq / 4), registers 4 (jq % 4) ..
  int jq0 = 0;                                    // c * QPC
  auto load_work = [&](int c) {                   // a uniform switch selects the source registers
    jq0 = c * QPC;
    static_for<NCH>([&](auto cc) {
      constexpr int C0 = decltype(cc)::value;
      if (c == C0) {
        static_for<QPC>([&](auto jj) {
          constexpr int j = decltype(jj)::value, jq = C0 * QPC + j, b = jq / 4, q = jq % 4;
          wk[j] = i32x4{accE[b % CB][b / CB][4 * q], accE[b % CB][b / CB][4 * q + 1], accE[b % CB][b / CB][4 * q + 2], accE[b % CB][b / CB][4 * q + 3]};
        });
      }
    });
  };
  auto quad_prefetch = [&](auto j_c) {
    constexpr int j = decltype(j_c)::value;       // quad j of the chunk = quad jq of the tile: block jq / 4 = (jc, jp) = (b % CB, b / CB), channels 4 (jq % 4) ..
    const int jq = jq0 + j, jc = (jq >> 2) & (CB - 1);
    const int8_t* const pp = parE + (wc * CW + jc * 32 + 4 * (jq & 3)) * 4 + hsel * 64;
    cmu[j & 1] = *reinterpret_cast<const f32x4*>(pp);
    cco[j & 1] = *reinterpret_cast<const i32x4*>(pp + PIPE_KMAX * 4);
    cbs[j & 1] = *reinterpret_cast<const f32x4*>(pp + 2 * PIPE_KMAX * 4);
  };
  auto quad_fast = [&](auto j_c) {
    constexpr int j = decltype(j_c)::value;
    const f32x4 mu = cmu[j & 1], bs = cbs[j & 1];
    const i32x4 co = cco[j & 1];
    const f32x2 ya = pk_fma(f32x2{(float)(wk[j].x + co.x), (float)(wk[j].y + co.y)}, f32x2{mu.x, mu.y}, f32x2{bs.x, bs.y});
    const f32x2 yb = pk_fma(f32x2{(float)(wk[j].z + co.z), (float)(wk[j].w + co.w)}, f32x2{mu.z, mu.w}, f32x2{bs.z, bs.w});
    qy = f32x4{ya.x, ya.y, yb.x, yb.y};
    qw = eq.code4_plain_fast(qy, qb);
  };
  auto quad_done = [&](auto j_c) {
    constexpr int j = decltype(j_c)::value;
    const int jq = jq0 + j, b = jq >> 2, jc = b & (CB - 1), jp = b / CB;
    if (__builtin_expect(qb != 0, false)) qw = eq.exact4(qy, qw);
    *reinterpret_cast<uint32_t*>(stgw + jp * 32 * SROW + jc * 32 + 4 * (jq & 3)) = qw;
  };
  // half-step h (0 .. 17) of a chunk carries the quad j with HS(j) = (j * 18) / QPC == h (at most one per half-step: 18 > QPC); its
  // constants are fetched a half-step earlier (quad 0's, due in half-step 0, at the end of the chunk before - or at the tile boundary)
  auto quads_prefetch = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if (PIPE_LAB(4)) return;
    static_for<QPC>([&](auto j_c) {
      if constexpr ((decltype(j_c)::value * 18) / QPC == h + 1) quad_prefetch(j_c);
    });
  };
  auto quads_before = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if (PIPE_LAB(4)) return;
    static_for<QPC>([&](auto j_c) {
      if constexpr ((decltype(j_c)::value * 18) / QPC == h) quad_fast(j_c);
    });
  };
  auto quads_after = [&](auto h_c) {
    constexpr int h = decltype(h_c)::value;
    if (PIPE_LAB(4)) return;
    static_for<QPC>([&](auto j_c) {
      if constexpr ((decltype(j_c)::value * 18) / QPC == h) quad_done(j_c);
    });
  };
  // the staged rows of the tile (q0E, n0E) out: 8 lanes = one pixel's 128 code bytes; rows that are no pixel of the batch (frame borders,
  // the tile's overhang, the "tile" before a workgroup's first) are stored nowhere - every lane issues NST stores whatever it owns
  const v4i r_codes = make_rsrc(ep.codes, (uint32_t)((int64_t)g.N * g.H * g.W * g.K));
  auto store_rows = [&](uint32_t q0E, int n0E, bool valid) {
    const int srow = tid >> 3, sseg = lane & 7;
#pragma unroll
    for (int it = 0; it < NST; ++it) {
      const int rr = it * (NW * 8) + srow;
      const uint32_t q = q0E + (uint32_t)rr;
      const uint32_t n = fdiv(q, g.fsdiv);
      const uint32_t rem = q - n * (uint32_t)g.FS;
      const uint32_t yy = fdiv(rem, g.wpdiv);
      const uint32_t xx = rem - yy * (uint32_t)g.Wp;
      const bool ok = valid && q < g.MQ && yy < (uint32_t)g.H && xx < (uint32_t)g.W;
      const int off = (int)(((n * (uint32_t)g.H + yy) * (uint32_t)g.W + xx) * (uint32_t)g.K) + n0E + sseg * 16;
      const i32x4 c16 = *reinterpret_cast<const i32x4*>(stg + rr * SROW + sseg * 16);
      bstore16i_nt(c16, ok ? off : BUF_BIG, r_codes);
    }
  };

  // ---- prologue: the first tile's chunk-0 halo tile and its first three slabs; the constant table finished in place ----
  uint32_t vb = blockIdx.x, q0 = 0, q0E = 0;
  int n0 = 0, n0E = 0;
  bool haveE = false;
  tile_at(vb, q0, n0);
  halo_sources(q0);
  weight_sources(n0);
  static_for<HPW>([&](auto i) { issue_halo(i, 0); });
  issue_w(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  issue_w(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  issue_w(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NBW) : "memory");
  for (int i = tid; i < g.K; i += NW * 64) {
    float* const pf = reinterpret_cast<float*>(par) + i;
    int* const pi = reinterpret_cast<int*>(par) + PIPE_KMAX + i;
    *pf = sin_early * *pf;
    *pi = (shift - zpi) * *pi;
    if (!bias) reinterpret_cast<float*>(par)[2 * PIPE_KMAX + i] = 0.0f;
  }
  read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, halo, wfA, pfA);
  store_rows(0u, 0, false);        // (NST stores that go nowhere: the queue every tile's first waits count on)

#ifdef DLMCQ_PIPE_STAMP
  int stamp_i = 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_pipe_stamps[0] = __builtin_readcyclecounter();
    g_pipe_stamps[1] = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0 && blockIdx.x < 512) g_pipe_wg[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
  for (;;) {
    const bool has_next = vb + G < g.ntiles;
    uint32_t q0N = 0;
    int n0N = 0;
    if (has_next) tile_at(vb + G, q0N, n0N);
    load_work(0);
    quad_prefetch(std::integral_constant<int, 0>{});          // (the tile's first quad is due in its first half-step)
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      const bool lastc = c == NCH - 1;
      const bool more = !lastc || has_next;                   // a chunk follows (of this tile, or chunk 0 of the next)
      const int8_t* const hcur = halo + (c & 1) * HALO;
      const int8_t* const hnxt = halo + ((c + 1) & 1) * HALO;
      if (lastc && has_next) halo_sources(q0N);               // (this tile's last halo piece was requested a chunk ago)
      static_for<9>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        constexpr int U = t % NBUF;
#ifdef DLMCQ_PIPE_STAMP
        PIPE_STAMP(stamp_i);
        ++stamp_i;
#endif
        // phase 0: this half-step's LDS reads (the other fragment set, the next quad's constants) in FRONT of its MFMAs
        read_frags(std::integral_constant<int, U>{}, t_c, std::integral_constant<int, 1>{}, hcur, wfB, pfB);
        quads_prefetch(std::integral_constant<int, 2 * t>{});
        quads_before(std::integral_constant<int, 2 * t>{});
        if (t == 0 && c == 0) multiply(std::true_type{}, wfA, pfA);
        else multiply(std::false_type{}, wfA, pfA);
        interleave();
        quads_after(std::integral_constant<int, 2 * t>{});
        // phase 1 (conv3x3_i8.hip's waits; a tile's first two also leave the boundary's NST row stores in flight)
        if (t < 8 || more) {
          constexpr bool HPREV = t >= 1 && t <= HPW && t <= 7;
          const bool first2 = t < 2 && c == 0;
          if (PIPE_LAB(63)) {      // (the ablations change what is in flight: plain full waits, with or without the barrier)
            if (PIPE_LAB(8)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          } else if (t >= 7 && !more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else if (first2) {
            if (HPREV) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NBW + 1 + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NBW + NST) : "memory");
          } else if (HPREV && more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NBW + 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NBW) : "memory");
          read_frags(std::integral_constant<int, (U + 1) % NBUF>{}, std::integral_constant<int, (t + 1) % 9>{}, std::integral_constant<int, 0>{},
                     t == 8 ? hnxt : hcur, wfA, pfA);
          if constexpr (t < HPW) {
            if (more) issue_halo(t_c, (c + 1) & 1);
          }
          if constexpr (t == 6) {
            if (lastc && has_next) weight_sources(n0N);       // slab s + 3 is the next tile's first
          }
          if (t < 6 || more) issue_w(std::integral_constant<int, U>{}, std::integral_constant<int, (t + 3) % 9>{});
        }
        if constexpr (t == 8) {
          // the next chunk's share of the epilogue: its work blocks out of accE, its first quad's constants (at a tile's end: the boundary does both)
          if (!lastc) {
            load_work(c + 1);
            quad_prefetch(std::integral_constant<int, 0>{});
          }
        } else {
          quads_prefetch(std::integral_constant<int, 2 * t + 1>{});
        }
        quads_before(std::integral_constant<int, 2 * t + 1>{});
        multiply(std::false_type{}, wfB, pfB);
        interleave();
        quads_after(std::integral_constant<int, 2 * t + 1>{});
      });
    }
    // ---- tile boundary: the previous tile's rows out, this tile's sums become the epilogue's ----
    // (a staged row holds the codes of two waves, and the next tile's quads overwrite the stage: a barrier on either side of the row stores)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    store_rows(q0E, n0E, haveE);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int jc = 0; jc < CB; ++jc)
#pragma unroll
      for (int jp = 0; jp < PB; ++jp) accE[jc][jp] = acc[jc][jp];
    q0E = q0; n0E = n0; haveE = true;
    parE = par + n0 * 4;
    if (!has_next) break;
    vb += G; q0 = q0N; n0 = n0N;
  }
  // ---- drain: the last tile's epilogue on its own ----
#pragma unroll 1
  for (int c = 0; c < NCH; ++c) {
    load_work(c);
    static_for<QPC>([&](auto j_c) {
      quad_prefetch(j_c);
      quad_fast(j_c);
      quad_done(j_c);
    });
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  store_rows(q0E, n0E, true);
#ifdef DLMCQ_PIPE_STAMP
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_pipe_stamps[2] = __builtin_readcyclecounter();
    g_pipe_stamps[3] = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0 && blockIdx.x < 512) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g_pipe_wg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

template <int NCH>
static int conv3x3_pipe_go(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                           const float* in_zero_point, const float* w_scale, const PipeGeom& g, int shift, const ConvEpi& ep, hipStream_t st,
                           uint32_t grid) {
  if (shift) hipLaunchKernelGGL((conv3x3_pipe_i8_kernel<NCH, true>), dim3(grid), dim3(512), 0, st, x, w, bias, wsum, in_scale, in_zero_point,
                                w_scale, g, shift, ep);
  else hipLaunchKernelGGL((conv3x3_pipe_i8_kernel<NCH, false>), dim3(grid), dim3(512), 0, st, x, w, bias, wsum, in_scale, in_zero_point,
                          w_scale, g, shift, ep);
  return launch_status();
}

#ifdef DLMCQ_PIPE_STAMP
}  // namespace dlmcq
extern "C" int dlmcq_x_pipe_stamps(unsigned long long* host256) {
  return (int)hipMemcpyFromSymbol(host256, HIP_SYMBOL(dlmcq::g_pipe_stamps), 256 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
extern "C" int dlmcq_x_pipe_wg(unsigned long long* host1024) {
  return (int)hipMemcpyFromSymbol(host1024, HIP_SYMBOL(dlmcq::g_pipe_wg), 1024 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
namespace dlmcq {
#endif
bool conv3x3_pipe_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int32_t stride, const ConvEpi& ep, int cus) {
  if (stride != 1 || !(C == 128 || C == 256 || C == 512) || K % 128 != 0 || K > PIPE_KMAX || !epi_plain(ep)) return false;
  if (W + 1 > 62) return false;                                                   // six halo pieces per wave
  const int64_t MQ = N * (H + 1) * (W + 1);
  const int64_t ntiles = ((MQ + 255) / 256) * (K / 128);
  if (ntiles < 2 * (int64_t)cus) return false;                                    // a pipeline needs tiles to run through: two per CU at least
  return N * H * W * K < (int64_t)BUF_BIG;                                        // 32-bit buffer offsets of the row stores
}

int conv3x3_pipe_launch(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                        const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int shift,
                        const ConvEpi& ep, hipStream_t st, int cus) {
  PipeGeom g;
  g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K;
  g.Wp = g.W + 1;
  g.FS = (g.H + 1) * (g.W + 1);
  g.MQ = (uint32_t)(N * g.FS);
  g.nblk_n = (int)(K / 128);
  g.hp = (256 + 2 * g.Wp + 2 + 15) / 16;
  g.ntiles = (uint32_t)(((int64_t)g.MQ + 255) / 256 * g.nblk_n);
  g.fsdiv = make_fastdiv((uint32_t)g.FS);
  g.wpdiv = make_fastdiv((uint32_t)g.Wp);
  g.nbdiv = make_fastdiv((uint32_t)g.nblk_n);
  g.lab = 0;
  uint32_t grid = (uint32_t)cus & ~7u;             // one workgroup per CU, a multiple of 8 (a workgroup's tiles stay on its XCD)
  if (grid > (g.ntiles & ~7u)) grid = g.ntiles & ~7u;      // (every workgroup owns at least one tile)
  if (grid < 8 || g.hp > 24) return DLMCQ_EINVAL;
  if (C == 128) return conv3x3_pipe_go<2>(x, w, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, st, grid);
  if (C == 256) return conv3x3_pipe_go<4>(x, w, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, st, grid);
  if (C == 512) return conv3x3_pipe_go<8>(x, w, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, st, grid);
  return DLMCQ_EINVAL;
}

}  // namespace dlmcq
