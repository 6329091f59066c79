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
// pixels x BN in {64, 128, 256} channels per workgroup (256: codes-only layers), 4 waves (one 32-row slab each), BK = 64,
// a 3-slot LDS ring fed by LDS-DMA (one barrier per K step, XOR-swizzled 64-byte rows: see the kernel).
#include <cstdlib>
#include <type_traits>

#include "conv_i8_common.h"

namespace dlmcq {


// ------------------------------------------------------------------------------------------------------
// The kernel: BM = 128 output pixels x BN channels per workgroup, 4 waves (one 32-row slab each), BK = 64.
//   * weights (B) go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging) into a 3-slot ring: two K
//     steps in flight, ONE raw s_barrier per step, counted s_waitcnt vmcnt.  A wave-instruction lands 16 rows x 64 B
//     contiguously, so rows cannot be padded; bank conflicts are avoided by an XOR swizzle applied on the SOURCE side
//     (LDS slot p of row r holds the logical 16-byte segment p ^ ((r >> 2) & 3)) and undone by the fragment reads;
//   * ADIR: activations (A) bypass LDS - a wave multiplies only its own 32 rows, so each lane loads its own fragment
//     bytes straight into registers, two steps ahead.  These loads are inline asm: beside an LDS-DMA in flight hipcc
//     drains the whole queue (vmcnt(0)) at the first use of any load it knows about.  (!ADIR stages A through the ring
//     as well: kept for 1x1 reductions of >= 256 channels into 64, where full-line DMA reads beat fragment loads);
//   * DMA cannot transform or synthesise bytes: the uint8 -> int8 shift (q ^ 0x80) is applied to the A fragment in
//     registers, and padded taps read a 64-byte line of a constant table instead;
//   * DUAL: a second (input, weight) pair - the shortcut convolution of a residual block's first unit - reduced FIRST
//     through the same ring into the same tile, so that conv3(x) + downsample(y) is one kernel;
//   * everything the epilogue needs from memory (per-channel scale / code sum / bias of both pairs and, for 64-wide
//     tiles, the fp32 shortcut tile) is requested BEFORE the first operand, so that a tile pays one memory round trip,
//     not four in a row: at ResNet sizes most tiles have 1-8 K steps and their lifetime is latency, not work.
//   * SWAP (layers that emit only their consumer's codes - no fp32 output, no shortcut: every 3x3 and every first 1x1 of a
//     residual block): the MFMA operands change places (weights as A, pixels as B), so a lane's 16 accumulator registers of a
//     32-channel block are 16 CHANNELS of ONE pixel instead of 16 pixels of one channel.  The weight rows are dealt to the LDS
//     rows in the order that makes those 16 channels consecutive (a source-side permutation of the DMA, free), so the lane
//     quantises them and holds 16 finished bytes: no transposition of fp32 values through LDS or DPP, the ReLU folded into the
//     quantiser's clamp, per-channel constants by broadcast ds_read_b128.  ~10 vector instructions per output element instead
//     of ~20; the code tile (1 B per element) goes through LDS once so that the stores are whole rows.
template <int BN, bool DUAL, bool ADIR, bool ASYM = false, int LAB = 0, bool SWAP = false>
__global__ __launch_bounds__(256, (BN == 256 ? 2 : DUAL ? (BN == 128 ? 2 : 4) : (SWAP && BN == 64 && ADIR && !ASYM ? 5 : SWAP && BN == 128 && ADIR && !ASYM ? 3 : ADIR || BN == 64 ? (ASYM ? 3 : 4) : 3))) void conv_i8_mfma_kernel(
    const int8_t* __restrict__ x, const int8_t* __restrict__ w, float* __restrict__ out, const float* __restrict__ bias,
    const int32_t* __restrict__ wsum, const float* __restrict__ s_in, const float* __restrict__ zp_in,
    const float* __restrict__ s_w, ConvGeom g, int shift, ConvEpi ep, ConvSeg2 sg) {
#ifndef DLMCQ_NB_SWAP128
#define DLMCQ_NB_SWAP128 3
#endif
  // ring depth: 3 slots (two K steps in flight); A/B hook for the 128-wide A-direct swapped kernel, whose LDS and registers leave room for more
#ifndef DLMCQ_NB_DUAL
#define DLMCQ_NB_DUAL 3
#endif
#ifndef DLMCQ_NB_PLAIN
#define DLMCQ_NB_PLAIN 3
#endif
  constexpr int BM = CV_BM, BK = CV_BK,
                NBUF = (SWAP && BN == 128 && ADIR && !ASYM) ? DLMCQ_NB_SWAP128 : (DUAL ? DLMCQ_NB_DUAL : ((!SWAP && !ASYM) ? DLMCQ_NB_PLAIN : 3)), PF = NBUF - 1;
  // LAB (lab library only): 1 = clock stamps; 2 = no A loads, 3 = no B loads, 4 = no MFMAs, 5 = all lanes load one A address,
  // 6 = swapped epilogue without its arithmetic (the accumulators' low bytes are stored), 7 = no epilogue at all
  // (what-bounds-the-step experiments: results are garbage, only the time means something)
  constexpr bool STAMP = LAB == 1;
  constexpr int TILE_A = ADIR ? 0 : BM * BK, TILE_B = BN * BK, TILE = TILE_A + TILE_B;
  constexpr int KS = BK / 32;           // MFMA K chunks per step
  constexpr int NT = BN / 32;           // 32-column slabs per wave
  constexpr int AI = BM / 64;           // A DMA wave-instructions per wave per step (16 rows each)
  constexpr int BI = BN / 64;           // B DMA wave-instructions per wave per step
  constexpr int NA = ADIR ? 1 : AI;     // A rows this lane addresses: its own fragment row (ADIR) or its DMA rows
  constexpr int EP_LD = 68;             // floats per staged epilogue row (64 + 4 pad)
  constexpr int EP_BYTES = SWAP ? 4 * 32 * (BN + 16) : 4 * 32 * EP_LD * 4;   // (swapped kernels stage 1-byte codes, not fp32 values)
  constexpr int LDS_BYTES = NBUF * TILE < EP_BYTES ? EP_BYTES : NBUF * TILE;   // the epilogue stage re-uses the ring
  constexpr int NH = NT / 2 + (NT & 1); // epilogue passes of 64 channels
  constexpr bool EARLY_RES = NH == 1;   // the whole shortcut tile is one pass: request it before the operands
  constexpr int GROUP = (ADIR ? KS : AI) + BI;   // vector-memory instructions per K step per wave
  constexpr int NPAR = DUAL ? 6 : (ASYM ? 4 : 3);   // per-channel constant arrays of the epilogue: (scale, code sum, bias) per pair (+ weight offset)
  constexpr int PAR_BYTES = NPAR * BN * 4;
  static_assert(!(ASYM && DUAL), "asymmetric weights: single pair only");
  static_assert(!(SWAP && DUAL), "SWAP: codes-only layers with one operand pair");
  __shared__ __attribute__((aligned(1024))) int8_t lds[LDS_BYTES + PAR_BYTES + (ASYM ? 4 * 32 * 4 : 0)];

  // XCD-aware tile order: the workgroups that share an activation tile (same row block, different column blocks) are
  // consecutive in `tile`, and consecutive tiles are dealt to the SAME XCD (its L2 then serves the re-reads)
  const unsigned long long t_start = STAMP ? __builtin_readcyclecounter() : 0ull;
  const uint32_t nwg = gridDim.x;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t qd = nwg >> 3, rm = nwg & 7u;
  const uint32_t tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
  const int bn = tile % g.nblk_n, bm = tile / g.nblk_n;
  const int64_t m0 = (int64_t)bm * BM;
  const int n0 = bn * BN;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const int wrow0 = wave * 32;          // first tile row of this wave

  // ---- epilogue operands, requested first.  The per-channel constants go by LDS-DMA into a table behind the ring (no
  // register is held across the K loop); the first K step's wait + barrier make them visible to every wave. ----
  auto request_par = [&](auto a_c, const void* arr) {
    constexpr int a = decltype(a_c)::value;
    if (!arr) return;
#pragma unroll
    for (int c = 0; c < BN / 64; ++c) {
      if (((a * (BN / 64) + c) & 3) != wave) continue;          // shared out among the waves
      int col = n0 + c * 64 + lane;
      col = col < g.K ? col : g.K - 1;                          // (columns beyond K are never stored)
      __builtin_amdgcn_global_load_lds((gptr_t)(static_cast<const int32_t*>(arr) + col),
                                       (lptr_t)(lds + LDS_BYTES + (a * BN + c * 64) * 4), 4, 0, 0);
    }
  };
  request_par(std::integral_constant<int, 0>{}, s_w);
  request_par(std::integral_constant<int, 1>{}, wsum);
  request_par(std::integral_constant<int, 2>{}, bias);
  if (ASYM) request_par(std::integral_constant<int, 3>{}, ep.w_off);
  if (DUAL) {
    request_par(std::integral_constant<int, 3>{}, sg.s_w);
    request_par(std::integral_constant<int, 4>{}, sg.wsum);
    request_par(std::integral_constant<int, 5>{}, sg.bias);
  }
  auto par_f = [&](int a, int j) { return *reinterpret_cast<const float*>(lds + LDS_BYTES + (a * BN + j * 32 + l31) * 4); };
  auto par_i = [&](int a, int j) { return *reinterpret_cast<const int*>(lds + LDS_BYTES + (a * BN + j * 32 + l31) * 4); };
  const int er = lane >> 4, ec = (lane & 15) * 4;   // row-major layout of the staged tile: 4 rows x 64 channels per wave-instruction
  f32x4 idt[8];
  auto load_residual = [&](int h) {
    const int colr = n0 + h * 64 + ec;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t row = m0 + wrow0 + it * 4 + er;
      idt[it] = (row < g.M && colr < g.K) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ep.residual + row * g.K + colr))
                                          : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  if (EARLY_RES && !DUAL && ep.residual && (g.K & 3) == 0) load_residual(0);

  // ---- DMA assignment: wave-instruction i of this wave covers tile rows (i*4 + wave)*16 .. +15 ----
  const int lrow = lane >> 2, pslot = lane & 3;
  int a_seg[AI], b_seg[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) a_seg[i] = pslot ^ ((((i * 4 + wave) * 16 + lrow) >> 2) & 3);
#pragma unroll
  for (int i = 0; i < BI; ++i) b_seg[i] = pslot ^ ((((i * 4 + wave) * 16 + lrow) >> 2) & 3);

  // One (input, weight) pair as a source of K steps.  DUAL kernels have two; their steps form ONE sequence through the
  // same LDS ring (the second pair's first steps are already in flight while the first pair's last steps multiply).
  struct Feed {
    const int8_t* x;
    const int8_t* padline;     // stored UNshifted: the xor happens on read
    int a_n[NA], a_h0[NA], a_w0[NA];
    bool a_ok[NA];
    int cc, s, r, cchunks, nsteps;
    uint32_t xorw;
    // running pointers: a K step costs one 64-bit add per operand row instead of the whole (bounds check, pixel
    // address, tap offset) computation - that arithmetic, not memory, was what bounded the loop
    const int8_t* ap[NA];   // this lane's A bytes for the current tap and channel chunk (or the pad line)
    int a_inc[NA];          // BK for a real pixel, 0 for a padded tap
    const int8_t* bp[BI];   // this lane's B source for the current step (KRSC: the reduction index is contiguous)
    int b_inc[BI];
  };
  auto retap = [&](Feed& f, const ConvGeom& gg) {   // A pointers of tap (f.r, f.s), channel chunk 0
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
      const int64_t m = m0 + (ADIR ? wrow0 + l31 : (i * 4 + wave) * 16 + lrow);
      f.a_ok[i] = m < gg.M;
      row_origin(gg, f.a_ok[i] ? (uint32_t)m : 0u, f.a_n[i], f.a_h0[i], f.a_w0[i]);
    }
    const int64_t wrow = (int64_t)gg.R * gg.S * gg.C;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int k = n0 + (i * 4 + wave) * 16 + lrow;
      if constexpr (SWAP) {   // LDS row d of a 32-row block holds channel 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3): see the epilogue
        const int d = ((i * 4 + wave) * 16 + lrow) & 31;
        k = (k & ~31) + 16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
      }
      f.bp[i] = k < gg.K ? ww + (int64_t)k * wrow + b_seg[i] * 16 : g_pad_table.b;
      f.b_inc[i] = k < gg.K ? BK : 0;
    }
    f.cc = f.s = f.r = 0;
    f.cchunks = gg.C / BK;
    f.nsteps = gg.R * gg.S * f.cchunks;
    retap(f, gg);
  };
  i32x4 areg[ADIR ? NBUF : 1][KS];   // ADIR: the A fragments of the steps in flight
  if (LAB == 2)
    for (auto& a3 : areg)
      for (auto& a : a3) a = i32x4{lane, 1, 2, 3};
  auto issue = [&](Feed& f, const ConvGeom& gg, auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
    int8_t* base = lds + SL * TILE;
    // B first: its DMA lands in LDS and is awaited by the whole workgroup
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      if (LAB != 3) __builtin_amdgcn_global_load_lds((gptr_t)f.bp[i], (lptr_t)(base + TILE_A + (i * 4 + wave) * 1024), 16, 0, 0);
      f.bp[i] += f.b_inc[i];
    }
    if (ADIR) {
      if (LAB == 5) {
        const uint64_t pa = (uint64_t)f.ap[0];
        const int8_t* pu = (const int8_t*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)pa));
        gload16<0>(areg[ADIR ? SL : 0][0], pu);
        gload16<32>(areg[ADIR ? SL : 0][1], pu);
      } else if (LAB != 2) {
        gload16<0>(areg[ADIR ? SL : 0][0], f.ap[0]);
        gload16<32>(areg[ADIR ? SL : 0][1], f.ap[0]);
      }
      f.ap[0] += f.a_inc[0];
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)f.ap[i], (lptr_t)(base + (i * 4 + wave) * 1024), 16, 0, 0);
        f.ap[i] += f.a_inc[i];
      }
    }
    if (++f.cc == f.cchunks) {
      f.cc = 0;
      if (++f.s == gg.S) {
        f.s = 0;
        ++f.r;
      }
      retap(f, gg);     // (past the last tap nothing is issued any more; the pointers are simply not used)
    }
  };

  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin_early = SWAP ? s_in[0] : 0.0f;
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

  i32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0;
  float extra[DUAL ? NT : 1][16];
  int s0 = 0;        // ASYM: running sum of this lane's share of its row's operand bytes (all taps, all channels)

  static_for<PF>([&](auto i) {
    if (decltype(i)::value < nsteps) issue_next(i);
  });
  // STAMP (lab builds only): one wave of a workgroup in the middle of the grid writes the shader clock at the phase
  // boundaries of its first 24 K steps into the buffer passed as ep.residual (which is then not used as a shortcut)
  unsigned long long* trace = nullptr;
  if (STAMP && blockIdx.x == gridDim.x / 2 && tid == 0) trace = (unsigned long long*)ep.residual;
  // ... and wave 0 of EVERY workgroup its start / K loop entered / K loop left / end clocks (6 values per workgroup behind those, with HW_ID and XCC_ID)
  unsigned long long* wgt = nullptr;
  if (STAMP && tid == 0) {
    wgt = (unsigned long long*)ep.residual + 24 * 8 + 6 * (size_t)blockIdx.x;
    wgt[0] = t_start;
    wgt[1] = __builtin_readcyclecounter();
    wgt[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));     // HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]
    wgt[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));    // XCC_ID
  }
  if (STAMP) ep.residual = nullptr;
  auto stamp = [&](int step, int k) {
    if (STAMP && trace && step < 24) trace[step * 8 + k] = __builtin_readcyclecounter();
  };
  auto one_step = [&](int step, auto slot_c) {
    constexpr int U = decltype(slot_c)::value;          // ring slot of this step; step + PF goes to slot (U + PF) % NBUF
    stamp(step, 0);
    // step's own loads must have landed; the younger group stays in flight
    static_assert((PF - 1) * GROUP <= 63, "vmcnt is a 6-bit count");
    // ... then the barrier: everyone's step-k bytes are in LDS; everyone left multiply(k-1).  Wait and barrier are ONE asm statement
    // with a memory clobber (a bare __builtin_amdgcn_s_barrier() is no memory operation for the compiler: nothing at IR level would
    // keep an LDS read from moving across it); the stamping lab build keeps them apart to time the wait alone
    if constexpr (STAMP) {
      if (step + PF - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * GROUP) : "memory");
      else if (PF > 2 && step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GROUP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(step, 1);
      asm volatile("s_barrier" ::: "memory");
    } else if (step + PF - 1 < nsteps) {           // PF - 1 younger steps stay in flight
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((PF - 1) * GROUP) : "memory");
    } else if (PF > 2 && step + 1 < nsteps) {      // (deeper rings, near the end: at least one younger step)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(GROUP) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // last step
    }
    stamp(step, 2);
    if constexpr (SWAP) {
      // the per-channel constants have landed: turn (s_w, SUM qw) into the epilogue's (s_in * s_w, (shift - zp) * SUM qw) in place,
      // once per channel instead of once per element (in the swapped layout every accumulator register is another channel)
      if (step == 0 && tid < BN) {
        float* pf = reinterpret_cast<float*>(lds + LDS_BYTES) + tid;
        int* pi = reinterpret_cast<int*>(lds + LDS_BYTES) + BN + tid;
        *pf = sin_early * *pf;
        *pi = (shift - zpi) * *pi;
        if constexpr (ASYM) pf[3 * BN] = sin_early * pf[3 * BN];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    if constexpr (ADIR)   // the asm-loaded A fragments of this step are valid from here on (orders their uses behind the wait)
      asm volatile("" : "+v"(areg[ADIR ? U : 0][0]), "+v"(areg[ADIR ? U : 0][1]));
    // (issuing the next loads BEHIND this step's MFMAs - which execute meanwhile - was measured: the operands then arrive
    //  late, +3 % per step; reading all fragments of a step before its first MFMA: no difference)
    if (step + PF < nsteps) issue_next(std::integral_constant<int, (U + PF) % NBUF>{});  // the slot multiply(k-1) released
    stamp(step, 3);
    if (DUAL && step == nfirst) {
      // the shortcut pair is complete: dequantise its sum into registers and start the layer's own sum from zero
      const float sin2 = sg.s_in[0];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float mult = sin2 * par_f(3, j);
        const int corr = (sg.shift - zpi2) * par_i(4, j);
        const float bv2 = sg.bias ? par_f(5, j) : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          extra[DUAL ? j : 0][i] = dequant1(acc[j][i] + corr, mult, bv2);
          acc[j][i] = 0;
        }
      }
    }
    const uint32_t xorw = (DUAL && step < nfirst) ? fs.xorw : fm.xorw;
    const int8_t* base = lds + U * TILE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int sg_ = ks * 2 + hsel;
      i32x4 t;
      if (ADIR) t = areg[ADIR ? U : 0][ks];
      else {
        const int arow = wrow0 + l31;
        t = *reinterpret_cast<const i32x4*>(base + arow * BK + ((sg_ ^ ((arow >> 2) & 3)) << 4));
      }
      const i32x4 af = i32x4{(int)(t.x ^ xorw), (int)(t.y ^ xorw), (int)(t.z ^ xorw), (int)(t.w ^ xorw)};
      if (ASYM) {    // SUM of the int8 operand over the reduction (v_dot4_i32_i8 with a vector of ones)
        s0 = __builtin_amdgcn_sdot4(af.x, 0x01010101, s0, false);
        s0 = __builtin_amdgcn_sdot4(af.y, 0x01010101, s0, false);
        s0 = __builtin_amdgcn_sdot4(af.z, 0x01010101, s0, false);
        s0 = __builtin_amdgcn_sdot4(af.w, 0x01010101, s0, false);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int brow = j * 32 + l31;
        const i32x4 bf = *reinterpret_cast<const i32x4*>(base + TILE_A + brow * BK + ((sg_ ^ ((brow >> 2) & 3)) << 4));
        if (LAB == 4) asm volatile("" ::"v"(af), "v"(bf));
        else if constexpr (SWAP) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, af, acc[j], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[j], 0, 0, 0);
      }
    }
    if (STAMP) {
      asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[NT - 1][15]));
      stamp(step, 4);
    }
  };
  for (int s0 = 0; s0 < nsteps; s0 += NBUF)
    static_for<NBUF>([&](auto u) {
      if (s0 + decltype(u)::value < nsteps) one_step(s0 + decltype(u)::value, u);
    });

  if (STAMP && wgt) wgt[2] = __builtin_readcyclecounter();
  if constexpr (SWAP) {
    // lane (p = l31, h = hsel): register i of block j = channel n0 + 32 j + 16 h + i of pixel m0 + wrow0 + p
    asm volatile("s_barrier" ::: "memory");         // every wave is done with the ring (the code tile is staged there); the
                                                    // constants' pre-pass is at least one barrier old
    const EpiQuant eq(ep, ep.relu != 0);            // code(relu(v)) = max(code(v), code(0))
    constexpr int SROW = BN + 16;                   // staged row: BN code bytes + 16 (conflict-free 16-byte accesses)
    int8_t* stg = lds + wave * (32 * SROW);
    const int8_t* par = lds + LDS_BYTES;
    float s0f = 0.0f;     // ASYM: SUM x' / s_in of this lane's pixel (lanes p and p + 32 hold the two halves of pixel p's operand bytes)
    if constexpr (ASYM) {
      s0 += __shfl_xor(s0, 32, 64);
      s0 += (shift - zpi) * (g.R * g.S * g.C);
      s0f = (float)s0;
    }
    if constexpr (LAB == 7) {
      int any = 0;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) any |= acc[j][i];
      if (any == 0x7fffffff) ep.codes[0] = 1;
      return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int cb = j * 32 + hsel * 16;
      if constexpr (LAB == 6) {
        uint32_t wl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          wl[q] = (uint32_t)(acc[j][4 * q] & 0xff) | ((uint32_t)(acc[j][4 * q + 1] & 0xff) << 8) | ((uint32_t)(acc[j][4 * q + 2] & 0xff) << 16) |
                  ((uint32_t)acc[j][4 * q + 3] << 24);
        *reinterpret_cast<i32x4*>(stg + l31 * SROW + cb) = i32x4{(int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3]};
        continue;
      }
      f32x4 y[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (cb + 4 * q) * 4);
        const i32x4 co = *reinterpret_cast<const i32x4*>(par + (BN + cb + 4 * q) * 4);
        const f32x4 bs = bias ? *reinterpret_cast<const f32x4*>(par + (2 * BN + cb + 4 * q) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        y[q] = f32x4{dequant1(acc[j][4 * q] + co.x, mu.x, bs.x), dequant1(acc[j][4 * q + 1] + co.y, mu.y, bs.y),
                     dequant1(acc[j][4 * q + 2] + co.z, mu.z, bs.z), dequant1(acc[j][4 * q + 3] + co.w, mu.w, bs.w)};
        if constexpr (ASYM) {
          const f32x4 wo = *reinterpret_cast<const f32x4*>(par + (3 * BN + cb + 4 * q) * 4);
          y[q] = f32x4{y[q].x + s0f * wo.x, y[q].y + s0f * wo.y, y[q].z + s0f * wo.z, y[q].w + s0f * wo.w};
        }
      }
      uint32_t wq[4];
      bool uq[4];
      eq.code4n(y, wq, uq);
      *reinterpret_cast<i32x4*>(stg + l31 * SROW + cb) = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // a wave reads back only its own 32 rows
    constexpr int LPR = BN / 16;                    // lanes per staged row
    constexpr int RPI = 64 / LPR;                   // whole rows per wave-instruction (192-wide tiles: 5, four lanes idle)
    const int srow = lane / LPR, sseg = lane % LPR;
#pragma unroll
    for (int it = 0; it < (32 + RPI - 1) / RPI; ++it) {
      const int r = it * RPI + srow;
      const int64_t row = m0 + wrow0 + r;
      if (srow < RPI && r < 32 && row < g.M) {
        const i32x4 c16 = *reinterpret_cast<const i32x4*>(stg + r * SROW + sseg * 16);
        __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(ep.codes + row * g.K + n0 + sseg * 16));
      }
    }
    if (STAMP && wgt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wgt[3] = __builtin_readcyclecounter();
    }
    return;
  }
  // ---- epilogue: one rounding chain  v = (acc + (shift - zp) * SUM qw) * (s_in * s_w[k]) + b[k] ----
  const float sin = s_in[0];
  const EpiQuant eq(ep);
  float mult[NT], p_bias[NT], woff[ASYM ? NT : 1];
  int corr[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    mult[j] = sin * par_f(0, j);
    corr[j] = (shift - zpi) * par_i(1, j);
    p_bias[j] = bias ? par_f(2, j) : 0.0f;
    if (ASYM) woff[ASYM ? j : 0] = sin * par_f(3, j);
  }
  // ASYM: SUM x' of a row = s_in * (SUM q' + (shift - zp) * taps * channels); lanes r and r + 32 hold the two halves of row r.
  // The sums go through a small LDS table into the accumulator layout (register i = row (i & 3) + 8 (i >> 2) + 4 hsel).
  float s0r[ASYM ? 16 : 1];
  if (ASYM) {
    s0 += __shfl_xor(s0, 32, 64);
    s0 += (shift - zpi) * (g.R * g.S * g.C);
    int* tab = reinterpret_cast<int*>(lds + LDS_BYTES + PAR_BYTES) + wave * 32;
    if (hsel == 0) tab[l31] = s0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (a wave reads only its own 32 sums)
#pragma unroll
    for (int i = 0; i < 16; ++i) s0r[ASYM ? i : 0] = (float)tab[(i & 3) + 8 * (i >> 2) + 4 * hsel];
  }
  // observer partials of the stored fp32 values (ep.mm; observer.hip's accumulator: max, min, unsigned max of |x|'s bits = the NaN detector)
  float mm_mx = -__builtin_inff(), mm_mn = __builtin_inff();
  uint32_t mm_ab = 0u;
  auto mm_add = [&](float v) {
    mm_ab = max(mm_ab, __float_as_uint(v) & 0x7fffffffu);
    mm_mx = __builtin_fmaxf(mm_mx, v);
    mm_mn = __builtin_fminf(mm_mn, v);
  };
  auto mm_flush = [&]() {          // wave butterfly, the four waves through LDS, one partial per workgroup
    if (!ep.mm) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mm_ab = max(mm_ab, (uint32_t)__shfl_xor((int)mm_ab, off, 64));
      mm_mx = __builtin_fmaxf(mm_mx, __shfl_xor(mm_mx, off, 64));
      mm_mn = __builtin_fminf(mm_mn, __shfl_xor(mm_mn, off, 64));
    }
    __syncthreads();                 // (every wave is done with the stage)
    float* red = reinterpret_cast<float*>(lds);
    if (lane == 0) {
      red[wave * 3] = mm_mx;
      red[wave * 3 + 1] = mm_mn;
      reinterpret_cast<uint32_t*>(red)[wave * 3 + 2] = mm_ab;
    }
    __syncthreads();
    if (tid == 0) {
      float mx = red[0], mn = red[1];
      uint32_t ab = reinterpret_cast<uint32_t*>(red)[2];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        mx = __builtin_fmaxf(mx, red[k * 3]);
        mn = __builtin_fminf(mn, red[k * 3 + 1]);
        ab = max(ab, reinterpret_cast<uint32_t*>(red)[k * 3 + 2]);
      }
      ep.mm[blockIdx.x] = mx;
      ep.mm[ep.mm_np + blockIdx.x] = mn;
      reinterpret_cast<uint32_t*>(ep.mm)[2 * ep.mm_np + blockIdx.x] = ab;
    }
  };
  if ((g.K & 3) == 0) {
    // through LDS: accumulator layout (lane = channel, register = row) -> row-major, so that each lane stores 16 B and
    // each wave-instruction writes 4 rows x 256 contiguous bytes (the 1x1 layers are bound by this output stream).
    asm volatile("s_barrier" ::: "memory");         // every wave is done reading the operand buffers
    float* stg = reinterpret_cast<float*>(lds) + wave * (32 * EP_LD);
    // codes: 4-byte stores straight from this loop (4 rows x 64 B per instruction) cost the dual kernel 16 % of its time; the words
    // wait in registers instead, go through the (then free) stage once and leave as whole 16-byte pieces of whole rows
    // (not where 16 more registers would spill: the 128-wide single-pair kernels compiled for four workgroups per CU)
    constexpr bool WIDE_OK = NH == 1 || DUAL || ASYM || !ADIR;
    const bool wide_codes = WIDE_OK && ep.codes && (g.K & 15) == 0 && (reinterpret_cast<uintptr_t>(ep.codes) & 15) == 0;
    uint32_t cw[NH][8];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int it = 0; it < 8; ++it) cw[h][it] = 0u;
    static_for<NH>([&](auto h_c) {    // (compile-time indices: the accumulators must stay in registers)
      constexpr int h = decltype(h_c)::value;
      // the shortcut tile's latency hides behind the dequantise-and-stage phase below (64-wide tiles: already here)
      if (ep.residual && !(EARLY_RES && !DUAL)) load_residual(h);
      static_for<2>([&](auto jj_c) {
        constexpr int jj = decltype(jj_c)::value;
        constexpr int j = h * 2 + jj < NT ? h * 2 + jj : NT - 1;
        if (h * 2 + jj >= NT) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (i & 3) + 8 * (i >> 2) + 4 * hsel;
          float v = dequant1(acc[j][i] + corr[j], mult[j], p_bias[j]);
          if (DUAL) v = v + extra[DUAL ? j : 0][i];
          if (ASYM) v = v + s0r[ASYM ? i : 0] * woff[ASYM ? j : 0];
          stg[r * EP_LD + jj * 32 + l31] = v;
        }
      });
      // a wave only reads back what it wrote itself: no block barrier, just the LDS counter
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int col = n0 + h * 64 + ec;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + er;
        const int64_t row = m0 + wrow0 + r;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * EP_LD + ec);
        if (row < g.M && col < g.K) {
          const int64_t at = row * g.K + col;
          if (ep.residual) v = f32x4{v.x + idt[it].x, v.y + idt[it].y, v.z + idt[it].z, v.w + idt[it].w};
          if (ep.relu) v = f32x4{relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)};
          if (out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + at));
          if (ep.mm) { mm_add(v.x); mm_add(v.y); mm_add(v.z); mm_add(v.w); }
          if (ep.codes) {
            const uint32_t c = eq.code4(v);
            if (wide_codes) cw[h][it] = c;
            else __builtin_nontemporal_store(c, reinterpret_cast<uint32_t*>(ep.codes + at));
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next pass overwrites the stage
    });
    if (wide_codes) {
      constexpr int CROW = BN + 16;                 // staged code row: BN bytes + 16 (the wave's 32 rows fit its fp32 stage: 32 * CROW <= 32 * EP_LD * 4)
      static_assert(CROW <= EP_LD * 4, "the code rows are staged where the fp32 rows were");
      int8_t* cst = reinterpret_cast<int8_t*>(stg);
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int it = 0; it < 8; ++it)
          *reinterpret_cast<uint32_t*>(cst + (it * 4 + er) * CROW + h * 64 + ec) = cw[h][it];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int LPR = BN / 16;                  // lanes per staged row
      const int srow = lane / LPR, sseg = lane % LPR;
#pragma unroll
      for (int it = 0; it < 32 / (64 / LPR); ++it) {
        const int r = it * (64 / LPR) + srow;
        const int64_t row = m0 + wrow0 + r;
        const i32x4 c16 = *reinterpret_cast<const i32x4*>(cst + r * CROW + sseg * 16);
        if (row < g.M && n0 + sseg * 16 < g.K) __builtin_nontemporal_store(c16, reinterpret_cast<i32x4*>(ep.codes + row * g.K + n0 + sseg * 16));
      }
    }
    if (STAMP && wgt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wgt[3] = __builtin_readcyclecounter();
    }
    mm_flush();
    return;
  }
  // K % 4 != 0 (e.g. a 1000-class head): element-wise stores straight from the accumulator layout
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 32 + l31;
    if (col >= g.K) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t row = m0 + wrow0 + (i & 3) + 8 * (i >> 2) + 4 * hsel;
      if (row >= g.M) continue;
      const int64_t at = row * g.K + col;
      float v = dequant1(acc[j][i] + corr[j], mult[j], p_bias[j]);
      if (DUAL) v = v + extra[DUAL ? j : 0][i];
      if (ASYM) v = v + s0r[ASYM ? i : 0] * woff[ASYM ? j : 0];
      if (ep.residual) v = v + ep.residual[at];
      if (ep.relu) v = relu_nan(v);
      if (out) __builtin_nontemporal_store(v, out + at);
      if (ep.mm) mm_add(v);
      if (ep.codes) ep.codes[at] = (uint8_t)eq.exact(v);
    }
  }
  mm_flush();
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

// What to launch.  The product dispatch (conv_plan) is a pure function of the problem; the tuning entry point of the lab
// library (DLMCQ_LAB builds only) passes its own.
struct ConvPlan {
  int bn;       // tile width: 64 or 128 channels
  bool adir;    // activations straight to registers (see the kernel)
  bool swap;    // codes-only layers in the swapped accumulator layout (see the kernel)
  bool halo = true;   // 3x3 / stride 1 / pad 1 codes-only layers on the halo-tile kernel of conv3x3_i8.hip where it applies
};

static ConvPlan conv_plan(int64_t C, int64_t K, int64_t R, int64_t S, bool dual) {
  ConvPlan p;
  p.swap = true;
  // 64-wide tiles: narrow outputs, widths that are no multiple of 128, and the dual kernel on short reductions (two
  // operand pairs + the shortcut sum keep 190 VGPRs at 128 columns: 655 vs 701 us on ResNet-50's first dual layer)
  const bool dual_short = dual && K <= 256 && R * S * C <= 256;
  p.bn = (dual_short || K <= 64 || (K % 128) != 0) ? 64 : 128;
  // (128 x 256 tiles - a third fewer operand bytes per MAC - gained 10-17 % on the 7x7 stage of the module path only and
  //  lost 40 % wherever the epilogue is fused: two workgroups per CU cannot hide a tile's latency chain.  Not built.)
  // long 1x1 reductions (>= 512 channels, or 256 into 64): both operands through the ring, 64-wide tiles (full-line DMA reads of
  // the long activation rows beat fragment-shaped loads: 5-15 % on ResNet-50's 256->64 ... 2048->512 layers)
  if (R * S == 1 && !dual && K % 64 == 0 && ((C >= 512 && K <= 512) || (C >= 256 && K <= 64))) {
    // (with the swapped epilogue the 1024 / 2048-deep reductions are 2-5 % faster on 128-wide tiles: 59.2 vs 62.4, 60.5 vs 61.9 us)
    p.bn = (C >= 1024 && K % 128 == 0) ? 128 : 64;
    p.adir = false;
    if (C >= 2048 && K % 256 == 0) p.bn = 256;      // (7x7 stage: 54.4 vs 61.2 us; conv_launch falls back to 128 when the layer is not codes-only)
  } else if (R * S == 1 && !dual && K % 64 == 0 && K >= 4 * C && C <= 256) {
    // 1x1 expansions (the block-end layers: HBM streams with a short reduction): 64-wide tiles, 3-6 % faster at every stage
    // (tools/conv_lab.py); activations through the ring once a row is >= 128 bytes
    p.bn = 64;
    p.adir = C < 128;
  } else {
    p.adir = true;
    // K-loop-bound 3x3 layers with 72 K steps per tile (512 channels): 256-wide tiles - a third fewer operand bytes per MAC through
    // the CU's vector-memory path, at two workgroups per CU - pay once the epilogue is the swapped one: 512->512 at 7^2 111.7 -> 91.5 us
    // on cold buffers (tools/conv_lab.py), 95 -> 81 us in the plan; 256->256 at 14^2 gains 6 % cold and nothing in the plan (128 kept)
    if (R * S > 1 && !dual && C >= 512 && K % 256 == 0) p.bn = 256;
  }
  return p;
}

static int conv_launch(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                       const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N, int64_t H,
                       int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                       int32_t dilation, int32_t x_is_unsigned, dlmcq_stream_t stream, const ConvEpi& ep_in = ConvEpi{},
                       const ConvSeg2* seg2 = nullptr, const ConvPlan* forced = nullptr, int64_t* mm_count = nullptr) {
  // (observer partials - ep.mm - come from the tiled kernel's fp32 epilogue only: a call another kernel takes reports 0 partials)
  ConvEpi ep = ep_in;
  if (mm_count) *mm_count = 0;
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
  ConvPlan plan = forced ? *forced : conv_plan(C, K, R, S, seg2 != nullptr);
  // asymmetric weights always take their activations straight to registers (the row sums come from the fragments), so conv_plan's
  // 64-wide choice for deep 1x1 reductions - made for the ring-fed kernels - only multiplies the activation re-reads there:
  // MobileOne-S1's 512 -> 512 layers at 14^2, batch 1024: 150 -> 113 us on 128-wide tiles
  if (!forced && ep.w_off && plan.bn == 64 && K > 64 && K % 128 == 0) plan.bn = 128;
  // ... and widths of 192, 576, ... (no multiple of 128) 192-wide ones - codes-only layers with the swapped epilogue (MobileOne-S1's
  // 192 -> 192 layers at 28^2: one tile column fewer, a third fewer re-reads of the activations)
  if (!forced && ep.w_off && plan.bn == 64 && K % 192 == 0 && plan.swap && ep.codes && !out && !ep.residual && aligned16(ep.codes)) plan.bn = 192;
  // the specialised kernels, unless the caller (DLMCQ_FORCE_TILED) or a lab plan keeps the call on this file's kernel;
  // DLMCQ_ROUTE_ONLY: the decision is the answer, nothing is launched
  const bool special = plan.halo && !(ep.ctl & DLMCQ_FORCE_TILED), route_only = (ep.ctl & DLMCQ_ROUTE_ONLY) != 0;
  float* const mm_req = ep.mm;
  ep.mm = nullptr;                    // (the specialised kernels below do not write partials)
  // (only the block-end kernel knows the chunk-major form of the fp32 block tensors: a call that carries the bits and would land
  //  elsewhere is refused, never silently read row-major.  Callers ask first: DLMCQ_ROUTE_ONLY without the bits)
  if ((ep.ctl & (DLMCQ_FP32_IN_CHUNK_MAJOR | DLMCQ_FP32_OUT_CHUNK_MAJOR)) &&
      !(special && conv_pwr_applies(N, H, W, C, K, R, S, stride, pad, dilation, ep, out, seg2)))
    return DLMCQ_EINVAL;
  if (special && conv_pw_applies(N, H, W, C, K, R, S, stride, pad, dilation, ep, out, seg2 != nullptr))
    return route_only ? DLMCQ_ROUTE_PW : conv_pw_launch(xs, w, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, shift, ep, st);
  if (special && conv_pwr_applies(N, H, W, C, K, R, S, stride, pad, dilation, ep, out, seg2))
    return route_only ? DLMCQ_ROUTE_PWR
                      : conv_pwr_launch(xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, stride, shift, ep, st, seg2);
  if (special && conv3x3_halo_applies(N, H, W, C, K, R, S, stride, pad, dilation, ep, out, seg2 != nullptr)) {
    // ... persistent and pipelined across tiles on request (DLMCQ_PIPELINED: csrc/conv3x3_pipe_i8.hip - bit-identical, measured slower)
    if ((ep.ctl & DLMCQ_PIPELINED) && conv3x3_pipe_applies(N, H, W, C, K, stride, ep, device_cus()))
      return route_only ? DLMCQ_ROUTE_HALO3X3_PIPE
                        : conv3x3_pipe_launch(xs, w, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, shift, ep, st, device_cus());
    return route_only ? DLMCQ_ROUTE_HALO3X3
                      : conv3x3_halo_launch(xs, w, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, stride, shift, ep, st);
  }
  // 256-wide tiles exist for the swapped codes-only layers only (one third fewer operand bytes per MAC, two workgroups per CU)
  const bool swap_ok = plan.swap && ep.codes && !out && !ep.residual && K % plan.bn == 0 && aligned16(ep.codes);
  // (192-wide tiles: the swapped asymmetric codes-only instantiation is the only one - a forced plan that asks for them anywhere
  //  else would compute nblk_n for 192 and launch the 128-wide kernel, leaving channels unwritten)
  if (plan.bn != 64 && plan.bn != 128 && plan.bn != 256 && !(plan.bn == 192 && ep.w_off && swap_ok && !seg2)) return DLMCQ_EINVAL;
  if (plan.bn == 256 && (seg2 || ep.w_off || !swap_ok)) {
    if (forced) return DLMCQ_EINVAL;
    ConvPlan p2 = plan;                  // (fp32 outputs, shortcuts, asymmetric weights: the 128-wide kernels)
    p2.bn = 128;
    return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                       x_is_unsigned, stream, ep_in, seg2, &p2, mm_count);
  }
  g.nblk_m = (int)((M + CV_BM - 1) / CV_BM);
  g.nblk_n = (int)((K + plan.bn - 1) / plan.bn);
  const int64_t nwg = (int64_t)g.nblk_m * g.nblk_n;
  if (nwg >= (1ll << 31)) return DLMCQ_ERANGE;
  ConvSeg2 s2{};
  if (seg2) {
    s2 = *seg2;
    s2.g.nblk_m = g.nblk_m;
    s2.g.nblk_n = g.nblk_n;
    if (s2.g.M != g.M || s2.g.K != g.K || s2.g.P != g.P || s2.g.Q != g.Q || ep.w_off) return DLMCQ_EINVAL;
  }
  if (route_only) return DLMCQ_ROUTE_TILED;
  if (mm_req && out) {                             // one partial per workgroup of this launch
    ep.mm = mm_req;
    ep.mm_np = (int)nwg;
    if (mm_count) *mm_count = nwg;
  }
#define DLMCQ_CONV_ARGS dim3((uint32_t)nwg), dim3(256), 0, st, xs, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, s2
  if (seg2) {
    if (plan.bn == 64) hipLaunchKernelGGL((conv_i8_mfma_kernel<64, true, true>), DLMCQ_CONV_ARGS);
    else hipLaunchKernelGGL((conv_i8_mfma_kernel<128, true, true>), DLMCQ_CONV_ARGS);
  } else if (ep.w_off && plan.swap && ep.codes && !out && !ep.residual && K % plan.bn == 0 && aligned16(ep.codes)) {
    if (plan.bn == 64) hipLaunchKernelGGL((conv_i8_mfma_kernel<64, false, true, true, 0, true>), DLMCQ_CONV_ARGS);
    else if (plan.bn == 192) hipLaunchKernelGGL((conv_i8_mfma_kernel<192, false, true, true, 0, true>), DLMCQ_CONV_ARGS);
    else hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, true, true, 0, true>), DLMCQ_CONV_ARGS);
  } else if (ep.w_off) {     // asymmetric per-channel weights (activations direct: the row sums come from their fragments)
    if (plan.bn == 64) hipLaunchKernelGGL((conv_i8_mfma_kernel<64, false, true, true>), DLMCQ_CONV_ARGS);
    else hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, true, true>), DLMCQ_CONV_ARGS);
  } else if (swap_ok && plan.bn == 256) {
    if (plan.adir) hipLaunchKernelGGL((conv_i8_mfma_kernel<256, false, true, false, 0, true>), DLMCQ_CONV_ARGS);
    else hipLaunchKernelGGL((conv_i8_mfma_kernel<256, false, false, false, 0, true>), DLMCQ_CONV_ARGS);
  } else if (swap_ok) {
    if (plan.bn == 64) {
      if (plan.adir) hipLaunchKernelGGL((conv_i8_mfma_kernel<64, false, true, false, 0, true>), DLMCQ_CONV_ARGS);
      else hipLaunchKernelGGL((conv_i8_mfma_kernel<64, false, false, false, 0, true>), DLMCQ_CONV_ARGS);
    } else {
      if (plan.adir) hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, true, false, 0, true>), DLMCQ_CONV_ARGS);
      else hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, false, false, 0, true>), DLMCQ_CONV_ARGS);
    }
  } else if (!plan.adir) {
    if (plan.bn == 64) hipLaunchKernelGGL((conv_i8_mfma_kernel<64, false, false>), DLMCQ_CONV_ARGS);
    else hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, false>), DLMCQ_CONV_ARGS);
  } else {
    if (plan.bn == 64) hipLaunchKernelGGL((conv_i8_mfma_kernel<64, false, true>), DLMCQ_CONV_ARGS);
    else hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, true>), DLMCQ_CONV_ARGS);
  }
#undef DLMCQ_CONV_ARGS
  return launch_status();
}

static ConvEpi make_epi(const float* residual, int32_t relu, void* codes, const float* q_scale, const float* q_zero_point,
                        int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g) {
  ConvEpi ep{};
  ep.residual = residual;
  ep.relu = relu != 0;
  ep.codes = static_cast<uint8_t*>(codes);
  ep.q_scale = q_scale;
  ep.q_zp = q_zero_point;
  ep.q_lo = (float)q_lo;
  ep.q_hi = (float)q_hi;
  ep.q_g = q_ste_g;
  if (!epi_set_form(ep, q_form, q_lo, q_hi)) ep.q_form = -1;     // (conv_launch refuses it)
  return ep;
}

extern "C" int dlmcq_conv2d_i8_nhwc_f32(const void* x, const int8_t* w, float* out, const float* bias,
                                        const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                        const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                                        int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                                        int32_t x_is_unsigned, dlmcq_stream_t stream) {
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream);
}

extern "C" int dlmcq_conv2d_i8_nhwc_fused(const void* x, const int8_t* w, float* out, const float* bias,
                                          const int32_t* wsum, const float* in_scale, const float* in_zero_point,
                                          const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                                          int64_t R, int64_t S, int32_t stride, int32_t pad, int32_t dilation,
                                          int32_t x_is_unsigned, const float* residual, int32_t relu, void* codes,
                                          const float* q_scale, const float* q_zero_point, int32_t q_lo, int32_t q_hi,
                                          int32_t q_form, float q_ste_g, dlmcq_stream_t stream) {
  const ConvEpi ep = make_epi(residual, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g);
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, ep);
}

extern "C" size_t dlmcq_conv2d_i8_observed_partials(int64_t M, int64_t K) {
  if (M < 0 || K < 1) return 0;
  return (size_t)(((M + CV_BM - 1) / CV_BM) * ((K + 63) / 64));      // the tiled kernel's finest tiling: 128 pixels x 64 channels per workgroup
}

extern "C" int dlmcq_conv2d_i8_nhwc_fused_observed(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                                   const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                                                   int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride,
                                                   int32_t pad, int32_t dilation, int32_t x_is_unsigned, const float* residual,
                                                   int32_t relu, void* codes, const float* q_scale, const float* q_zero_point,
                                                   int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g, float* partials,
                                                   int64_t partials_capacity, int64_t* partials_count, dlmcq_stream_t stream) {
  if (!partials || !partials_count || !out) return DLMCQ_EINVAL;
  const int64_t P = (H + 2 * pad - dilation * (R - 1) - 1) / (stride > 0 ? stride : 1) + 1;
  const int64_t Q = (W + 2 * pad - dilation * (S - 1) - 1) / (stride > 0 ? stride : 1) + 1;
  if (partials_capacity < (int64_t)dlmcq_conv2d_i8_observed_partials(N * P * Q, K)) return DLMCQ_ESCRATCH;
  ConvEpi ep = make_epi(residual, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g);
  ep.mm = partials;
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, ep, nullptr, nullptr, partials_count);
}

extern "C" int dlmcq_conv2d_i8_nhwc_asym(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                         const float* in_scale, const float* in_zero_point, const float* w_scale,
                                         const float* w_offset, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R,
                                         int64_t S, int32_t stride, int32_t pad, int32_t dilation, int32_t x_is_unsigned,
                                         const float* residual, int32_t relu, void* codes, const float* q_scale,
                                         const float* q_zero_point, int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g,
                                         dlmcq_stream_t stream) {
  if (!w_offset) return DLMCQ_EINVAL;
  ConvEpi ep = make_epi(residual, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g);
  ep.w_off = w_offset;
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, ep);
}

static int make_seg2(ConvSeg2& s2, int64_t N, int64_t K, const void* x2, const int8_t* w2, const float* bias2,
                     const int32_t* wsum2, const float* in_scale2, const float* in_zero_point2, const float* w_scale2,
                     int64_t H2, int64_t W2, int64_t C2, int64_t R2, int64_t S2, int32_t stride2, int32_t pad2,
                     int32_t dilation2, int32_t x2_is_unsigned) {
  if (H2 < 1 || W2 < 1 || C2 < 1 || R2 < 1 || S2 < 1 || stride2 < 1 || pad2 < 0 || dilation2 < 1 || C2 % CV_BK != 0)
    return DLMCQ_EINVAL;
  if (N > 0 && (!x2 || !w2 || !wsum2 || !in_scale2 || !w_scale2)) return DLMCQ_EINVAL;
  if (!aligned16(x2) || !aligned16(w2)) return DLMCQ_EALIGN;
  if (N * H2 * W2 * C2 >= (1ll << 40)) return DLMCQ_ERANGE;
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
  return DLMCQ_OK;
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
  ConvSeg2 s2{};
  const int rc = make_seg2(s2, N, K, x2, w2, bias2, wsum2, in_scale2, in_zero_point2, w_scale2, H2, W2, C2, R2, S2, stride2,
                           pad2, dilation2, x2_is_unsigned);
  if (rc != DLMCQ_OK) return rc;
  const ConvEpi ep = make_epi(nullptr, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g);
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, ep, &s2);
}

#ifdef DLMCQ_LAB
// phase stamps of one wave (tools/conv_trace.py): `trace` receives 24 x 8 uint64 shader-clock values
extern "C" int dlmcq_x_conv2d_i8_trace(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                       const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                                       int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride,
                                       int32_t pad, int32_t dilation, int32_t x_is_unsigned, void* codes, const float* q_scale,
                                       dlmcq_stream_t stream, void* trace) {
  ConvEpi ep = make_epi(static_cast<const float*>(trace), 1, codes, q_scale, nullptr, 0, 255, DLMCQ_FORM_ZEROPOINT, 0.0f);
  const int64_t P = (H + 2 * pad - dilation * (R - 1) - 1) / stride + 1, Q = (W + 2 * pad - dilation * (S - 1) - 1) / stride + 1;
  if (C % CV_BK || K % 128 || P < 1 || Q < 1) return DLMCQ_EINVAL;
  ConvGeom g;
  g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K; g.R = (int)R; g.S = (int)S;
  g.stride = stride; g.pad = pad; g.dil = dilation; g.P = (int)P; g.Q = (int)Q; g.M = N * P * Q;
  g.qdiv = make_fastdiv((uint32_t)Q);
  g.pdiv = make_fastdiv((uint32_t)P);
  g.nblk_m = (int)((g.M + CV_BM - 1) / CV_BM);
  g.nblk_n = (int)(K / 128);
  hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, true, false, 1>), dim3((uint32_t)((int64_t)g.nblk_m * g.nblk_n)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const int8_t*>(x), w, out, bias, wsum, in_scale,
                     in_zero_point, w_scale, g, x_is_unsigned ? 128 : 0, ep, ConvSeg2{});
  return launch_status();
}

// the same stamps for the dual kernel (a block's last 1x1 convolution + the 1x1 / stride-s convolution on its shortcut; fp32 out + codes)
extern "C" int dlmcq_x_conv2d_i8_dual_trace(const void* x, const int8_t* w, const int32_t* wsum, const float* w_scale, const void* x2,
                                            const int8_t* w2, const int32_t* wsum2, const float* w_scale2, const float* in_scale,
                                            const float* in_zero_point, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K,
                                            int64_t H2, int64_t W2, int64_t C2, int32_t stride2, float* out, void* codes,
                                            const float* q_scale, dlmcq_stream_t stream, void* trace) {
  ConvSeg2 s2{};
  const int rc = make_seg2(s2, N, K, x2, w2, nullptr, wsum2, in_scale, in_zero_point, w_scale2, H2, W2, C2, 1, 1, stride2, 0, 1, 1);
  if (rc != DLMCQ_OK) return rc;
  ConvEpi ep = make_epi(static_cast<const float*>(trace), 1, codes, q_scale, nullptr, 0, 255, DLMCQ_FORM_ZEROPOINT, 0.0f);
  if (C % CV_BK || C2 % CV_BK || K % 128) return DLMCQ_EINVAL;
  ConvGeom g;
  g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K; g.R = 1; g.S = 1;
  g.stride = 1; g.pad = 0; g.dil = 1; g.P = (int)H; g.Q = (int)W; g.M = N * H * W;
  g.qdiv = make_fastdiv((uint32_t)W);
  g.pdiv = make_fastdiv((uint32_t)H);
  g.nblk_m = (int)((g.M + CV_BM - 1) / CV_BM);
  g.nblk_n = (int)(K / 128);
  s2.g.nblk_m = g.nblk_m;
  s2.g.nblk_n = g.nblk_n;
  if (s2.g.M != g.M) return DLMCQ_EINVAL;
  hipLaunchKernelGGL((conv_i8_mfma_kernel<128, true, true, false, 1>), dim3((uint32_t)((int64_t)g.nblk_m * g.nblk_n)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const int8_t*>(x), w, out, nullptr, wsum, in_scale,
                     in_zero_point, w_scale, g, 128, ep, s2);
  return launch_status();
}

// ---- lab library only (libdlmcq_lab.so, `make lab`): the same calls with an explicit tile plan, and the persistent
// kernel of lab/conv_i8_pp.hip, for A/B measurements in one process (tools/conv_lab.py).  Not part of the ABI. ----
int dlmcq_conv_pp_launch(const int8_t* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                         const float* in_scale, const float* in_zero_point, const float* w_scale, dlmcq::ConvGeom g, int shift,
                         const dlmcq::ConvEpi& ep, const dlmcq::ConvSeg2* seg2, int bn, int nbuf, int wps, hipStream_t st);

extern "C" int dlmcq_x_conv2d_i8_tuned(const void* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                                       const float* in_scale, const float* in_zero_point, const float* w_scale, int64_t N,
                                       int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride,
                                       int32_t pad, int32_t dilation, int32_t x_is_unsigned, const float* residual,
                                       int32_t relu, void* codes, const float* q_scale, const float* q_zero_point,
                                       int32_t q_lo, int32_t q_hi, int32_t q_form, float q_ste_g, dlmcq_stream_t stream,
                                       int32_t bn, int32_t adir, int32_t pp_nbuf, int32_t pp_wps) {
  const ConvEpi ep = make_epi(residual, relu, codes, q_scale, q_zero_point, q_lo, q_hi, q_form, q_ste_g);
  if (pp_wps > 0) {
    const int64_t P = (H + 2 * pad - dilation * (R - 1) - 1) / stride + 1, Q = (W + 2 * pad - dilation * (S - 1) - 1) / stride + 1;
    if (C % CV_BK || K % bn || P < 1 || Q < 1 || N * P * Q >= (1ll << 31)) return DLMCQ_EINVAL;
    ConvGeom g;
    g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K; g.R = (int)R; g.S = (int)S;
    g.stride = stride; g.pad = pad; g.dil = dilation; g.P = (int)P; g.Q = (int)Q; g.M = N * P * Q;
    g.qdiv = make_fastdiv((uint32_t)Q);
    g.pdiv = make_fastdiv((uint32_t)P);
    return dlmcq_conv_pp_launch(reinterpret_cast<const int8_t*>(x), w, out, bias, wsum, in_scale, in_zero_point, w_scale, g,
                                x_is_unsigned ? 128 : 0, ep, nullptr, bn, pp_nbuf, pp_wps, reinterpret_cast<hipStream_t>(stream));
  }
  if (pp_wps <= -100) {    // the halo kernel's what-bounds-the-step variants (LAB = -pp_wps - 100); 1 = stamps into `residual`
    if (!conv3x3_halo_applies(N, H, W, C, K, R, S, stride, pad, dilation, ConvEpi{nullptr, nullptr, ep.codes}, nullptr, false)) return DLMCQ_EINVAL;
    ConvEpi e2 = ep;
    e2.residual = nullptr;
    return conv3x3_halo_launch(reinterpret_cast<const int8_t*>(x), w, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K,
                               stride, x_is_unsigned ? 128 : 0, e2, reinterpret_cast<hipStream_t>(stream), -pp_wps - 100,
                               const_cast<float*>(residual));
  }
  if (pp_wps <= -40 && pp_wps > -100) {   // ... of the pointwise kernel (LAB = -pp_wps - 40; `bias` doubles as the weight offsets, `residual` as the stamp buffer)
    ConvEpi e2 = ep;
    e2.residual = nullptr;
    e2.w_off = bias;
    if (!conv_pw_applies(N, H, W, C, K, R, S, stride, pad, dilation, e2, out, false)) return DLMCQ_EINVAL;
    return conv_pw_launch(reinterpret_cast<const int8_t*>(x), w, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K,
                          x_is_unsigned ? 128 : 0, e2, reinterpret_cast<hipStream_t>(stream), -pp_wps - 40, const_cast<float*>(residual));
  }
  if (pp_wps <= -20 && pp_wps > -100) {   // ... of the swapped asymmetric kernels (LAB = -pp_wps - 20, 0 = as built; `bias` doubles as the weight offsets)
    const int64_t P = (H + 2 * pad - dilation * (R - 1) - 1) / stride + 1, Q = (W + 2 * pad - dilation * (S - 1) - 1) / stride + 1;
    if (C % CV_BK || K % bn || (bn != 128 && bn != 192) || P < 1 || Q < 1 || N * P * Q >= (1ll << 31) || !ep.codes || out) return DLMCQ_EINVAL;
    ConvGeom g;
    g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K; g.R = (int)R; g.S = (int)S;
    g.stride = stride; g.pad = pad; g.dil = dilation; g.P = (int)P; g.Q = (int)Q; g.M = N * P * Q;
    g.qdiv = make_fastdiv((uint32_t)Q);
    g.pdiv = make_fastdiv((uint32_t)P);
    g.nblk_m = (int)((g.M + CV_BM - 1) / CV_BM);
    g.nblk_n = (int)(K / bn);
    const dim3 grid((uint32_t)((int64_t)g.nblk_m * g.nblk_n));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int8_t* xx = reinterpret_cast<const int8_t*>(x);
    const int shift = x_is_unsigned ? 128 : 0;
    ConvEpi e2 = ep;
    e2.residual = nullptr;
    e2.w_off = bias;
#define DLMCQ_LABK(B, V) hipLaunchKernelGGL((conv_i8_mfma_kernel<B, false, true, true, V, true>), grid, dim3(256), 0, st, xx, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, e2, ConvSeg2{})
#define DLMCQ_LABB(V) if (bn == 128) DLMCQ_LABK(128, V); else DLMCQ_LABK(192, V)
    switch (-pp_wps - 20) {
      case 0: DLMCQ_LABB(0); break;
      case 2: DLMCQ_LABB(2); break;
      case 3: DLMCQ_LABB(3); break;
      case 4: DLMCQ_LABB(4); break;
      case 6: DLMCQ_LABB(6); break;
      case 7: DLMCQ_LABB(7); break;
      default: return DLMCQ_EINVAL;
    }
#undef DLMCQ_LABB
#undef DLMCQ_LABK
    return launch_status();
  }
  if (pp_wps < 0) {   // what-bounds-the-step variants of the 128-wide direct-A kernel (LAB = -pp_wps)
    const int64_t P = (H + 2 * pad - dilation * (R - 1) - 1) / stride + 1, Q = (W + 2 * pad - dilation * (S - 1) - 1) / stride + 1;
    if (C % CV_BK || K % 128 || P < 1 || Q < 1 || N * P * Q >= (1ll << 31)) return DLMCQ_EINVAL;
    ConvGeom g;
    g.N = (int)N; g.H = (int)H; g.W = (int)W; g.C = (int)C; g.K = (int)K; g.R = (int)R; g.S = (int)S;
    g.stride = stride; g.pad = pad; g.dil = dilation; g.P = (int)P; g.Q = (int)Q; g.M = N * P * Q;
    g.qdiv = make_fastdiv((uint32_t)Q);
    g.pdiv = make_fastdiv((uint32_t)P);
    g.nblk_m = (int)((g.M + CV_BM - 1) / CV_BM);
    g.nblk_n = (int)(K / 128);
    const dim3 grid((uint32_t)((int64_t)g.nblk_m * g.nblk_n));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int8_t* xx = reinterpret_cast<const int8_t*>(x);
    const int shift = x_is_unsigned ? 128 : 0;
#define DLMCQ_LABK(V) hipLaunchKernelGGL((conv_i8_mfma_kernel<128, false, true, false, V>), grid, dim3(256), 0, st, xx, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, ConvSeg2{})
    switch (-pp_wps) {
      case 2: DLMCQ_LABK(2); break;
      case 3: DLMCQ_LABK(3); break;
      case 4: DLMCQ_LABK(4); break;
      case 5: DLMCQ_LABK(5); break;
      default: return DLMCQ_EINVAL;
    }
#undef DLMCQ_LABK
    return launch_status();
  }
  const ConvPlan plan{bn, (adir & 1) != 0, (adir & 2) == 0, (adir & 4) != 0};     // adir bit 1: the unswapped epilogue; bit 2: the halo kernel (A/B runs)
  return conv_launch(x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, N, H, W, C, K, R, S, stride, pad, dilation,
                     x_is_unsigned, stream, ep, nullptr, &plan);
}
#endif
