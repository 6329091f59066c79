// Persistent, cross-tile pipelined int8 implicit-GEMM convolution / linear for gfx950 (the default K9 kernel).
//
// Same arithmetic and layouts as conv_i8.hip (see its header): NHWC int8 activations, KRSC int8 weights, exact int32
// accumulation on v_mfma_i32_32x32x32_i8, one fp32 rounding chain, the fused epilogue of conv_epilogue.h.
//
// What is different is WHEN things happen.  In a one-tile-per-workgroup kernel every tile pays a serial latency chain
// (per-channel constants -> first operands -> MFMA -> shortcut tile -> stores) and hides it only behind the other
// workgroups of its CU; at ResNet sizes most tiles have 1-8 K steps, so the chain IS the tile.  Here ONE workgroup per
// CU (one wave per SIMD: 256 VGPRs + 256 accumulator registers each) owns a contiguous run of output tiles
// (consecutive column blocks of the same 128 pixels) and treats their K steps as ONE stream:
//   * operands run PF = NBUF - 1 K steps ahead of the multiplies, ACROSS tile boundaries (weights: LDS-DMA ring shared
//     by the waves; activations: each lane's own MFMA fragments, a register ring);
//   * the next tile's per-channel constants and its whole fp32 shortcut tile are requested during this tile's epilogue,
//     one tile ahead (64 KB of shortcut in flight per CU: the epilogue is an HBM stream and HBM needs that much);
//   * EVERY vector-memory load is inline asm that the compiler does not see, counted in one software counter: vmcnt
//     retires in order, so "wait for load X" is  s_waitcnt vmcnt(ops issued since X)  - beside an LDS-DMA in flight
//     hipcc would otherwise drain the queue (vmcnt(0)) at the first use of any load it knows about;
//   * every such load lands in an explicitly named accumulator register (lab/agpr_asm.h) and is moved to a VGPR only after
//     its wait: a register the compiler allocates may be copied, spilled or re-used while the load is still in flight
//     (tools/lint_agpr.py checks the compiled listing: no compiler access to those registers, no spills).
// Stores stay ordinary (the compiler never waits for a store); they are NOT counted, which only makes a wait that
// follows an epilogue conservative.
#include "../conv_i8_common.h"
#include "agpr_asm.h"

namespace dlmcq {

// s_waitcnt vmcnt(n') with the largest n' <= n of a fixed ladder (waiting for more is always safe).  n is wave-uniform.
#define DLMCQ_W(k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
__device__ __forceinline__ void wait_vm(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n >= 16) {
    if (n >= 32) {
      if (n >= 48) { if (n >= 63) DLMCQ_W(63); else if (n >= 56) DLMCQ_W(56); else if (n >= 52) DLMCQ_W(52); else DLMCQ_W(48); }
      else { if (n >= 44) DLMCQ_W(44); else if (n >= 40) DLMCQ_W(40); else if (n >= 36) DLMCQ_W(36); else DLMCQ_W(32); }
    } else {
      if (n >= 24) { if (n >= 30) DLMCQ_W(30); else if (n >= 28) DLMCQ_W(28); else if (n >= 26) DLMCQ_W(26); else DLMCQ_W(24); }
      else { if (n >= 22) DLMCQ_W(22); else if (n >= 20) DLMCQ_W(20); else if (n >= 18) DLMCQ_W(18); else DLMCQ_W(16); }
    }
  } else if (n >= 8) {
    if (n >= 12) { if (n >= 15) DLMCQ_W(15); else if (n >= 14) DLMCQ_W(14); else DLMCQ_W(12); }
    else { if (n >= 10) DLMCQ_W(10); else if (n >= 9) DLMCQ_W(9); else DLMCQ_W(8); }
  } else if (n >= 4) {
    if (n >= 6) { if (n >= 7) DLMCQ_W(7); else DLMCQ_W(6); }
    else { if (n >= 5) DLMCQ_W(5); else DLMCQ_W(4); }
  } else {
    if (n >= 2) { if (n >= 3) DLMCQ_W(3); else DLMCQ_W(2); }
    else { if (n >= 1) DLMCQ_W(1); else DLMCQ_W(0); }
  }
}
#undef DLMCQ_W

// BM = 128 pixels x BN channels per tile, 4 waves (32 pixels each), BK = 64.  DUAL: a second (input, weight) pair - the
// shortcut convolution - reduced FIRST into the same tile.  NBUF: depth of the operand rings.  WPS: waves per SIMD the
// kernel is built for (1: 256 VGPRs + 256 AGPRs per lane, one workgroup per CU; 2: 128 + 128, two workgroups per CU).
template <int BN, bool DUAL, int NBUF, int WPS>
__global__ __launch_bounds__(256, WPS) void conv_i8_pp_kernel(const int8_t* __restrict__ x, const int8_t* __restrict__ w,
                                                            float* __restrict__ out, const float* __restrict__ bias,
                                                            const int32_t* __restrict__ wsum, const float* __restrict__ s_in,
                                                            const float* __restrict__ zp_in, const float* __restrict__ s_w,
                                                            ConvGeom g, int shift, ConvEpi ep, ConvSeg2 sg) {
  constexpr int BM = 128, BK = 64, PF = NBUF - 1, KS = BK / 32, NT = BN / 32, BI = BN / 64;
  constexpr int TILE_B = BN * BK;                     // bytes per ring slot
  constexpr int EP_LD = 68;                           // floats per staged epilogue row (64 + 4 pad)
  constexpr int RING = NBUF * TILE_B, STG = 4 * 32 * EP_LD * 4;
  constexpr int NH = NT / 2;                          // epilogue passes of 64 channels
  constexpr int GROUP = BI + KS;                      // vector-memory operations per K step per wave
  constexpr int NSEG = DUAL ? 2 : 1;
  static_assert(BN == 64 || BN == 128, "tile width");
  // accumulator-register map (lab/agpr_asm.h), counted DOWN from a255 (the compiler allocates upwards from a0): what is in
  // flight lives here, out of the compiler's reach
  constexpr int AQ_RES = 0;                            // quads: the fp32 shortcut tile, NH passes x 8 rows (not in DUAL kernels)
  constexpr int AQ_FRAG = DUAL ? 0 : NH * 8;           // quads: A fragments, NBUF slots x KS
  constexpr int A1_PAR = (AQ_FRAG + NBUF * KS) * 4;    // singles: per-channel constants, 3 (DUAL: 6) x NT
  constexpr int ATOP = WPS == 1 ? 256 : 128;           // WPS waves per SIMD: 512 / WPS registers per lane, half of them AGPRs
  static_assert(A1_PAR + 6 * NT <= ATOP - NT * 16 - 16, "the low AGPRs are left to the compiler (MFMA accumulators, its own spills)");
  static_assert(RING + STG <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(1024))) int8_t lds[RING + STG];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hsel = lane >> 5, l31 = lane & 31;

  // ---- this workgroup's run of tiles (column blocks fastest); workgroups of one XCD (equal blockIdx % 8) are neighbours ----
  const uint32_t G = gridDim.x, bid = blockIdx.x;
  const uint32_t T = (uint32_t)g.nblk_m * (uint32_t)g.nblk_n;
  const uint32_t widx = (G & 7u) == 0 ? (bid & 7u) * (G >> 3) + (bid >> 3) : bid;
  const uint32_t tq = T / G, trem = T - tq * G;
  const uint32_t t0 = widx * tq + (widx < trem ? widx : trem);
  int c_tiles = (int)(tq + (widx < trem ? 1u : 0u));
  if (c_tiles == 0) return;
  int c_bm = (int)(t0 / (uint32_t)g.nblk_n), c_bn = (int)(t0 - (uint32_t)c_bm * (uint32_t)g.nblk_n);

  // ---- per-pair constants (the layer's own pair; the shortcut pair of a DUAL kernel) ----
  const float zpf = zp_in ? zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin = s_in[0];
  const uint32_t xorw_m = shift ? 0x80808080u : 0u;
  const int8_t* const padline_m = g_pad_table.b + ((zpi & 0xff) << 6);
  int zpi2 = 0;
  float sin2 = 0.0f;
  uint32_t xorw_s = 0u;
  const int8_t* padline_s = g_pad_table.b;
  if (DUAL) {
    const float zf2 = sg.zp_in ? sg.zp_in[0] : 0.0f;
    zpi2 = (int)__builtin_rintf(zf2);
    sin2 = sg.s_in[0];
    xorw_s = sg.shift ? 0x80808080u : 0u;
    padline_s = g_pad_table.b + ((zpi2 & 0xff) << 6);
  }
  const int nsteps_m = g.R * g.S * (g.C / BK);
  const int nsteps_s = DUAL ? sg.g.R * sg.g.S * (sg.g.C / BK) : 0;

  // ---- software count of the vector-memory LOADS this wave has issued (see the header) ----
  int issued = 0;
  int seq[NBUF];                       // `issued` right after the operand group of each ring slot
#pragma unroll
  for (int i = 0; i < NBUF; ++i) seq[i] = 0;

  // ---- the loader: walks (tile, pair, K step) PF steps ahead of the multiplies ----
  const int lrow = lane >> 2, pslot = lane & 3;          // DMA: 16 weight rows x 4 slots of 16 B per wave-instruction
  int b_seg[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) b_seg[i] = pslot ^ ((((i * 4 + wave) * 16 + lrow) >> 2) & 3);
  int l_bm = c_bm, l_bn = c_bn, l_seg = 0, l_left = 0, l_tiles = c_tiles;
  bool l_active = true;
  const int8_t* ap = nullptr;      // this lane's A bytes for the current tap and channel chunk (or the pad line)
  int a_inc = 0;                   // BK for a real pixel, 0 for a padded tap
  const int8_t* bp[BI];            // this lane's DMA source for the current step (KRSC: a step is BK bytes further)
  int a_n = 0, a_h0 = 0, a_w0 = 0;
  bool a_ok = false;
  int l_cc = 0, l_s = 0, l_r = 0, l_cch = 1, l_S = 1;

  auto retap = [&]() {
    const bool shortcut = DUAL && l_seg == 0;
    const int H = shortcut ? sg.g.H : g.H, W = shortcut ? sg.g.W : g.W, C = shortcut ? sg.g.C : g.C;
    const int dil = shortcut ? sg.g.dil : g.dil;
    const int8_t* xx = shortcut ? sg.x : x;
    const int h = a_h0 + l_r * dil, ww = a_w0 + l_s * dil;
    const bool in = a_ok && h >= 0 && h < H && ww >= 0 && ww < W;
    ap = in ? xx + (((int64_t)a_n * H + h) * W + ww) * C + hsel * 16 : (shortcut ? padline_s : padline_m);
    a_inc = in ? BK : 0;
  };
  auto loader_begin = [&]() {      // start of (l_bm, l_bn, l_seg)
    const bool shortcut = DUAL && l_seg == 0;
    const int64_t m = (int64_t)l_bm * BM + wave * 32 + l31;
    if (shortcut) {
      a_ok = m < sg.g.M;
      row_origin(sg.g, a_ok ? (uint32_t)m : 0u, a_n, a_h0, a_w0);
    } else {
      a_ok = m < g.M;
      row_origin(g, a_ok ? (uint32_t)m : 0u, a_n, a_h0, a_w0);
    }
    const int R = shortcut ? sg.g.R : g.R, S = shortcut ? sg.g.S : g.S, C = shortcut ? sg.g.C : g.C;
    const int64_t wrow = (int64_t)R * S * C;
    const int8_t* ww = shortcut ? sg.w : w;
#pragma unroll
    for (int i = 0; i < BI; ++i) bp[i] = ww + (int64_t)(l_bn * BN + (i * 4 + wave) * 16 + lrow) * wrow + b_seg[i] * 16;
    l_cc = l_s = l_r = 0;
    l_cch = C / BK;
    l_S = S;
    l_left = shortcut ? nsteps_s : nsteps_m;
    retap();
  };
  auto issue = [&](auto slot_c) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int i = 0; i < BI; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)bp[i], (lptr_t)(lds + SL * TILE_B + (i * 4 + wave) * 1024), 16, 0, 0);
    AQ<ATOP / 4 - 1 - (AQ_FRAG + SL * KS + 0)>::template load<0>(ap);
    AQ<ATOP / 4 - 1 - (AQ_FRAG + SL * KS + 1)>::template load<32>(ap);
    issued += GROUP;
    seq[SL] = issued;
  };
  auto post_issue = [&]() {        // advance the loader by one K step (one copy of this code, outside the slot switch)
#pragma unroll
    for (int i = 0; i < BI; ++i) bp[i] += BK;
    ap += a_inc;
    bool moved = false;
    if (++l_cc == l_cch) {
      l_cc = 0;
      if (++l_s == l_S) {
        l_s = 0;
        ++l_r;
      }
      moved = true;
    }
    if (--l_left == 0) {
      if (DUAL && l_seg == 0) {
        l_seg = 1;
        loader_begin();
      } else if (--l_tiles > 0) {
        l_seg = 0;
        if (++l_bn == g.nblk_n) {
          l_bn = 0;
          ++l_bm;
        }
        loader_begin();
      } else {
        l_active = false;
      }
    } else if (moved) {
      retap();
    }
  };

  // ---- fragment addresses of the B operand (same for every 32-column slab: (row >> 2) & 3 only depends on the lane) ----
  const int bfo0 = l31 * BK + (((0 + hsel) ^ ((l31 >> 2) & 3)) << 4);
  const int bfo1 = l31 * BK + (((2 + hsel) ^ ((l31 >> 2) & 3)) << 4);

  i32x16 acc[NT];
  float extra[DUAL ? NT : 1][16];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0;

  bool did_issue = false;
  uint32_t c_xorw = (DUAL ? xorw_s : xorw_m);
  auto step = [&](auto slot_c) {
    constexpr int U = decltype(slot_c)::value;
    wait_vm(issued - seq[U]);                      // this step's operands have landed; younger loads stay in flight
    __builtin_amdgcn_s_barrier();                  // everyone's bytes are in LDS; everyone left the previous multiply
    did_issue = l_active;
    if (l_active) issue(std::integral_constant<int, (U + PF) % NBUF>{});   // into the slot the previous multiply released
    const int8_t* base = lds + U * TILE_B;
    static_for<KS>([&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      const i32x4 t = AQ<ATOP / 4 - 1 - (AQ_FRAG + U * KS + ks)>::read_i();     // (volatile asm: stays behind the wait above)
      const i32x4 af = i32x4{(int)(t.x ^ c_xorw), (int)(t.y ^ c_xorw), (int)(t.z ^ c_xorw), (int)(t.w ^ c_xorw)};
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const i32x4 bf = *reinterpret_cast<const i32x4*>(base + j * (32 * BK) + (ks ? bfo1 : bfo0));
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[j], 0, 0, 0);
      }
    });
  };

  // ---- per-tile constants and the shortcut tile: requested one tile ahead, awaited by count ----
  int seq_p = 0;
  int seq_r[DUAL ? 1 : NH];
  const bool has_res = !DUAL && ep.residual != nullptr;
  const bool has_bias = bias != nullptr, has_bias2 = DUAL && sg.bias != nullptr;
  const EpiQuant eq(ep);
  float* const stg = reinterpret_cast<float*>(lds + RING) + wave * (32 * EP_LD);
  const int er = lane >> 4, ec = (lane & 15) * 4;
  const int64_t Mrows = g.M;
  auto issue_params = [&](int n0) {         // per-channel constants of a tile's columns (accumulator layout: lane = channel)
    static_for<NT>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const int col = n0 + j * 32 + l31;
      A1<ATOP - 1 - (A1_PAR + j)>::load(s_w + col);
      A1<ATOP - 1 - (A1_PAR + NT + j)>::load(wsum + col);
      if (has_bias) A1<ATOP - 1 - (A1_PAR + 2 * NT + j)>::load(bias + col);
      if (DUAL) {
        A1<ATOP - 1 - (A1_PAR + 3 * NT + j)>::load(sg.s_w + col);
        A1<ATOP - 1 - (A1_PAR + 4 * NT + j)>::load(sg.wsum + col);
        if (has_bias2) A1<ATOP - 1 - (A1_PAR + 5 * NT + j)>::load(sg.bias + col);
      }
    });
    issued += NT * ((DUAL ? 4 : 2) + (has_bias ? 1 : 0) + (has_bias2 ? 1 : 0));
    seq_p = issued;
  };
  auto issue_residual = [&](auto h_c, int64_t m0, int n0) {   // one 64-channel pass of this wave's fp32 shortcut rows
    constexpr int h = decltype(h_c)::value;
    static_for<8>([&](auto it_c) {
      constexpr int it = decltype(it_c)::value;
      int64_t row = m0 + wave * 32 + it * 4 + er;
      row = row < Mrows ? row : Mrows - 1;                     // rows beyond M: a valid address, value unused
      AQ<ATOP / 4 - 1 - (DUAL ? 0 : AQ_RES + h * 8 + it)>::load_nt(ep.residual + row * g.K + (n0 + h * 64 + ec));
    });
    issued += 8;
    seq_r[DUAL ? 0 : h] = issued;
  };

  // ---- prologue: the first tile's constants and shortcut, then the first PF operand groups ----
  issue_params(c_bn * BN);
  if (has_res) static_for<DUAL ? 0 : NH>([&](auto h_c) { issue_residual(h_c, (int64_t)c_bm * BM, c_bn * BN); });
  loader_begin();
  static_for<PF>([&](auto i) {
    if (l_active) {
      issue(i);
      post_issue();
    }
  });

  int phase = 0;
  for (; c_tiles > 0; --c_tiles) {
    const int64_t m0 = (int64_t)c_bm * BM;
    const int n0 = c_bn * BN;
    int nx_bm = c_bm, nx_bn = c_bn + 1;        // the next tile of this workgroup's run
    if (nx_bn == g.nblk_n) {
      nx_bn = 0;
      ++nx_bm;
    }
    const bool has_next = c_tiles > 1;

#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
      const bool shortcut = DUAL && seg == 0;
      const int nst = shortcut ? nsteps_s : nsteps_m;
      c_xorw = shortcut ? xorw_s : xorw_m;
      for (int st = 0; st < nst; ++st) {
        static_for<NBUF>([&](auto u) {
          if (phase == decltype(u)::value) step(u);
        });
        phase = phase == NBUF - 1 ? 0 : phase + 1;
        if (did_issue) post_issue();
      }
      if (DUAL && seg == 0) {
        // the shortcut pair is complete: dequantise its sum into registers, start the layer's own sum from zero
        wait_vm(issued - seq_p);
        static_for<NT>([&](auto j_c) {
          constexpr int j = decltype(j_c)::value;
          const float mult = sin2 * A1<ATOP - 1 - (A1_PAR + 3 * NT + j)>::read_f();
          const int corr = (sg.shift - zpi2) * A1<ATOP - 1 - (A1_PAR + 4 * NT + j)>::read_i();
          const float bv = has_bias2 ? A1<ATOP - 1 - (A1_PAR + 5 * NT + j)>::read_f() : 0.0f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            extra[DUAL ? j : 0][i] = dequant1(acc[j][i] + corr, mult, bv);
            acc[j][i] = 0;
          }
        });
      }
    }

    // ---- epilogue of this tile (the next tiles' operand groups are in flight meanwhile) ----
    wait_vm(issued - seq_p);
    float mult[NT], bv[NT];
    int corr[NT];
    static_for<NT>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      mult[j] = sin * A1<ATOP - 1 - (A1_PAR + j)>::read_f();
      corr[j] = (shift - zpi) * A1<ATOP - 1 - (A1_PAR + NT + j)>::read_i();
      bv[j] = has_bias ? A1<ATOP - 1 - (A1_PAR + 2 * NT + j)>::read_f() : 0.0f;
    });
    if (has_next) issue_params(nx_bn * BN);        // (the registers were just read)
    static_for<NH>([&](auto h_c) {
      constexpr int h = decltype(h_c)::value;
      static_for<2>([&](auto jj_c) {
        constexpr int jj = decltype(jj_c)::value;
        constexpr int j = h * 2 + jj;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (i & 3) + 8 * (i >> 2) + 4 * hsel;
          float v = dequant1(acc[j][i] + corr[j], mult[j], bv[j]);
          if (DUAL) v = v + extra[DUAL ? j : 0][i];
          stg[r * EP_LD + jj * 32 + l31] = v;
          acc[j][i] = 0;
        }
      });
      if (has_res) wait_vm(issued - seq_r[DUAL ? 0 : h]);
      // a wave only reads back what it wrote itself: no block barrier, just the LDS counter
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int col = n0 + h * 64 + ec;
      f32x4 res[8];
      if (has_res) {
        static_for<8>([&](auto it_c) {
          constexpr int it = decltype(it_c)::value;
          res[it] = AQ<ATOP / 4 - 1 - (DUAL ? 0 : AQ_RES + h * 8 + it)>::read_f();
        });
        if (has_next) issue_residual(h_c, (int64_t)nx_bm * BM, nx_bn * BN);   // one tile ahead, into the registers just read
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + er;
        const int64_t row = m0 + wave * 32 + r;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * EP_LD + ec);
        if (row < Mrows) {
          const int64_t at = row * g.K + col;
          if (has_res) v = f32x4{v.x + res[it].x, v.y + res[it].y, v.z + res[it].z, v.w + res[it].w};
          if (ep.relu) v = f32x4{relu_nan(v.x), relu_nan(v.y), relu_nan(v.z), relu_nan(v.w)};
          if (out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + at));
          if (ep.codes) __builtin_nontemporal_store(eq.code4(v), reinterpret_cast<uint32_t*>(ep.codes + at));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next pass overwrites the stage
    });

    c_bm = nx_bm;
    c_bn = nx_bn;
  }
}

}  // namespace dlmcq

using namespace dlmcq;

// Launch the persistent kernel (called by conv_launch in conv_i8.hip).  Requirements checked by the caller:
// C % 64 == 0, K % bn == 0, M < 2^31; g.nblk_m / g.nblk_n are filled in here.
int dlmcq_conv_pp_launch(const int8_t* x, const int8_t* w, float* out, const float* bias, const int32_t* wsum,
                         const float* in_scale, const float* in_zero_point, const float* w_scale, ConvGeom g, int shift,
                         const ConvEpi& ep, const ConvSeg2* seg2, int bn, int nbuf, int wps, hipStream_t st) {
  g.nblk_m = (int)((g.M + 127) / 128);
  g.nblk_n = g.K / bn;
  const int64_t tiles = (int64_t)g.nblk_m * g.nblk_n;
  if (tiles >= (1ll << 31)) return DLMCQ_ERANGE;
  const int64_t slots = (int64_t)DLMCQ_CUS * wps;
  const int64_t grid = tiles < slots ? tiles : slots;
  ConvSeg2 s2{};
  if (seg2) {
    s2 = *seg2;
    s2.g.nblk_m = g.nblk_m;
    s2.g.nblk_n = g.nblk_n;
  }
#define DLMCQ_PP_ARGS dim3((uint32_t)grid), dim3(256), 0, st, x, w, out, bias, wsum, in_scale, in_zero_point, w_scale, g, shift, ep, s2
#define DLMCQ_PP_GO(BN_, DUAL_)                                                                        \
  do {                                                                                                 \
    if (wps == 2) hipLaunchKernelGGL((conv_i8_pp_kernel<BN_, DUAL_, 4, 2>), DLMCQ_PP_ARGS);            \
    else if (nbuf <= 4) hipLaunchKernelGGL((conv_i8_pp_kernel<BN_, DUAL_, 4, 1>), DLMCQ_PP_ARGS);      \
    else hipLaunchKernelGGL((conv_i8_pp_kernel<BN_, DUAL_, 8, 1>), DLMCQ_PP_ARGS);                     \
  } while (0)
  if (wps == 2 && bn != 64) return DLMCQ_EINVAL;
  if (seg2) {
    if (bn == 64) DLMCQ_PP_GO(64, true);
    else if (bn == 128) hipLaunchKernelGGL((conv_i8_pp_kernel<128, true, 8, 1>), DLMCQ_PP_ARGS);
    else return DLMCQ_EINVAL;
  } else {
    if (bn == 64) DLMCQ_PP_GO(64, false);
    else if (bn == 128) hipLaunchKernelGGL((conv_i8_pp_kernel<128, false, 8, 1>), DLMCQ_PP_ARGS);
    else return DLMCQ_EINVAL;
  }
#undef DLMCQ_PP_GO
#undef DLMCQ_PP_ARGS
  return launch_status();
}
