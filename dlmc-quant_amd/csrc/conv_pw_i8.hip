// Pointwise (1 x 1, stride 1) convolution from activation codes to the consumer's codes with the weights RESIDENT in LDS
// (modules/conv.py:13-19 on FSPTQuant/base.py:108-109 / ops.py:129-136 operands; MobileOne-S1's 21 pointwise layers, BASELINE
// configs[4]).
//
// Why a kernel of its own (tools/pw_lab.py, LABNOTES 12): on these layers the reduction is 1-8 K steps long and the tiled kernel of
// conv_i8.hip spends its time in three phases that ADD UP - a tile's operand round trip (weights re-streamed per tile: 37 KB of
// weights for 25 KB of activations at 192 -> 192), its quantising epilogue (25-50 % of the kernel: a wave64 vector instruction
// occupies its SIMD for 4 clocks whatever it is) and its staged stores -, with a barrier per K step keeping the four waves of a
// workgroup in the same phase.  Here
//   * a workgroup loads its slice of the weights (BN output channels x C bytes, LDS-DMA, conv_i8.hip's swizzled 64-byte rows)
//     and the per-channel constants ONCE and then never synchronises again: no ring, no barrier in the loop;
//   * every wave walks blocks of 32 pixels on its own: activation fragments straight to registers (buffer loads, the NEXT
//     block's requested in the middle of this block's epilogue behind a counted wait, so the epilogue runs in the shadow of the
//     loads), C / 32 x BN / 32 MFMAs with the weights as the A operand (swapped layout: a lane owns 16 consecutive output channels
//     of one pixel) in passes of NTP accumulator blocks, the swapped epilogue of conv_i8.hip on channel pairs with the plain
//     quantiser (conv_epilogue.h: code4n_plain; the launcher checks epi_plain), the codes through a wave-private LDS stage so
//     that the stores are whole rows (32-byte pieces straight from the accumulator layout cost as much as all the rest);
//   * waves drift apart freely, so one wave's vector arithmetic overlaps another's matrix and memory work.
// Same integers, same fp32 chain, the same bytes as conv_i8_mfma_kernel (tests/test_gpu_pointwise.py).  tools/lint_pw.py checks
// the listing: no scratch, nothing touches a fragment register between its load and the counted wait (control-flow walk).
#include "conv_i8_common.h"

namespace dlmcq {

struct PwArgs {
  const int8_t* x;         // [M][C] codes
  const int8_t* w;         // [K][C] int8
  const float* s_w;        // [K]
  const int32_t* wsum;     // [K] SUM qw
  const float* bias;       // [K] or null
  const float* s_in;
  const float* zp_in;      // null: 0
  int M, K, shift, nslice; // nslice = K / BN column slices; workgroup b: slice (b >> 3) % nslice, group ((b >> 3) / nslice) * 8 + (b & 7)
  int nblk;                // 32-pixel blocks
  unsigned long long* trace;   // lab builds: clock stamps of one wave (tools/pw_lab.py --trace); null otherwise
};

// waves per SIMD the kernel is compiled for (registers: fragments C / 8 + accumulators of one pass + ~70 for the epilogue)
constexpr int PW_WPS(int c, int bn) { return c >= 1024 ? 1 : (c >= 512 ? 3 : 4); }

// LAB (lab library only, what-bounds-the-block experiments: results are garbage): 1 = clock stamps of one wave, 2 = no activation
// loads, 4 = no MFMAs, 6 = epilogue without its arithmetic, 7 = no epilogue and no stores, 8 = no stores, 9 = non-temporal stores
template <int C, int BN, int NTP, bool ASYM, int NW, int LAB = 0>
__global__ __launch_bounds__(NW * 64, PW_WPS(C, BN)) void conv_pw_i8_kernel(PwArgs a, ConvEpi ep) {
  constexpr int S = C / 64;             // 64-byte K steps
  constexpr int NA = C / 32;            // A fragments (16 bytes per lane each)
  constexpr int NT = BN / 32;           // 32-channel accumulator blocks
  constexpr int WB = C * BN;            // weight bytes of the slice
  constexpr int PIECES = S * (BN / 16); // 1 KB DMA pieces of the slice
  constexpr int NPAR = ASYM ? 4 : 3;
  constexpr int NP = NT / NTP;          // passes over the slice's accumulator blocks
  constexpr int JP = NTP > 1 ? 1 : 0;   // accumulator blocks of the last pass finished before the next block's fragments are requested
  static_assert(NT % NTP == 0 && JP < NTP && NTP >= 1, "the request sits inside the last pass's epilogue");
  extern __shared__ __attribute__((aligned(1024))) int8_t pw_lds[];
  int8_t* const wl = pw_lds;                    // [S][BN rows][64 B], LDS slot p of row r = logical segment p ^ ((r >> 2) & 3)
  int8_t* const par = pw_lds + WB;              // s_in s_w | (shift - zp) SUM qw | bias | s_in o_w
  constexpr int SROW = BN + 16;                 // a staged row of codes (+ 16: conflict-free 16-byte accesses)
  constexpr int LPR = BN / 16, RPI = 64 / LPR;  // lanes per staged row, whole rows per wave-instruction (192 channels: 5, four lanes idle)
  constexpr int NST = (32 + RPI - 1) / RPI;     // store instructions per block
  int8_t* const stg = pw_lds + WB + 4 * BN * 4 + (threadIdx.x >> 6) * (32 * SROW);   // this wave's 32 rows
  const int srow = (int)((threadIdx.x & 63) / LPR), sseg = (int)(threadIdx.x & 63) - srow * LPR;   // this lane's place in a row-major read of the stage

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const uint32_t b = blockIdx.x;
  const int slice = (int)((b >> 3) % (uint32_t)a.nslice);
  const int group = (int)((b >> 3) / (uint32_t)a.nslice) * 8 + (int)(b & 7u);
  const int ngroups = (int)(gridDim.x / (uint32_t)a.nslice);
  const int n0 = slice * BN;

  // lab builds: clock stamps of wave 0 of the middle workgroup: [0] real-time clock (100 MHz) at entry, [1] at exit, [2] stamps written,
  // [3...] shader clocks: entry, loop entered, then per block: loop top, fragments landed, (K loop, epilogue) per pass, stores issued
  unsigned long long* tr = nullptr;
  if (LAB == 1 && a.trace && blockIdx.x == gridDim.x / 2 && tid == 0) tr = a.trace;
  int nstamp = 3;
  auto stamp = [&]() {
    if (LAB == 1 && tr && nstamp < 256) tr[nstamp++] = __builtin_readcyclecounter();
  };
  if (LAB == 1 && tr) tr[0] = __builtin_amdgcn_s_memrealtime();
  stamp();
  unsigned long long* wgt = nullptr;     // ... and per workgroup behind those 256 slots: real-time clock at entry / exit, HW_ID, XCC_ID
  if (LAB == 1 && a.trace && tid == 0) {
    wgt = a.trace + 256 + 4 * (size_t)blockIdx.x;
    wgt[0] = __builtin_amdgcn_s_memrealtime();
    wgt[2] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    wgt[3] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  }
  // ---- once per workgroup: constants and weights by LDS-DMA ----
  {
    const void* arrs[4] = {a.s_w, a.wsum, a.bias, ASYM ? ep.w_off : nullptr};
#pragma unroll
    for (int r = 0; r < NPAR; ++r) {
      if (!arrs[r]) {                                  // (no bias: zeros, so that the epilogue reads a table either way)
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
  }
  const float zpf = a.zp_in ? a.zp_in[0] : 0.0f;
  const int zpi = (int)__builtin_rintf(zpf);
  const float sin = a.s_in[0];
  const uint32_t xorw = a.shift ? 0x80808080u : 0u;
  const EpiQuant eq(ep, ep.relu != 0);            // code(relu(v)) = max(code(v), code(0))
  const v4i r_x = make_rsrc(a.x, (uint32_t)((int64_t)a.M * C));
  const v4i r_c = make_rsrc(ep.codes, (uint32_t)((int64_t)a.M * a.K));

  i32x4 areg[NA];
  auto request = [&](int blk) {       // this lane's fragments of block `blk`: bytes 32 f + 16 hsel .. + 15 of pixel 32 blk + l31
    const int row = blk * 32 + l31;
    const int vo = (blk < a.nblk && row < a.M) ? row * C + hsel * 16 : BUF_BIG;
    if constexpr (LAB == 2) {
#pragma unroll
      for (int f = 0; f < NA; ++f) areg[f] = i32x4{vo, 1, 2, 3};
      return;
    }
    static_for<NA>([&](auto f) { bload16i<decltype(f)::value * 32>(areg[decltype(f)::value], vo, r_x); });
  };
  const int stride = ngroups * NW;
  int blk = group * NW + wave;
  request(blk);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid < BN) {      // (s_w, SUM qw) -> the epilogue's (s_in * s_w, (shift - zp) * SUM qw), once per channel
    float* pf = reinterpret_cast<float*>(par) + tid;
    int* pi = reinterpret_cast<int*>(par) + BN + tid;
    *pf = sin * *pf;
    *pi = (a.shift - zpi) * *pi;
    if constexpr (ASYM) pf[3 * BN] = sin * pf[3 * BN];
  }
  __syncthreads();

  stamp();
  for (; blk < a.nblk; blk += stride) {
    // this block's fragments have landed (the first block's: the prologue waited for everything); the previous block's NST stores
    // may still be in flight
    stamp();
    if constexpr (LAB == 2 || LAB == 7 || LAB == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    stamp();
#pragma unroll
    for (int f = 0; f < NA; ++f) {
      asm volatile("" : "+v"(areg[f]));     // the asm-loaded fragments are valid from here on
      areg[f] = i32x4{(int)(areg[f].x ^ xorw), (int)(areg[f].y ^ xorw), (int)(areg[f].z ^ xorw), (int)(areg[f].w ^ xorw)};   // uint8 -> int8, once for all passes
    }
    f32x2 s0f2 = f32x2{0.0f, 0.0f};
    // NP passes over NTP accumulator blocks each (the fragments stay in registers; what a pass holds besides them is NTP x 16
    // accumulators and the epilogue's ~70 registers)
    static_for<NP>([&](auto p_c) {
      constexpr int p = decltype(p_c)::value;
      // the accumulators start from (shift - zp) * SUM qw of their channels (register i of block j: channel 32 j + 16 hsel + i):
      // four LDS reads instead of sixteen additions per block
      i32x16 acc[NTP];
#pragma unroll
      for (int j = 0; j < NTP; ++j) {
        const i32x4* cop = reinterpret_cast<const i32x4*>(par + (BN + (p * NTP + j) * 32 + hsel * 16) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const i32x4 c4 = cop[q];
          acc[j][4 * q] = c4.x;
          acc[j][4 * q + 1] = c4.y;
          acc[j][4 * q + 2] = c4.z;
          acc[j][4 * q + 3] = c4.w;
        }
      }
      int s0 = 0;
      static_for<NA>([&](auto f_c) {
        constexpr int f = decltype(f_c)::value;
        constexpr int s = f >> 1, ks = f & 1;
        const i32x4 af = areg[f];
        if constexpr (ASYM && p == 0) {
          s0 = __builtin_amdgcn_sdot4(af.x, 0x01010101, s0, false);
          s0 = __builtin_amdgcn_sdot4(af.y, 0x01010101, s0, false);
          s0 = __builtin_amdgcn_sdot4(af.z, 0x01010101, s0, false);
          s0 = __builtin_amdgcn_sdot4(af.w, 0x01010101, s0, false);
        }
        const int sg = ks * 2 + hsel;
#pragma unroll
        for (int j = 0; j < NTP; ++j) {
          const int brow = (p * NTP + j) * 32 + l31;
          const i32x4 bf = *reinterpret_cast<const i32x4*>(wl + s * (BN * 64) + brow * 64 + ((sg ^ ((brow >> 2) & 3)) << 4));
          if constexpr (LAB == 4) asm volatile("" ::"v"(af), "v"(bf));
          else acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, af, acc[j], 0, 0, 0);
        }
      });
      if constexpr (LAB == 1) {
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[NTP - 1][15]));
        stamp();
      }
      // ---- the swapped epilogue (conv_i8.hip): register i of block j = channel n0 + 32 j + 16 hsel + i of pixel `row` ----
      if constexpr (ASYM && p == 0) {
        s0 += __shfl_xor(s0, 32, 64);
        s0 += (a.shift - zpi) * C;
        s0f2 = f32x2{(float)s0, (float)s0};
      }
#pragma unroll
      for (int jj = 0; jj < NTP; ++jj) {
        // the next block's fragments, requested in the last pass once JP of its accumulator blocks are finished and early enough
        // for the remaining blocks' arithmetic to cover the round trip.  (Past the end: out-of-range offsets, the loads return
        // zeros and are never used.)
        if (p == NP - 1 && jj == JP) request(blk + stride);
        const int j = p * NTP + jj;
        const int cb = j * 32 + hsel * 16;
        if constexpr (LAB == 6 || LAB == 7) {
          uint32_t wl4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            wl4[q] = (uint32_t)(acc[jj][4 * q] & 0xff) | ((uint32_t)(acc[jj][4 * q + 1] & 0xff) << 8) | ((uint32_t)(acc[jj][4 * q + 2] & 0xff) << 16) |
                     ((uint32_t)acc[jj][4 * q + 3] << 24);
          if constexpr (LAB == 6) *reinterpret_cast<i32x4*>(stg + l31 * SROW + cb) = i32x4{(int)wl4[0], (int)wl4[1], (int)wl4[2], (int)wl4[3]};
          else asm volatile("" ::"v"(wl4[0]), "v"(wl4[1]), "v"(wl4[2]), "v"(wl4[3]));
          continue;
        }
        // on pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the same roundings as conv_i8.hip's scalar chain, half the instructions)
        f32x4 y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (cb + 4 * q) * 4);
          const f32x4 bs = *reinterpret_cast<const f32x4*>(par + (2 * BN + cb + 4 * q) * 4);
          f32x2 ya = pk_fma(f32x2{(float)acc[jj][4 * q], (float)acc[jj][4 * q + 1]}, f32x2{mu.x, mu.y}, f32x2{bs.x, bs.y});
          f32x2 yb = pk_fma(f32x2{(float)acc[jj][4 * q + 2], (float)acc[jj][4 * q + 3]}, f32x2{mu.z, mu.w}, f32x2{bs.z, bs.w});
          if constexpr (ASYM) {
            const f32x4 wo = *reinterpret_cast<const f32x4*>(par + (3 * BN + cb + 4 * q) * 4);
            ya = ya + s0f2 * f32x2{wo.x, wo.y};
            yb = yb + s0f2 * f32x2{wo.z, wo.w};
          }
          y[q] = f32x4{ya.x, ya.y, yb.x, yb.y};
        }
        uint32_t wq[4];
        eq.code4n_plain(y, wq);
        // the block's 16 bytes of this pixel wait in the wave's LDS stage: the stores below leave as whole rows (a store straight
        // from here writes 32 contiguous bytes per pixel - 32 partial lines per instruction - and the layer spends half its
        // time in them: tools/pw_lab.py, 192 -> 192 at 28^2 95 us with, 43 without its stores and arithmetic, 91 with the stores alone)
        *reinterpret_cast<i32x4*>(stg + l31 * SROW + cb) = i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
      }
      stamp();
    });
    if constexpr (LAB != 7) {
      // row-major out of the stage: LPR lanes x 16 bytes per row (LDS operations of one wave execute in order: no barrier)
      // (one per-lane byte offset for the block, the per-instruction steps are scalars: nothing per instruction lives in a register)
      const int r0 = blk * 32 + srow;
      const int vbase = r0 * a.K + n0 + sseg * 16;
      const int8_t* const sbase = stg + srow * SROW + sseg * 16;
#pragma unroll
      for (int it = 0; it < NST; ++it) {
        const bool ok = srow < RPI && it * RPI + srow < 32 && r0 + it * RPI < a.M;
        // (the last instruction may reach past row 31 - into the next wave's stage or past the allocation: its lanes read row 31 instead and store nothing)
        const bool whole = (it + 1) * RPI <= 32;
        const i32x4 c16 = *reinterpret_cast<const i32x4*>(whole || it * RPI + srow < 32 ? sbase + it * RPI * SROW : stg + 31 * SROW + sseg * 16);
        if constexpr (LAB == 8) asm volatile("" ::"v"(c16));
        else if constexpr (LAB == 9) bstore16i_nt(c16, ok ? vbase + it * RPI * a.K : BUF_BIG, r_c);
        else bstore16i(c16, ok ? vbase + it * RPI * a.K : BUF_BIG, r_c);
      }
    }
    stamp();
  }
  if (LAB == 1 && tr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();
    tr[1] = __builtin_amdgcn_s_memrealtime();
    tr[2] = (unsigned long long)nstamp;
  }
  if (LAB == 1 && wgt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wgt[1] = __builtin_amdgcn_s_memrealtime();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the requests past the end: nothing may be in flight into registers at s_endpgm)
}

// The (input channels, slice width) pairs the kernel is instantiated for - ONE table for conv_pw_applies and conv_pw_launch (round 4's
// predicate admitted e.g. 128 -> 64 or 512 -> 192, which the launch table did not hold: conv_launch then returned DLMCQ_EINVAL for a
// layer the tiled kernel used to take).  A K-wide layer is cut into slices of pw_slice(K) output channels.
#define DLMCQ_PW_PAIRS(X) X(64, 64) X(64, 128) X(64, 192) X(128, 128) X(128, 192) X(192, 128) X(192, 192) X(512, 128) X(1024, 128)
constexpr int pw_slice(int64_t K) { return K == 192 ? 192 : (K == 64 ? 64 : (K % 128 == 0 ? 128 : 0)); }
constexpr bool pw_built(int64_t C, int bn) {
#define DLMCQ_PW_HAS(CC, BB) if (C == CC && bn == BB) return true;
  DLMCQ_PW_PAIRS(DLMCQ_PW_HAS)
#undef DLMCQ_PW_HAS
  return false;
}

// where the kernel applies: codes in, codes out, nothing else attached; the widths MobileOne-S1 uses
bool conv_pw_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                     int32_t dilation, const ConvEpi& ep, const float* out, bool dual) {
  if (R != 1 || S != 1 || stride != 1 || pad != 0 || dilation != 1 || dual || out || ep.residual || !ep.codes) return false;
  if (!epi_plain(ep)) return false;                             // (other quantisers: the tiled kernel)
  if (K > 1024 || !pw_built(C, pw_slice(K))) return false;       // (exactly the (C, slice) pairs conv_pw_launch instantiates)
  if (!aligned16(ep.codes)) return false;
  const int64_t M = N * H * W;
  if (M < 4096) return false;                                   // (the weights are loaded once per workgroup: a few blocks per wave at least)
  if (M * C >= (int64_t)BUF_BIG || M * K >= (int64_t)BUF_BIG) return false;     // 32-bit buffer offsets
  return true;
}

template <int C, int BN, bool ASYM, int LAB = 0>
static int pw_go(const PwArgs& a0, const ConvEpi& ep, hipStream_t st) {
  constexpr int WPS = PW_WPS(C, BN);
  constexpr int LDS1 = C * BN + 4 * BN * 4, STG = 4 * 32 * (BN + 16);   // weights + constants; one 4-wave workgroup's code stages
  // workgroups of 4 waves where WPS of them fit a CU's LDS, else ONE workgroup of 4 WPS waves per CU (one copy of the weights)
  // (1 024 input channels: the 128 KB slice leaves room for six waves' stages)
  constexpr bool BIGWG = C >= 1024 || (LDS1 + STG) * WPS > 158 * 1024;
  constexpr int NW = C >= 1024 ? 6 : (BIGWG ? 4 * WPS : 4);
  constexpr int LDS = LDS1 + NW * (STG / 4);
  static_assert(LDS <= 160 * 1024, "LDS");
  PwArgs a = a0;
  a.nslice = a.K / BN;
  static int cus = 0;      // (one device per process: dlmc/_native.py)
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  const int wgs = cus * (BIGWG ? 1 : WPS);
  int ngroups = (wgs / a.nslice) & ~7;                            // groups of `nslice` workgroups on one XCD share their pixels
  if (ngroups < 8) ngroups = 8;
  const int maxg = ((a.nblk + NW - 1) / NW + 7) & ~7;
  if (ngroups > maxg) ngroups = maxg;
  constexpr int NTP = BN == 192 ? 3 : (C >= 512 ? 1 : 2);      // (64-wide slices: one pass of two blocks)
  auto kern = conv_pw_i8_kernel<C, BN, NTP, ASYM, NW, LAB>;
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((uint32_t)(ngroups * a.nslice)), dim3(NW * 64), LDS, st, a, ep);
  return launch_status();
}

int conv_pw_launch(const int8_t* x, const int8_t* w, const float* bias, const int32_t* wsum, const float* in_scale,
                   const float* in_zero_point, const float* w_scale, int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int shift,
                   const ConvEpi& ep, hipStream_t st, int lab, void* lab_trace) {
  PwArgs a{};
  a.trace = static_cast<unsigned long long*>(lab_trace);
  a.x = x; a.w = w; a.s_w = w_scale; a.wsum = wsum; a.bias = bias; a.s_in = in_scale; a.zp_in = in_zero_point;
  a.M = (int)(N * H * W); a.K = (int)K; a.shift = shift;
  a.nblk = (a.M + 31) / 32;
  const bool asym = ep.w_off != nullptr;
  const int bn = pw_slice(K);
#ifdef DLMCQ_LAB
#define DLMCQ_PWL(CC, BB, L) \
  if (C == CC && bn == BB && lab == L && asym) return pw_go<CC, BB, true, L>(a, ep, st)
#define DLMCQ_PWLS(CC, BB) DLMCQ_PWL(CC, BB, 1); DLMCQ_PWL(CC, BB, 2); DLMCQ_PWL(CC, BB, 4); DLMCQ_PWL(CC, BB, 6); DLMCQ_PWL(CC, BB, 7); DLMCQ_PWL(CC, BB, 8); DLMCQ_PWL(CC, BB, 9)
  DLMCQ_PWLS(128, 128);
  DLMCQ_PWLS(192, 192);
  DLMCQ_PWLS(192, 128);
  DLMCQ_PWLS(512, 128);
#undef DLMCQ_PWLS
#undef DLMCQ_PWL
#endif
  if (lab) return DLMCQ_EINVAL;
#define DLMCQ_PW(CC, BB)                                         \
  if (C == CC && bn == BB) return asym ? pw_go<CC, BB, true>(a, ep, st) : pw_go<CC, BB, false>(a, ep, st);
  DLMCQ_PW_PAIRS(DLMCQ_PW)
#undef DLMCQ_PW
  return DLMCQ_EINVAL;
}

}  // namespace dlmcq
