// Pointwise (1 x 1, stride 1) convolution from activation codes to the consumer's codes with the weights RESIDENT in LDS
// (modules/conv.py:13-19 on FSPTQuant/base.py:108-109 / ops.py:129-136 operands; MobileOne-S1's 21 pointwise layers, BASELINE
// configs[4]).
//
// Why a kernel of its own (tools/pw_lab.py, round 4): on these layers the reduction is 1-8 K steps long and the tiled kernel of
// conv_i8.hip spends its time in three phases that ADD UP - a tile's operand round trip (weights re-streamed per tile: 37 KB of
// weights for 25 KB of activations at 192 -> 192), its quantising epilogue (25-50 % of the kernel, one wave issues a vector
// instruction every ~5 clocks, a SIMD could take one every ~2.5 from several waves) and its staged stores -, with a barrier per
// K step keeping the four waves of a workgroup in the same phase.  Here
//   * a workgroup loads its slice of the weights (BN output channels x C bytes, LDS-DMA, conv_i8.hip's swizzled 64-byte rows)
//     and the per-channel constants ONCE and then never synchronises again: no ring, no barrier in the loop;
//   * every wave walks blocks of 32 pixels on its own: activation fragments straight to registers (buffer loads, the NEXT
//     block's requested before this block's epilogue, so the epilogue runs in the shadow of the loads), C / 32 x BN / 32 MFMAs
//     with the weights as the A operand (swapped layout: a lane owns 16 consecutive output channels of one pixel), the swapped
//     epilogue of conv_i8.hip operation for operation, 16-byte stores straight from the registers;
//   * waves drift apart freely, so one wave's vector arithmetic overlaps another's matrix and memory work.
// Same integers, same fp32 chain, same quantiser as conv_i8_mfma_kernel: bit-identical (tests/test_gpu_pointwise.py).
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
};

// registers decide the waves per SIMD: A fragments C / 8 + accumulators BN / 2 + ~45 for the epilogue
constexpr int PW_WPS(int c, int bn) { return 3; }

template <int C, int BN, int NTP, bool ASYM, int NW>
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

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hsel = lane >> 5;
  const uint32_t b = blockIdx.x;
  const int slice = (int)((b >> 3) % (uint32_t)a.nslice);
  const int group = (int)((b >> 3) / (uint32_t)a.nslice) * 8 + (int)(b & 7u);
  const int ngroups = (int)(gridDim.x / (uint32_t)a.nslice);
  const int n0 = slice * BN;

  // ---- once per workgroup: constants and weights by LDS-DMA ----
  {
    const void* arrs[4] = {a.s_w, a.wsum, a.bias, ASYM ? ep.w_off : nullptr};
#pragma unroll
    for (int r = 0; r < NPAR; ++r) {
      if (!arrs[r]) continue;
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

  bool first = true;
  for (; blk < a.nblk; blk += stride) {
    // this block's fragments have landed; the previous block's last NTP - JP stores may still be in flight (first block: nothing younger)
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NTP - JP) : "memory");
    first = false;
#pragma unroll
    for (int f = 0; f < NA; ++f) asm volatile("" : "+v"(areg[f]));     // the asm-loaded fragments are valid from here on
    const int row = blk * 32 + l31;
    const int so = row < a.M ? row * a.K + n0 + hsel * 16 : BUF_BIG;
    float s0f = 0.0f;
    // NP passes over NTP accumulator blocks each (the fragments stay in registers; what a pass holds besides them is NTP x 16
    // accumulators and the epilogue's ~70 registers)
    static_for<NP>([&](auto p_c) {
      constexpr int p = decltype(p_c)::value;
      i32x16 acc[NTP];
#pragma unroll
      for (int j = 0; j < NTP; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0;
      int s0 = 0;
      static_for<NA>([&](auto f_c) {
        constexpr int f = decltype(f_c)::value;
        constexpr int s = f >> 1, ks = f & 1;
        const i32x4 t = areg[f];
        const i32x4 af = i32x4{(int)(t.x ^ xorw), (int)(t.y ^ xorw), (int)(t.z ^ xorw), (int)(t.w ^ xorw)};
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
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf, af, acc[j], 0, 0, 0);
        }
      });
      // ---- the swapped epilogue (conv_i8.hip): register i of block j = channel n0 + 32 j + 16 hsel + i of pixel `row` ----
      if constexpr (ASYM && p == 0) {
        s0 += __shfl_xor(s0, 32, 64);
        s0 += (a.shift - zpi) * C;
        s0f = (float)s0;
      }
#pragma unroll
      for (int jj = 0; jj < NTP; ++jj) {
        // the next block's fragments, requested in the last pass once JP of its accumulator blocks are finished and early enough
        // for the remaining blocks' arithmetic to cover the round trip.  (Past the end: out-of-range offsets, the loads return
        // zeros and are never used.)
        if (p == NP - 1 && jj == JP) request(blk + stride);
        const int j = p * NTP + jj;
        const int cb = j * 32 + hsel * 16;
        f32x4 y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 mu = *reinterpret_cast<const f32x4*>(par + (cb + 4 * q) * 4);
          const i32x4 co = *reinterpret_cast<const i32x4*>(par + (BN + cb + 4 * q) * 4);
          const f32x4 bs = a.bias ? *reinterpret_cast<const f32x4*>(par + (2 * BN + cb + 4 * q) * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          y[q] = f32x4{dequant1(acc[jj][4 * q] + co.x, mu.x, bs.x), dequant1(acc[jj][4 * q + 1] + co.y, mu.y, bs.y),
                       dequant1(acc[jj][4 * q + 2] + co.z, mu.z, bs.z), dequant1(acc[jj][4 * q + 3] + co.w, mu.w, bs.w)};
          if constexpr (ASYM) {
            const f32x4 wo = *reinterpret_cast<const f32x4*>(par + (3 * BN + cb + 4 * q) * 4);
            y[q] = f32x4{y[q].x + s0f * wo.x, y[q].y + s0f * wo.y, y[q].z + s0f * wo.z, y[q].w + s0f * wo.w};
          }
        }
        uint32_t wq[4];
        bool uq[4];
        eq.code4n(y, wq, uq);
        bstore16i(i32x4{(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]}, so == BUF_BIG ? BUF_BIG : so + j * 32, r_c);
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the requests past the end: nothing may be in flight into registers at s_endpgm)
}

// where the kernel applies: codes in, codes out, nothing else attached; the widths MobileOne-S1 uses
bool conv_pw_applies(int64_t N, int64_t H, int64_t W, int64_t C, int64_t K, int64_t R, int64_t S, int32_t stride, int32_t pad,
                     int32_t dilation, const ConvEpi& ep, const float* out, bool dual) {
  if (R != 1 || S != 1 || stride != 1 || pad != 0 || dilation != 1 || dual || out || ep.residual || !ep.codes) return false;
  if (!(C == 64 || C == 128 || C == 192 || C == 512)) return false;
  if (!(K == 192 || K % 128 == 0) || K > 1024) return false;
  if (!aligned16(ep.codes)) return false;
  const int64_t M = N * H * W;
  if (M < 4096) return false;                                   // (the weights are loaded once per workgroup: a few blocks per wave at least)
  if (M * C >= (int64_t)BUF_BIG || M * K >= (int64_t)BUF_BIG) return false;     // 32-bit buffer offsets
  return true;
}

template <int C, int BN, bool ASYM>
static int pw_go(const PwArgs& a0, const ConvEpi& ep, hipStream_t st) {
  constexpr int WPS = PW_WPS(C, BN);
  constexpr int LDS = C * BN + 4 * BN * 4;
  // workgroups of 4 waves where WPS of them fit a CU's LDS, else ONE workgroup of 4 WPS waves per CU (one copy of the weights)
  constexpr bool BIGWG = LDS * WPS > 150 * 1024;
  constexpr int NW = BIGWG ? 4 * WPS : 4;
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
  constexpr int NTP = BN == 192 ? 3 : (C >= 512 ? 1 : BN / 32);
  auto kern = conv_pw_i8_kernel<C, BN, NTP, ASYM, NW>;
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
                   const ConvEpi& ep, hipStream_t st) {
  PwArgs a{};
  a.x = x; a.w = w; a.s_w = w_scale; a.wsum = wsum; a.bias = bias; a.s_in = in_scale; a.zp_in = in_zero_point;
  a.M = (int)(N * H * W); a.K = (int)K; a.shift = shift;
  a.nblk = (a.M + 31) / 32;
  const bool asym = ep.w_off != nullptr;
  const int bn = K == 192 ? 192 : 128;
#define DLMCQ_PW(CC, BB)                                         \
  if (C == CC && bn == BB) return asym ? pw_go<CC, BB, true>(a, ep, st) : pw_go<CC, BB, false>(a, ep, st)
  DLMCQ_PW(64, 128);
  DLMCQ_PW(64, 192);
  DLMCQ_PW(128, 128);
  DLMCQ_PW(128, 192);
  DLMCQ_PW(192, 128);
  DLMCQ_PW(192, 192);
  DLMCQ_PW(512, 128);
#undef DLMCQ_PW
  return DLMCQ_EINVAL;
}

}  // namespace dlmcq
