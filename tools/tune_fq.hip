// Stand-alone geometry / cache-policy sweep for the streaming fake-quant kernel (per-tensor ZEROPOINT form).
// Build:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/tune_fq.hip -o /tmp/tune_fq
// Not part of the product; it exists to choose U / grid / nt flags from measurements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float clampn(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float ster(float v) { float r = __builtin_rintf(v); return (r - v) + v; }
__device__ __forceinline__ float fq1(float x, float s, float zp, float lo, float hi) {
  float q = clampn(ster(x / s) + zp, lo, hi);
  return (q - zp) * s;
}
__device__ __forceinline__ f32x4 fq4(f32x4 v, float s, float zp, float lo, float hi) {
  return f32x4{fq1(v.x, s, zp, lo, hi), fq1(v.y, s, zp, lo, hi), fq1(v.z, s, zp, lo, hi), fq1(v.w, s, zp, lo, hi)};
}

template <bool NT> __device__ __forceinline__ f32x4 ld(const f32x4* p) {
  if (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f32x4* p, f32x4 v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0: compute; MODE 1: pure copy (ceiling)
template <int BLOCK, int U, bool NTL, bool NTS, int MODE>
__global__ __launch_bounds__(BLOCK) void k(const float* x, float* y, const float* sc, long n4) {
  const float s = sc[0], zp = sc[1];
  const f32x4* x4 = (const f32x4*)x;
  f32x4* y4 = (f32x4*)y;
  const long nchunks = (n4 + BLOCK * U - 1) / (BLOCK * U);
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long i0 = c * (BLOCK * U) + threadIdx.x;
    f32x4 v[U];
    if (c + 1 < nchunks || n4 % (BLOCK * U) == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = ld<NTL>(x4 + i0 + u * BLOCK);
#pragma unroll
      for (int u = 0; u < U; ++u) st<NTS>(y4 + i0 + u * BLOCK, MODE ? v[u] : fq4(v[u], s, zp, 0.f, 255.f));
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) if (i0 + u * BLOCK < n4) v[u] = ld<NTL>(x4 + i0 + u * BLOCK);
#pragma unroll
      for (int u = 0; u < U; ++u) if (i0 + u * BLOCK < n4) st<NTS>(y4 + i0 + u * BLOCK, MODE ? v[u] : fq4(v[u], s, zp, 0.f, 255.f));
    }
  }
}


// ---- per-channel prototypes: tensor (outer, C, inner); slab = C*inner; chunk = BLOCK*4 elements ----
__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned mul, unsigned sh) { return (unsigned)(((unsigned long long)n * mul) >> sh); }

// (a) LDS-staged rows touched by the chunk, one barrier
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void kc_lds(const float* x, float* y, const float* sc, const float* zp, unsigned slab,
                                                unsigned cps, unsigned imul, unsigned ish, unsigned cmul, unsigned csh) {
  __shared__ float2 tbl[BLOCK * 4 + 2];
  const unsigned slab_i = fdiv(blockIdx.x, cmul, csh), cx = blockIdx.x - slab_i * cps;
  const unsigned e0 = cx * (BLOCK * 4), e_end = min(e0 + BLOCK * 4, slab);
  const unsigned ch0 = fdiv(e0, imul, ish), nrows = fdiv(e_end - 1, imul, ish) - ch0 + 1;
  const unsigned e = e0 + threadIdx.x * 4;
  f32x4 v;
  if (e < slab) v = __builtin_nontemporal_load((const f32x4*)(x + (long)slab_i * slab + e));
  for (unsigned t = threadIdx.x; t < nrows; t += BLOCK) tbl[t] = make_float2(sc[ch0 + t], zp[ch0 + t]);
  __syncthreads();
  if (e < slab) {
    const float2 so = tbl[fdiv(e, imul, ish) - ch0];
    __builtin_nontemporal_store(fq4(v, so.x, so.y, 0.f, 255.f), (f32x4*)(y + (long)slab_i * slab + e));
  }
}

// (b) no LDS: wave-uniform fast path through SGPRs (scalar loads), per-lane gather when a row boundary falls in the wave
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void kc_sgpr(const float* x, float* y, const float* sc, const float* zp, unsigned slab,
                                                 unsigned cps, unsigned imul, unsigned ish, unsigned cmul, unsigned csh) {
  const unsigned slab_i = fdiv(blockIdx.x, cmul, csh), cx = blockIdx.x - slab_i * cps;
  const unsigned e0 = cx * (BLOCK * 4);
  const unsigned e = e0 + threadIdx.x * 4;
  f32x4 v;
  if (e < slab) v = __builtin_nontemporal_load((const f32x4*)(x + (long)slab_i * slab + e));
  // wave-level range
  const unsigned w0 = e0 + (threadIdx.x & ~63u) * 4;
  const unsigned w1 = min(w0 + 255u, slab - 1);
  const unsigned r0 = __builtin_amdgcn_readfirstlane(fdiv(w0, imul, ish));
  const unsigned r1 = __builtin_amdgcn_readfirstlane(fdiv(w1, imul, ish));
  float s, z;
  if (r0 == r1) { s = sc[r0]; z = zp[r0]; }
  else { const unsigned r = fdiv(e < slab ? e : slab - 1, imul, ish); s = sc[r]; z = zp[r]; }
  if (e < slab) __builtin_nontemporal_store(fq4(v, s, z, 0.f, 255.f), (f32x4*)(y + (long)slab_i * slab + e));
}


// ---- observer prototype: min/max/absmax, one partial per block ----
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void kobs(const float* x, float* part, long n4) {
  const f32x4* x4 = (const f32x4*)x;
  float mx = -__builtin_inff(), mn = __builtin_inff(); unsigned ab = 0;
  const long nchunks = (n4 + BLOCK * U - 1) / (BLOCK * U);
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long i0 = c * (BLOCK * U) + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * BLOCK < n4) v[u] = __builtin_nontemporal_load(x4 + i0 + u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * BLOCK < n4) {
      float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { mx = fmaxf(mx, e[j]); mn = fminf(mn, e[j]); ab = max(ab, __float_as_uint(e[j]) & 0x7fffffffu); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); ab = max(ab, (unsigned)__shfl_xor((int)ab, o, 64)); }
  __shared__ float sm[3][16];
  const int w = threadIdx.x / 64;
  if ((threadIdx.x & 63) == 0) { sm[0][w] = mx; sm[1][w] = mn; sm[2][w] = __uint_as_float(ab); }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < BLOCK / 64; ++k) { mx = fmaxf(mx, sm[0][k]); mn = fminf(mn, sm[1][k]); ab = max(ab, __float_as_uint(sm[2][k])); }
    part[blockIdx.x] = mx; part[gridDim.x + blockIdx.x] = mn; part[2 * gridDim.x + blockIdx.x] = __uint_as_float(ab);
  }
}


// ---- codes-only prototypes (4 B read + 1 B written per element) ----
__device__ __forceinline__ unsigned code4(f32x4 v, float s, float zp) {
  float q0 = clampn(ster(v.x / s) + zp, 0.f, 255.f), q1 = clampn(ster(v.y / s) + zp, 0.f, 255.f);
  float q2 = clampn(ster(v.z / s) + zp, 0.f, 255.f), q3 = clampn(ster(v.w / s) + zp, 0.f, 255.f);
  return (unsigned)(int)q0 | ((unsigned)(int)q1 << 8) | ((unsigned)(int)q2 << 16) | ((unsigned)(int)q3 << 24);
}
// (a) lane i: float4 i of the chunk -> one dword store (256 B per wave-instruction)
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void kcode_a(const float* x, unsigned* y, const float* sc, long n4) {
  const float s = sc[0], zp = sc[1];
  const f32x4* x4 = (const f32x4*)x;
  const long nchunks = (n4 + BLOCK * U - 1) / (BLOCK * U);
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long i0 = c * (BLOCK * U) + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * BLOCK < n4) v[u] = __builtin_nontemporal_load(x4 + i0 + u * BLOCK);
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * BLOCK < n4) __builtin_nontemporal_store(code4(v[u], s, zp), y + i0 + u * BLOCK);
  }
}
// (b) lane i: four float4 at stride BLOCK (coalesced loads), codes exchanged so each lane stores 16 contiguous bytes
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void kcode_b(const float* x, unsigned* y, const float* sc, long n4) {
  const float s = sc[0], zp = sc[1];
  const f32x4* x4 = (const f32x4*)x;
  const long nchunks = (n4 + BLOCK * 4 - 1) / (BLOCK * 4);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    // wave w handles 256 consecutive float4: sub-row u = float4s [u*64, u*64+64)
    const long base = c * (BLOCK * 4) + wv * 256;
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (base + u * 64 + lane < n4) v[u] = __builtin_nontemporal_load(x4 + base + u * 64 + lane);
    unsigned cd[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) cd[u] = code4(v[u], s, zp);
    // dword d = u*64 + lane of this wave's 256 dwords; lane L wants dwords 4L..4L+3 = (u = L>>4, lanes 4(L&15)+j)
    u32x4 o;
    const int src = 4 * (lane & 15);
    unsigned t0[4], t1[4], t2[4], t3[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { t0[u] = __shfl(cd[u], src + 0, 64); t1[u] = __shfl(cd[u], src + 1, 64); t2[u] = __shfl(cd[u], src + 2, 64); t3[u] = __shfl(cd[u], src + 3, 64); }
    const int uu = lane >> 4;
    o.x = uu == 0 ? t0[0] : uu == 1 ? t0[1] : uu == 2 ? t0[2] : t0[3];
    o.y = uu == 0 ? t1[0] : uu == 1 ? t1[1] : uu == 2 ? t1[2] : t1[3];
    o.z = uu == 0 ? t2[0] : uu == 1 ? t2[1] : uu == 2 ? t2[2] : t2[3];
    o.w = uu == 0 ? t3[0] : uu == 1 ? t3[1] : uu == 2 ? t3[2] : t3[3];
    if (base + 4 * lane < n4) __builtin_nontemporal_store(o, (u32x4*)(y + base) + lane);
  }
}

struct Var { const char* name; void (*fn)(const float*, float*, const float*, long); int block, u, copy; };

#define V(B, U, NL, NS, M) {#B "x" #U " ntl=" #NL " nts=" #NS, (void (*)(const float*, float*, const float*, long))k<B, U, NL, NS, M>, B, U, M}

int main(int argc, char** argv) {
  long mb = argc > 1 ? atol(argv[1]) : 196;     // MiB of input
  int nbuf = argc > 2 ? atoi(argv[2]) : 4;      // rotate buffers to defeat the 256 MiB Infinity Cache
  int iters = argc > 3 ? atoi(argv[3]) : 24;
  long n = mb * 1024 * 1024 / 4, n4 = n / 4;
  std::vector<float*> xs(nbuf), ys(nbuf);
  for (int i = 0; i < nbuf; ++i) { hipMalloc(&xs[i], n * 4); hipMalloc(&ys[i], n * 4); hipMemset(xs[i], 0x3c, n * 4); }
  float h[2] = {0.0123f, 3.0f}; float* sc; hipMalloc(&sc, 8); hipMemcpy(sc, h, 8, hipMemcpyHostToDevice);
  Var vars[] = {
    V(64, 1, true, true, 0), V(128, 1, true, true, 0), V(256, 1, true, true, 0), V(512, 1, true, true, 0),
    V(64, 2, true, true, 0), V(256, 2, true, true, 0), V(256, 1, true, true, 1),
  };
  int grids[] = {0};   // 0 = one chunk per block (capped grid-stride grids lost 5-8% in the first sweep)
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("# %ld MiB in, %d rotating buffer pairs, %d iters; GB/s = 8 B/elem / median time\n", mb, nbuf, iters);
  for (auto& v : vars) {
    for (int g : grids) {
      long nchunks = (n4 + (long)v.block * v.u - 1) / ((long)v.block * v.u);
      long grid = g == 0 ? nchunks : std::min<long>(g, nchunks);
      if (g != 0 && grid == nchunks) continue;
      std::vector<float> ts;
      for (int it = 0; it < iters + 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(v.fn, dim3((unsigned)grid), dim3(v.block), 0, 0, xs[it % nbuf], ys[it % nbuf], sc, n4);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it >= 3) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      double med = ts[ts.size() / 2], mn = ts[0];
      printf("%-5s %-26s grid %-7ld med %8.2f us %7.1f GB/s   min %8.2f us %7.1f GB/s\n", v.copy ? "COPY" : "FQ", v.name, grid, med * 1e3,
             8.0 * n / med / 1e6, mn * 1e3, 8.0 * n / mn / 1e6);
      fflush(stdout);
    }
  }

  // ---- per-channel sweep: (outer, C, inner) with C*inner*outer = n ----
  {
    struct G { long C, inner; } geos[] = {{256, 3136}, {512, 784}, {1024, 196}, {2048, 48}};
    float *scv, *zpv; hipMalloc(&scv, 4096 * 4); hipMalloc(&zpv, 4096 * 4);
    std::vector<float> hs(4096, 0.0123f), hz(4096, 3.0f);
    hipMemcpy(scv, hs.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(zpv, hz.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (auto ge : geos) {
      unsigned slab = (unsigned)(ge.C * ge.inner);
      long outer = n / slab;
      for (int variant = 0; variant < 6; ++variant) {
        int block = variant % 3 == 0 ? 64 : variant % 3 == 1 ? 128 : 256;
        bool lds = variant < 3;
        unsigned CH = block * 4, cps = (slab + CH - 1) / CH;
        auto mk = [](unsigned d, unsigned& mul, unsigned& sh) { unsigned l = 0; while ((1ull << l) < d) ++l; sh = 31 + l; mul = (unsigned)(((1ull << sh) / d) + 1ull); };
        unsigned imul, ish, cmul, csh; mk((unsigned)ge.inner, imul, ish); mk(cps, cmul, csh);
        long grid = (long)cps * outer;
        std::vector<float> ts;
        for (int it = 0; it < iters + 3; ++it) {
          hipEventRecord(a);
#define LC(K, B) hipLaunchKernelGGL((K<B>), dim3((unsigned)grid), dim3(B), 0, 0, xs[it % nbuf], ys[it % nbuf], scv, zpv, slab, cps, imul, ish, cmul, csh)
          if (lds) { if (block == 64) LC(kc_lds, 64); else if (block == 128) LC(kc_lds, 128); else LC(kc_lds, 256); }
          else { if (block == 64) LC(kc_sgpr, 64); else if (block == 128) LC(kc_sgpr, 128); else LC(kc_sgpr, 256); }
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          if (it >= 3) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        double med = ts[ts.size() / 2];
        printf("CHAN  %s block %-4d C=%-5ld inner=%-5ld grid %-8ld med %8.2f us %7.1f GB/s\n", lds ? "lds " : "sgpr", block, ge.C, ge.inner, grid,
               med * 1e3, 8.0 * (double)outer * slab / med / 1e6);
        fflush(stdout);
      }
    }
  }

  // ---- observer sweep ----
  {
    float* part; hipMalloc(&part, 3 * 4 * (size_t)(1 << 22));
    struct OV { const char* name; void (*fn)(const float*, float*, long); int block, u; };
#define OVV(B, U) {#B "x" #U, (void (*)(const float*, float*, long))kobs<B, U>, B, U}
    OV ovs[] = {OVV(64, 4), OVV(64, 8), OVV(64, 16), OVV(128, 8), OVV(256, 2), OVV(256, 4), OVV(256, 8), OVV(256, 16), OVV(512, 4), OVV(1024, 4)};
    long caps[] = {0, 2048, 8192, 32768};
    for (auto& v : ovs) for (long cap : caps) {
      long nchunks = (n4 + (long)v.block * v.u - 1) / ((long)v.block * v.u);
      long grid = cap == 0 ? nchunks : std::min(cap, nchunks);
      if (cap != 0 && grid == nchunks) continue;
      if (grid > (1 << 22)) continue;
      std::vector<float> ts;
      for (int it = 0; it < iters + 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(v.fn, dim3((unsigned)grid), dim3(v.block), 0, 0, xs[it % nbuf], part, n4);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it >= 3) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      double med = ts[ts.size() / 2];
      printf("OBS   %-8s grid %-8ld med %8.2f us %7.1f GB/s\n", v.name, grid, med * 1e3, 4.0 * n / med / 1e6);
      fflush(stdout);
    }
  }

  // ---- codes-only sweep ----
  {
    struct CV { const char* name; void (*fn)(const float*, unsigned*, const float*, long); int block, u; };
#define CA(B, U) {"a " #B "x" #U, (void (*)(const float*, unsigned*, const float*, long))kcode_a<B, U>, B, U}
#define CB(B) {"b " #B "x4", (void (*)(const float*, unsigned*, const float*, long))kcode_b<B>, B, 4}
    CV cvs[] = {CA(64, 1), CA(64, 2), CA(64, 4), CA(64, 8), CA(256, 1), CA(256, 2), CA(256, 4), CA(256, 8), CB(64), CB(256)};
    long caps[] = {0, 2048, 8192};
    for (auto& v : cvs) for (long cap : caps) {
      long nchunks = (n4 + (long)v.block * v.u - 1) / ((long)v.block * v.u);
      long grid = cap == 0 ? nchunks : std::min(cap, nchunks);
      if (cap != 0 && grid == nchunks) continue;
      std::vector<float> ts;
      for (int it = 0; it < iters + 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(v.fn, dim3((unsigned)grid), dim3(v.block), 0, 0, xs[it % nbuf], (unsigned*)ys[it % nbuf], sc, n4);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (it >= 3) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      double med = ts[ts.size() / 2];
      printf("CODE  %-10s grid %-8ld med %8.2f us %7.1f GB/s\n", v.name, grid, med * 1e3, 5.0 * n / med / 1e6);
      fflush(stdout);
    }
  }
  return 0;



}
