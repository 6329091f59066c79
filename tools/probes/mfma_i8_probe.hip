// How fast does ONE wave issue v_mfma_i32_32x32x32_i8 back to back (8 independent accumulators), alone on its SIMD and with a partner?
// hipcc -O2 --offload-arch=gfx950 tools/probes/mfma_i8_probe.hip -o /tmp/mfma_i8_probe && /tmp/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512, 1) void probe(unsigned long long* out, int iters, int seed) {
  i32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0;
  // operands: zeros (seed 0) or hashed bytes (the clock a chip holds under load depends on the data: MI355X_MICROARCH.md, DVFS give-back)
  const unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  i32x4 a = {seed ? (int)(h * 3u) : 0, seed ? (int)(h * 7u + 1u) : 0, seed ? (int)(h * 11u + 5u) : 0, seed ? (int)(h * 13u + 9u) : 0};
  i32x4 b = {seed ? (int)(h * 17u) : 0, seed ? (int)(h * 19u + 3u) : 0, seed ? (int)(h * 23u + 7u) : 0, seed ? (int)(h * 29u + 1u) : 0};
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
  }
  int s = 0;
  for (int j = 0; j < NACC; ++j) s += acc[j][threadIdx.x & 15];
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = (r1 - r0) + ((unsigned long long)(s & 1) << 62);     // 100 MHz ticks
  }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8 * 2 * sizeof(unsigned long long));
  const int iters = 400000;       // ~ 0.1 - 0.2 s per launch: long enough for the clock to settle
  for (int threads : {256, 512}) {
    for (int rep = 0; rep < 4; ++rep) {
      hipMemset(d, 0, 256 * 8 * 2 * sizeof(unsigned long long));
      hipLaunchKernelGGL(probe<8>, dim3(256), dim3(threads), 0, 0, d, iters, rep < 2 ? 0 : 3 + rep);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256 * 8 * 2);
      hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double sum = 0, rt = 0;
      int n = 0;
      for (int b = 0; b < 256; ++b)
        for (int w = 0; w < threads / 64; ++w) { sum += (double)h[(b * 8 + w) * 2]; rt += (double)(h[(b * 8 + w) * 2 + 1] & ((1ull << 62) - 1)); ++n; }
      const double ghz = sum / rt * 0.1;       // shader clocks per 100 MHz tick
      const double tops = 256.0 * (threads / 64) * iters * 16.0 * 65536.0 / (rt / n * 1e-8) / 1e12;
      printf("%d waves per SIMD, %s operands: %.1f clocks per MFMA per wave (%.1f per SIMD), in-kernel clock %.2f GHz, %.0f TOP/s on the chip\n", threads / 256,
             rep < 2 ? "zero  " : "random", sum / n / (iters * 16.0), sum / n / (iters * 16.0) / (threads / 256), ghz, tops);
    }
  }
  return 0;
}
