// per-launch cost of (nearly) empty kernels by workgroup size / dynamic LDS size / register count
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ int lds_dyn[];
template <int REGS>
__global__ __launch_bounds__(1024) void k_empty(int* out, int n) {
  int v[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) v[i] = threadIdx.x * (i + 1) + n;
  if (n == 12345) {
    lds_dyn[threadIdx.x] = 1;
    int s = 0;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += v[i] * v[(i + 1) % REGS];
    out[threadIdx.x] = s + lds_dyn[(threadIdx.x + 1) & 63];
  }
}
template <int REGS>
void run(const char* name, int grid, int block, int lds, int* out) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_empty<REGS>, dim3(grid), dim3(block), lds, 0, out, w);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  const int reps = 50;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_empty<REGS>, dim3(grid), dim3(block), lds, 0, out, r);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-40s grid %5d block %4d lds %6d : %7.2f us per launch\n", name, grid, block, lds, ms * 1e3 / reps);
  fflush(stdout);
}
int main() {
  int* out;
  hipMalloc(&out, 4096 * 4);
  run<4>("small", 256, 256, 0, out);
  run<4>("small, 768 threads", 256, 768, 0, out);
  run<4>("small, 768 threads, 119 KB LDS", 256, 768, 119 * 1024, out);
  run<4>("small, 256 threads, 50 KB LDS", 768, 256, 50 * 1024, out);
  run<4>("small, 256 threads, 35 KB LDS x768", 768, 256, 35 * 1024, out);
  run<4>("small, 768 threads, 60 KB LDS", 256, 768, 60 * 1024, out);
  run<4>("small, 768 threads, 70 KB LDS", 256, 768, 70 * 1024, out);
  run<4>("small, 1024 threads, 150 KB LDS", 256, 1024, 150 * 1024, out);
  run<4>("6272 small workgroups", 6272, 256, 0, out);
  run<4>("6272 workgroups, 40 KB LDS", 6272, 256, 40 * 1024, out);
  return 0;
}
