// What v_cvt_pk_u8_f32 does outside [0, 255] on gfx950 (is the quantiser's clamp redundant for an unsigned byte?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, unsigned* o, int n) {
  const int i = threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}
int main() {
  float h[] = {-5.0f, -1.0f, -0.0f, 0.0f, 1.0f, 254.0f, 255.0f, 256.0f, 300.0f, 1e9f, 3e38f, INFINITY, -INFINITY, NAN, 0.5f, 1.5f, 2.5f, 254.5f, 255.5f, -0.5f};
  const int n = sizeof h / sizeof h[0];
  float* d; unsigned* o; unsigned r[32];
  hipMalloc(&d, sizeof h); hipMalloc(&o, 32 * 4);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n);
  hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%g -> %u\n", h[i], r[i] & 0xff);
  return 0;
}
