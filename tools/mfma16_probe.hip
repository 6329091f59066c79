// Layout probe for v_mfma_i32_16x16x64_i8 and v_permlane{16,32}_swap on gfx950 (run on the GPU box): prints which (m, n) of
// D = A x B^T each (lane, register) holds, and what the two swaps exchange.  hipcc --offload-arch=gfx950 tools/mfma16_probe.hip -o tools/mfma16_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int* d, int* s32, int* s16) {
  const int l = threadIdx.x;
  // A: row m = l % 16 holds value m + 1 at k = 0 (lane group 0, byte 0); B: column n = l % 16 holds n + 1 at k = 0, 17 * 0 elsewhere
  i32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  if (l < 16) { a.x = (l + 1); b.x = (l + 1) + 16 > 127 ? 0 : (l + 1); }
  i32x4 acc = {0, 0, 0, 0};
  // make m and n distinguishable: D[m][n] = (m + 1) * (n + 1) is symmetric, so run a second product with B doubled in the upper half
  acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
  i32x4 b2 = b;
  if (l < 16) b2.x = (l >= 8) ? 2 * (l + 1) : (l + 1);
  i32x4 acc2 = {0, 0, 0, 0};
  acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b2, acc2, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[256 + l * 4 + i] = acc2[i];
  auto r = __builtin_amdgcn_permlane32_swap(l, 1000 + l, false, false);
  s32[l * 2] = r[0]; s32[l * 2 + 1] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(l, 1000 + l, false, false);
  s16[l * 2] = q[0]; s16[l * 2 + 1] = q[1];
}
int main() {
  int *d, *s32, *s16;
  hipMalloc(&d, 512 * 4); hipMalloc(&s32, 128 * 4); hipMalloc(&s16, 128 * 4);
  probe<<<1, 64>>>(d, s32, s16);
  int h[512], a[128], b[128];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(a, s32, sizeof a, hipMemcpyDeviceToHost); hipMemcpy(b, s16, sizeof b, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) {
    printf("lane %2d:", l);
    for (int i = 0; i < 4; ++i) {
      const int v = h[l * 4 + i], v2 = h[256 + l * 4 + i];
      // v = (m+1)(n+1); v2 = v * (n >= 8 ? 2 : 1)
      printf("  r%d=%4d/%4d", i, v, v2);
    }
    printf("   | swap32: dst=%4d src=%4d | swap16: dst=%4d src=%4d\n", a[l * 2], a[l * 2 + 1], b[l * 2], b[l * 2 + 1]);
  }
  return 0;
}
