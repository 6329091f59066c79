// VALU issue-rate probe: cycles per instruction per SIMD for a few instructions, at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITERS 256
#define STR(x) #x
#define XSTR(x) STR(x)
template <int OP>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* clk, float a, float b) {
  float v0 = threadIdx.x * a, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0{v0, v1}, p1{v2, v3}, p2{v4, v5}, p3{v6, v7};
  f2 pa{a, a}, pb{b, b};
  uint32_t u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
      if (OP == 2) asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
      if (OP == 3) asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));
      if (OP == 4) asm volatile("v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %5, 1, %1\n v_cvt_pk_u8_f32 %2, %6, 2, %2\n v_cvt_pk_u8_f32 %3, %7, 3, %3\n v_cvt_pk_u8_f32 %0, %4, 1, %0\n v_cvt_pk_u8_f32 %1, %5, 2, %1\n v_cvt_pk_u8_f32 %2, %6, 3, %2\n v_cvt_pk_u8_f32 %3, %7, 0, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
      if (OP == 5) asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
      if (OP == 6) asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4\n v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0x05010400u));
      if (OP == 7) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0\n v_dot4_i32_i8 %1, %2, %3, %1\n v_dot4_i32_i8 %2, %3, %0, %2\n v_dot4_i32_i8 %3, %0, %1, %3\n v_dot4_i32_i8 %0, %1, %2, %0\n v_dot4_i32_i8 %1, %2, %3, %1\n v_dot4_i32_i8 %2, %3, %0, %2\n v_dot4_i32_i8 %3, %0, %1, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
      if (OP == 8) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
      if (OP == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a) : "vcc");
      if (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %5\n v_pk_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
      if (OP == 11) asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0\n v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
      if (OP == 20) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(a), "v"(b));
      if (OP == 21) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(v0), "+v"(v1) : "v"(a), "v"(b));
      if (OP == 22) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a), "v"(b));
      if (OP == 23) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0\n v_cvt_pk_u8_f32 %0, %2, 1, %0\n v_cvt_pk_u8_f32 %0, %3, 2, %0\n v_cvt_pk_u8_f32 %0, %4, 3, %0\n v_cvt_pk_u8_f32 %0, %1, 0, %0\n v_cvt_pk_u8_f32 %0, %2, 1, %0\n v_cvt_pk_u8_f32 %0, %3, 2, %0\n v_cvt_pk_u8_f32 %0, %4, 3, %0" : "+v"(u0) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
      if (OP == 24) asm volatile("v_cmp_nlt_f32_e64 s[20:21], |%0|, %4\n v_cmp_nlt_f32_e64 s[22:23], |%1|, %4\n s_or_b64 s[20:21], s[20:21], s[22:23]\n v_cmp_nlt_f32_e64 s[22:23], |%2|, %4\n s_or_b64 s[20:21], s[20:21], s[22:23]\n v_cmp_nlt_f32_e64 s[22:23], |%3|, %4\n s_or_b64 s[20:21], s[20:21], s[22:23]\n v_cmp_nlt_f32_e64 s[24:25], |%0|, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "scc");
      if (OP == 25) asm volatile("v_add_u32 %0, %0, %4\n v_cvt_f32_i32 %0, %0\n v_add_u32 %1, %1, %4\n v_cvt_f32_i32 %1, %1\n v_add_u32 %2, %2, %4\n v_cvt_f32_i32 %2, %2\n v_add_u32 %3, %3, %4\n v_cvt_f32_i32 %3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));
      if (OP == 26) asm volatile("v_rndne_f32 %0, %0\n v_med3_f32 %0, %0, %1, %2\n v_rndne_f32 %0, %0\n v_med3_f32 %0, %0, %1, %2\n v_rndne_f32 %0, %0\n v_med3_f32 %0, %0, %1, %2\n v_rndne_f32 %0, %0\n v_med3_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(a), "v"(b));
      if (OP == 12) asm volatile("v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %1, %1, 8, %2\n v_lshl_or_b32 %2, %2, 8, %3\n v_lshl_or_b32 %3, %3, 8, %0\n v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %1, %1, 8, %2\n v_lshl_or_b32 %2, %2, 8, %3\n v_lshl_or_b32 %3, %3, 8, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  p0 += p1 + p2 + p3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0.x + p0.y + (float)(u0 + u1 + u2 + u3);
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, float* out, unsigned long long* clk) {
  printf("%-22s", name);
  for (int wps : {1, 2, 4, 8}) {          // waves per SIMD: one workgroup of 256 * wps threads on one CU... use wps workgroups of 256 threads, grid = 256 CUs * wps
    int grid = 256 * wps;
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, out, clk, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, out, clk, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    unsigned long long h[2048];
    hipMemcpy(h, clk, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < grid; ++i) s += (double)h[i];
    s /= grid;
    // cycles per instruction per SIMD: each SIMD runs wps waves, each REP * ITERS instructions
    printf("  wps %d: %6.2f clk/instr/SIMD", wps, s / ((double)REP * ITERS * wps));
  }
  printf("\n");
  fflush(stdout);
}
__global__ void sem(const float* in, uint32_t* o, int n) {
  int i = threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
  {
    float h[16] = {-5.f, -0.4f, 0.5f, 1.5f, 2.5f, 0.49f, 0.51f, 254.5f, 255.5f, 300.f, 1e9f, __builtin_nanf(""), __builtin_inff(), -__builtin_inff(), 3.0f, 254.f};
    float* d; uint32_t* o; uint32_t ho[16];
    hipMalloc(&d, 64); hipMalloc(&o, 64);
    hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d, o, 16);
    hipMemcpy(ho, o, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("cvt_pk_u8_f32(%g) = %u\n", h[i], ho[i]);
  }
  float* out; unsigned long long* clk;
  hipMalloc(&out, 2048 * 256 * 4);
  hipMalloc(&clk, 2048 * 8);
  run<0>("v_fma_f32", out, clk);
  run<1>("v_pk_fma_f32", out, clk);
  run<10>("v_pk_add/mul_f32", out, clk);
  run<2>("v_rndne_f32", out, clk);
  run<3>("v_med3_f32", out, clk);
  run<4>("v_cvt_pk_u8_f32", out, clk);
  run<5>("v_cvt_f32_i32", out, clk);
  run<6>("v_perm_b32", out, clk);
  run<7>("v_dot4_i32_i8", out, clk);
  run<8>("v_add_u32", out, clk);
  run<9>("v_cmp_lt_f32", out, clk);
  run<11>("v_mul_lo_u32", out, clk);
  run<12>("v_lshl_or_b32", out, clk);
  run<20>("fma dep chain x1", out, clk);
  run<21>("fma dep chains x2", out, clk);
  run<22>("fma dep chains x4", out, clk);
  run<23>("cvt_pk_u8 dep chain", out, clk);
  run<24>("cmp->sgpr + s_or", out, clk);
  run<25>("add_u32->cvt pairs", out, clk);
  run<26>("rndne->med3 chain", out, clk);
  // cvt_pk_u8 semantics
  return 0;
}
