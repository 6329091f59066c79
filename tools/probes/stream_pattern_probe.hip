// What rate does the CHAIN kernel's fp32 access pattern reach with no arithmetic at all?  (round 5, LABNOTES 16)
//
// conv_chain_i8_kernel reads a shortcut tile and writes an output tile of 64 rows x 256 B per chunk, chunk after chunk along a row of
// KD * 4 bytes, loads requested one chunk ahead, 2-3 workgroups of 4 waves per CU.  Its HBM-bound launches sit at ~5.4 TB/s where a
// plain stream of the same read-write mix (fq_tensor_kernel) reaches 6.4-6.8.  This probe issues exactly those loads and stores
// (buffer_load/store_dwordx4 ... offen nt, counted vmcnt) and nothing else, and varies what the kernel could vary:
//   map   0: the kernel's form (a wave-instruction = 8 rows x 128 B)   1: 4 rows x 256 B   2: 2 rows x 512 B (128-channel chunks)   3: 1 row x 1 KB
//   depth loads requested `depth` chunks ahead (1 = the kernel)
//   wgs   workgroups per CU (LDS padding)
//   gap   idle clocks between a chunk's loads landing and its stores (the arithmetic's place), in units of 64
//   pers  1: persistent workgroups (tiles dealt round-robin) instead of one workgroup per tile
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/stream_pattern_probe.hip -o tools/probes/stream_pattern_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i make_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  return v4i{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
template <bool NT>
__device__ __forceinline__ void bload16(f32x4& dst, int voff, const v4i& rsrc) {
  if (NT) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
template <bool NT>
__device__ __forceinline__ void bstore16(const f32x4& v, int voff, const v4i& rsrc) {
  if (NT) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
  else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}

struct Args {
  const float* in;
  float* out;
  int M, rowbytes, nchunks, gap, ntiles, cw, mode;   // mode 0: read + write, 1: stores only, 2: loads only   // cw: bytes of a chunk along the row (256, or 512 / 1024 for the wider maps)
};

// DEPTH chunks in flight; every wave issues 4 loads and 4 stores per chunk of 64 rows x 256 B (maps 0, 1) - or per 64 rows x cw bytes the
// matching multiple, so that a "chunk" is always 16 KB per workgroup and the queue arithmetic stays the same
template <int MAP, int DEPTH, bool NT, bool PERS>
__global__ __launch_bounds__(256) void pattern_kernel(Args a) {
  extern __shared__ int8_t pad[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t bytes = (uint32_t)((int64_t)a.M * a.rowbytes);
  const v4i r_in = make_rsrc(a.in, bytes), r_out = make_rsrc(a.out, bytes);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += PERS ? gridDim.x : a.ntiles) {
    const int64_t row0 = (int64_t)tile * 64;
    int fo[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int lr, lb;
      if (MAP == 0) { lr = (wave >> 1) * 32 + 8 * g + (lane >> 3); lb = (wave & 1) * 128 + (lane & 7) * 16; }
      else if (MAP == 1) { lr = wave * 16 + 4 * g + (lane >> 4); lb = (lane & 15) * 16; }
      else if (MAP == 2) { lr = wave * 8 + 2 * g + (lane >> 5); lb = (lane & 31) * 16; }      // 32 rows x 512 B per chunk
      else if (MAP == 3) { lr = wave * 4 + g; lb = lane * 16; }                                // 16 rows x 1 KB per chunk
      else { lr = wave * 16 + 4 * g + (lane >> 4); lb = (lane & 15) * 16; }                    // map 5: CHUNK-MAJOR tensor [KD/64][M][64]: a chunk of a tile is 16 KB contiguous
      fo[g] = row0 + lr < a.M ? (int)((row0 + lr) * (MAP == 5 ? 256 : a.rowbytes) + lb) : 0x7fff0000;
    }
    // chunk c of the tile: maps 0/1: column chunk c (256 B); map 2: rows 32 (c & 1).., column chunk c >> 1 (512 B); map 3: rows 16 (c & 3).., c >> 2 (1 KB)
    auto coff = [&](int c) {
      if (MAP == 5) return (int)((int64_t)c * a.M * 256);
      if (MAP <= 1) return c * 256;
      if (MAP == 2) return (c & 1) * 32 * a.rowbytes + (c >> 1) * 512;
      return (c & 3) * 16 * a.rowbytes + (c >> 2) * 1024;
    };
    f32x4 res[DEPTH + 1][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < a.nchunks) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bload16<NT>(res[d][g], a.mode == 1 ? 0x7fff0000 : fo[g] + coff(d), r_in);
      }
    for (int n0 = 0; n0 < a.nchunks; n0 += DEPTH + 1) {
#pragma unroll
      for (int u = 0; u < DEPTH + 1; ++u) {
        const int n = n0 + u;
        if (n >= a.nchunks) break;
        // queue, oldest first: behind loads(n) sit the load groups n + 1 .. n + DEPTH - 1 that exist and the store groups of the (up to DEPTH)
        // chunks worked on since loads(n) was issued: that many operations may stay in flight
        {
          const int yl = a.nchunks - 1 - n < DEPTH - 1 ? a.nchunks - 1 - n : DEPTH - 1, ys = n < DEPTH ? n : DEPTH;
          switch (yl + ys) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(res[u][g]));
        if (n + DEPTH < a.nchunks) {
#pragma unroll
          for (int g = 0; g < 4; ++g) bload16<NT>(res[(u + DEPTH) % (DEPTH + 1)][g], a.mode == 1 ? 0x7fff0000 : fo[g] + coff(n + DEPTH), r_in);
        }
        for (int i = 0; i < a.gap; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) bstore16<NT>(res[u][g] + 1.0f, a.mode == 2 ? 0x7fff0000 : fo[g] + coff(n), r_out);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (pad[0] == 77 && tid == 99999) a.out[0] = 0.0f;
}

// The comparison: a LINEAR SWEEP (what fq_tensor_kernel does) - piece p = round * grid + workgroup, PIECE bytes each, consecutive workgroups
// on consecutive memory, loads one piece ahead.  The chip's workgroups then sit in one window of grid * PIECE bytes that moves through the tensor.
template <int PIECE, bool NT>
__global__ __launch_bounds__(256) void sweep_kernel(Args a) {
  extern __shared__ int8_t pad[];
  const int tid = threadIdx.x;
  const int64_t total = (int64_t)a.M * a.rowbytes;
  const uint32_t bytes = (uint32_t)total;
  const v4i r_in = make_rsrc(a.in, bytes), r_out = make_rsrc(a.out, bytes);
  constexpr int NL = PIECE / 4096;          // 16-byte loads per thread and piece
  const int64_t npieces = total / PIECE;
  f32x4 res[2][NL];
  int64_t p = blockIdx.x;
  if (p >= npieces) return;
#pragma unroll
  for (int g = 0; g < NL; ++g) bload16<NT>(res[0][g], (int)(p * PIECE + g * 4096 + tid * 16), r_in);
  bool first = true;
  for (;;) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");
      first = false;
#pragma unroll
      for (int g = 0; g < NL; ++g) asm volatile("" : "+v"(res[u][g]));
      const int64_t pn = p + gridDim.x;
      if (pn < npieces) {
#pragma unroll
        for (int g = 0; g < NL; ++g) bload16<NT>(res[1 - u][g], (int)(pn * PIECE + g * 4096 + tid * 16), r_in);
      }
      for (int i = 0; i < a.gap; ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int g = 0; g < NL; ++g) bstore16<NT>(res[u][g] + 1.0f, (int)(p * PIECE + g * 4096 + tid * 16), r_out);
      p = pn;
      if (p >= npieces) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (pad[0] == 77 && tid == 99999) a.out[0] = 0.0f;
        return;
      }
    }
  }
}

template <int PIECE>
static float run_sweep(const Args& a, int wgs, int cus) {
  const size_t dyn = wgs >= 8 ? 0 : (size_t)(160 * 1024 / wgs - 2048);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0, 0);
    hipFuncSetAttribute((const void*)sweep_kernel<PIECE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipLaunchKernelGGL((sweep_kernel<PIECE, true>), dim3(cus * wgs), dim3(256), dyn, 0, a);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return best;
}

template <int MAP, int DEPTH>
static float run(const Args& a, int wgs, bool nt, bool pers, int cus) {
  // LDS padding: 160 KB / wgs (minus a little) keeps `wgs` workgroups on a CU
  const size_t dyn = wgs >= 8 ? 0 : (size_t)(160 * 1024 / wgs - 2048);
  const int grid = pers ? cus * wgs : a.ntiles;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0, 0);
#define GO(NT_, PERS_)                                                                                               \
  do {                                                                                                               \
    hipFuncSetAttribute((const void*)pattern_kernel<MAP, DEPTH, NT_, PERS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
    hipLaunchKernelGGL((pattern_kernel<MAP, DEPTH, NT_, PERS_>), dim3(grid), dim3(256), dyn, 0, a);                  \
  } while (0)
    if (nt && pers) GO(true, true);
    else if (nt) GO(true, false);
    else if (pers) GO(false, true);
    else GO(false, false);
#undef GO
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return best;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  struct Shape { const char* name; int M, KD; };
  const Shape shapes[] = {{"56^2 mid (KD 256)", 512 * 56 * 56, 256}, {"28^2 mid (KD 512)", 512 * 28 * 28, 512}, {"14^2 (KD 1024)", 512 * 14 * 14, 1024}};
  for (const Shape& s : shapes) {
    const size_t bytes = (size_t)s.M * s.KD * 4;
    float *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0, bytes);
    hipMemset(out, 0, bytes);
    Args a{in, out, s.M, s.KD * 4, s.KD * 4 / 256, 0, (s.M + 63) / 64, 256};
    printf("== %s: %d rows x %d B, %.0f MB read + %.0f MB written, %d tiles of 64 rows x %d chunks\n", s.name, s.M, s.KD * 4, bytes / 1e6, bytes / 1e6, a.ntiles, a.nchunks);
    auto report = [&](const char* what, float ms) { printf("  %-70s %8.1f us  %5.2f TB/s\n", what, ms * 1e3, 2.0 * bytes / ms / 1e9); fflush(stdout); };
    char buf[160];
    for (int wgs : {2, 3, 4, 8}) {
      snprintf(buf, sizeof buf, "LINEAR SWEEP, 16 KB pieces, %d wg/CU, nt", wgs);
      report(buf, run_sweep<16384>(a, wgs, cus));
    }
    report("LINEAR SWEEP, 8 KB pieces, 3 wg/CU, nt", run_sweep<8192>(a, 3, cus));
    report("LINEAR SWEEP, 8 KB pieces, 8 wg/CU, nt", run_sweep<8192>(a, 8, cus));
    report("LINEAR SWEEP, 4 KB pieces, 8 wg/CU, nt", run_sweep<4096>(a, 8, cus));
    for (int gap : {32, 64, 96}) {
      Args b = a;
      b.gap = gap;
      snprintf(buf, sizeof buf, "LINEAR SWEEP, 16 KB pieces, 3 wg/CU, nt, gap %d x 64 clocks", gap);
      report(buf, run_sweep<16384>(b, 3, cus));
    }
    for (int wgs : {2, 3, 4, 8}) {
      snprintf(buf, sizeof buf, "map 0 (8 rows x 128 B), depth 1, %d wg/CU, nt", wgs);
      report(buf, run<0, 1>(a, wgs, true, false, cus));
    }
    report("map 0, depth 1, 3 wg/CU, temporal", run<0, 1>(a, 3, false, false, cus));
    report("map 0, depth 2, 3 wg/CU, nt", run<0, 2>(a, 3, true, false, cus));
    report("map 0, depth 3, 3 wg/CU, nt", run<0, 3>(a, 3, true, false, cus));
    report("map 0, depth 1, 3 wg/CU, nt, persistent", run<0, 1>(a, 3, true, true, cus));
    report("map 1 (4 rows x 256 B), depth 1, 3 wg/CU, nt", run<1, 1>(a, 3, true, false, cus));
    report("map 2 (2 rows x 512 B), depth 1, 3 wg/CU, nt", run<2, 1>(a, 3, true, false, cus));
    report("map 3 (1 row x 1 KB), depth 1, 3 wg/CU, nt", run<3, 1>(a, 3, true, false, cus));
    report("map 3, depth 2, 3 wg/CU, nt", run<3, 2>(a, 3, true, false, cus));
    report("map 5 (chunk-major planes [KD/64][M][64]), depth 1, 2 wg/CU, nt", run<5, 1>(a, 2, true, false, cus));
    report("map 5 (chunk-major planes [KD/64][M][64]), depth 1, 3 wg/CU, nt", run<5, 1>(a, 3, true, false, cus));
    report("map 5, depth 2, 3 wg/CU, nt", run<5, 2>(a, 3, true, false, cus));
    {
      auto half = [&](const char* what, float ms) { printf("  %-70s %8.1f us  %5.2f TB/s (one direction)\n", what, ms * 1e3, 1.0 * bytes / ms / 1e9); fflush(stdout); };
      Args w = a;
      w.mode = 1;
      half("STORES ONLY, map 0 (row-major tiles), 3 wg/CU, nt", run<0, 1>(w, 3, true, false, cus));
      half("STORES ONLY, map 5 (chunk-major planes), 3 wg/CU, nt", run<5, 1>(w, 3, true, false, cus));
      half("STORES ONLY, map 0, 3 wg/CU, temporal", run<0, 1>(w, 3, false, false, cus));
      half("STORES ONLY, map 5, 3 wg/CU, temporal", run<5, 1>(w, 3, false, false, cus));
      w.gap = 32;
      half("STORES ONLY, map 0, 3 wg/CU, nt, gap 32 x 64 clocks", run<0, 1>(w, 3, true, false, cus));
      half("STORES ONLY, map 5, 3 wg/CU, nt, gap 32 x 64 clocks", run<5, 1>(w, 3, true, false, cus));
      w.gap = 0;
      w.mode = 2;
      half("LOADS ONLY, map 0, 3 wg/CU, nt", run<0, 1>(w, 3, true, false, cus));
      half("LOADS ONLY, map 5, 3 wg/CU, nt", run<5, 1>(w, 3, true, false, cus));
    }
    for (int gap : {32, 64, 96}) {
      Args b = a;
      b.gap = gap;
      snprintf(buf, sizeof buf, "map 5, depth 1, 3 wg/CU, nt, gap %d x 64 clocks", gap);
      report(buf, run<5, 1>(b, 3, true, false, cus));
      snprintf(buf, sizeof buf, "map 5, depth 1, 2 wg/CU, nt, gap %d x 64 clocks", gap);
      report(buf, run<5, 1>(b, 2, true, false, cus));
    }
    for (int gap : {16, 32, 64, 96, 128}) {
      Args b = a;
      b.gap = gap;
      snprintf(buf, sizeof buf, "map 0, depth 1, 3 wg/CU, nt, gap %d x 64 clocks", gap);
      report(buf, run<0, 1>(b, 3, true, false, cus));
    }
    for (int gap : {32, 64, 96}) {
      Args b = a;
      b.gap = gap;
      snprintf(buf, sizeof buf, "map 0, depth 2, 3 wg/CU, nt, gap %d x 64 clocks", gap);
      report(buf, run<0, 2>(b, 3, true, false, cus));
      snprintf(buf, sizeof buf, "map 0, depth 1, 2 wg/CU, nt, gap %d x 64 clocks", gap);
      report(buf, run<0, 1>(b, 2, true, false, cus));
    }
    hipFree(in);
    hipFree(out);
  }
  return 0;
}
