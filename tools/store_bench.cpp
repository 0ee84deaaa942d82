// Per-CU global STORE throughput on gfx950 (what bounds a GEMM epilogue): one 512-thread workgroup per CU writes
// `tiles` x 128 KiB (a 256x256 bf16 output tile) with 16 B per lane, in several address patterns / cache policies.
//   store_bench [ncu] [tiles]
// pattern 0: wave instruction = 1 KiB contiguous                      (best case)
// pattern 1: wave instruction = 8 rows x 128 B, row stride 6 KiB      (the GEMM epilogue: N = 3072 bf16)
// pattern 2: wave instruction = 16 rows x 64 B, row stride 6 KiB      (direct-from-accumulator stores after a permlane swap)
// pattern 3: 8 B per lane, 16 rows x 32 B                             (raw 16x16 MFMA accumulator layout)
// pattern 4: 8 full rows per instruction like pattern 1, but a row's 8 lanes are scattered: lane (m = l & 15, g = l >> 4) writes
//            row m & ~1 (+1 in the second instruction), chunk (m & 1) * 4 + {0,2,1,3}[g]   (transpose-free MFMA epilogue)
// policy 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int POL>
__device__ __forceinline__ void st16(char* p, f32x4 v) {
  if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int POL>
__device__ __forceinline__ void st8(char* p, f32x2 v) {
  if constexpr (POL == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if constexpr (POL == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  if constexpr (POL == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if constexpr (POL == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// The workgroup's tile t is a 256 x 256 bf16 block (row stride LD bytes) at column block (wg % 12), row block wg / 12 + t * gridrows.
template <int PAT, int POL>
__global__ __launch_bounds__(512) void store_kernel(char* out, long ld, int tiles, int rowblocks) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  f32x4 v = {1.f * tid, 2.f, 3.f, 4.f};
  for (int t = 0; t < tiles; ++t) {
    const long rb = (long)(blockIdx.x / 12) + (long)t * rowblocks, cb = blockIdx.x % 12;
    char* tile = out + rb * 256 * ld + cb * 512;
    char* wt = tile + (long)(wr * 128) * ld + wc * 128;       // the wave's 128 rows x 64 columns (128 B per row)
    if constexpr (PAT == 0) {
      char* p = out + ((long)blockIdx.x * tiles + t) * 131072 + wave * 16384 + lane * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) st16<POL>(p + i * 1024, v);
    } else if constexpr (PAT == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st16<POL>(wt + (long)(i * 8 + (lane >> 3)) * ld + (lane & 7) * 16, v);
    } else if constexpr (PAT == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) st16<POL>(wt + (long)((i >> 1) * 16 + (lane & 15)) * ld + (i & 1) * 64 + (lane >> 4) * 16, v);
    } else if constexpr (PAT == 4) {
      const int m = lane & 15, g = lane >> 4, chunk = (m & 1) * 4 + ((g & 1) << 1 | (g >> 1));
#pragma unroll
      for (int i = 0; i < 16; ++i) st16<POL>(wt + (long)((i >> 1) * 16 + (m & ~1) + (i & 1)) * ld + chunk * 16, v);
    } else {
      f32x2 w = {v[0], v[1]};
#pragma unroll
      for (int i = 0; i < 32; ++i) st8<POL>(wt + (long)((i >> 2) * 16 + (lane & 15)) * ld + (i & 3) * 32 + (lane >> 4) * 8, w);
    }
  }
}

template <int PAT, int POL>
static void run(char* buf, int ncu, int tiles, long ld) {
  const int rowblocks = (ncu + 11) / 12;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((store_kernel<PAT, POL>), dim3(ncu), dim3(512), 0, 0, buf, ld, tiles, rowblocks);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes = (double)ncu * tiles * 131072.0;
  printf("  pattern %d policy %d: %8.1f us  %7.1f GB/s total  %6.2f GB/s per CU = %5.1f B/clk @2.4 GHz\n", PAT, POL, best * 1e3, bytes / (best * 1e-3) / 1e9,
         bytes / (best * 1e-3) / 1e9 / ncu, bytes / (best * 1e-3) / ncu / 2.4e9);
}

int main(int argc, char** argv) {
  const int ncu = argc > 1 ? atoi(argv[1]) : 256, tiles = argc > 2 ? atoi(argv[2]) : 16;
  const long ld = 6144;
  const size_t bytes = (size_t)((ncu + 11) / 12) * tiles * 256 * ld + (size_t)ncu * tiles * 131072 + (1 << 20);
  char* buf; CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  printf("%d workgroups x %d tiles of 128 KiB:\n", ncu, tiles);
  run<0, 0>(buf, ncu, tiles, ld); run<0, 1>(buf, ncu, tiles, ld); run<0, 2>(buf, ncu, tiles, ld); run<0, 3>(buf, ncu, tiles, ld);
  run<1, 0>(buf, ncu, tiles, ld); run<1, 1>(buf, ncu, tiles, ld); run<1, 2>(buf, ncu, tiles, ld);
  run<2, 0>(buf, ncu, tiles, ld); run<2, 1>(buf, ncu, tiles, ld);
  run<3, 0>(buf, ncu, tiles, ld); run<3, 1>(buf, ncu, tiles, ld);
  run<4, 0>(buf, ncu, tiles, ld);
  return 0;
}
