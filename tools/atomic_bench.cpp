// Split-K accumulation of fp32 tiles on gfx950: what does it cost to combine `splits` partial 256x256 fp32 tiles per output tile?
//   atomic_bench [tiles] [splits]
// mode 0: every workgroup STORES its 256 KiB partial tile to its own slab (what gemm_tn_p8 does), then a reduce kernel sums the slabs into C
// mode 1: every workgroup adds its partial tile to C with global_atomic_add_f32 (no return value)
// mode 2: like 1 with global_atomic_pk_add... (not available for fp32) -> skipped
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void slab_store(float* slabs, int tiles) {
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  float* dst = slabs + ((long)split * tiles + tile) * 65536;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll 8
  for (int i = 0; i < 32; ++i) *(f32x4*)(dst + (i * 512 + threadIdx.x) * 4) = v;
}
__global__ __launch_bounds__(256) void slab_reduce(const float* slabs, float* C, int tiles, int splits) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  f32x4 a = *(const f32x4*)(C + i);
  for (int s = 0; s < splits; ++s) a += *(const f32x4*)(slabs + (long)s * tiles * 65536 + i);
  *(f32x4*)(C + i) = a;
}
template <int ORDER>
__global__ __launch_bounds__(512) void atomic_acc(float* C, int tiles) {
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  float* dst = C + (long)tile * 65536;
  // ORDER 1: start each split at a different offset of the tile so that concurrent workgroups do not hit the same lines together
  const int rot = ORDER ? (split * 5) & 31 : 0;
#pragma unroll 8
  for (int i = 0; i < 32; ++i) {
    float* p = dst + (((i + rot) & 31) * 512 + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(p + j, 1.0f);
  }
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 9, splits = argc > 2 ? atoi(argv[2]) : 28;
  float *slabs, *C;
  CK(hipMalloc(&slabs, (size_t)tiles * splits * 262144));
  CK(hipMalloc(&C, (size_t)tiles * 262144));
  CK(hipMemset(C, 0, (size_t)tiles * 262144));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto fn, const char* what) {
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {
      CK(hipEventRecord(e0, 0)); fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("  %-46s %8.1f us   (%.1f MB of partial tiles -> %.0f GB/s)\n", what, best * 1e3, tiles * splits * 0.262144, tiles * splits * 262144.0 / (best * 1e-3) / 1e9);
  };
  printf("%d tiles x %d splits (= %d workgroups)\n", tiles, splits, tiles * splits);
  timeit([&] { hipLaunchKernelGGL(slab_store, dim3(tiles * splits), dim3(512), 0, 0, slabs, tiles); }, "slab store");
  timeit([&] { hipLaunchKernelGGL(slab_reduce, dim3(tiles * 64), dim3(256), 0, 0, slabs, C, tiles, splits); }, "slab reduce");
  timeit([&] { hipLaunchKernelGGL(slab_store, dim3(tiles * splits), dim3(512), 0, 0, slabs, tiles);
               hipLaunchKernelGGL(slab_reduce, dim3(tiles * 64), dim3(256), 0, 0, slabs, C, tiles, splits); }, "slab store + reduce");
  timeit([&] { hipLaunchKernelGGL(atomic_acc<0>, dim3(tiles * splits), dim3(512), 0, 0, C, tiles); }, "atomic add, same order in every split");
  timeit([&] { hipLaunchKernelGGL(atomic_acc<1>, dim3(tiles * splits), dim3(512), 0, 0, C, tiles); }, "atomic add, rotated start per split");
  float h[4]; CK(hipMemcpy(h, C, 16, hipMemcpyDeviceToHost)); printf("  C[0] = %.1f\n", h[0]);
  return 0;
}
