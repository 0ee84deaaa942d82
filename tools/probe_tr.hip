// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds lds[e] = e (16-bit); lane l supplies byte address l*8.
// Output: for every destination (lane, elem) the 16-bit value it received = 4*srclane + srcelem.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 8;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  unsigned short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (L%2d,e%d)", h[l*4+j] / 4, h[l*4+j] % 4); printf("\n"); }
  return 0;
}
