// Semantics probe: v_permlane16_swap_b32 and the quad_perm [1,0,3,2] DPP move on gfx950 (used by the register-transposed GEMM
// epilogue).  hipcc --offload-arch=gfx950 tools/probe_swap.hip -o build/probe_swap && build/probe_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  unsigned x = __builtin_amdgcn_update_dpp(0u, a, 0xB1, 0xf, 0xf, false);
  out[threadIdx.x * 3 + 0] = r[0];
  out[threadIdx.x * 3 + 1] = r[1];
  out[threadIdx.x * 3 + 2] = x;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 3 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane: vdst' src' dpp_xor1   (vdst = lane, src = 100 + lane)\n");
  for (int l = 0; l < 64; l += 1) if ((l & 15) == 0 || (l & 15) == 1 || (l & 15) == 15) printf("%2d: %3u %3u %2u\n", l, h[l * 3], h[l * 3 + 1], h[l * 3 + 2]);
  return 0;
}
