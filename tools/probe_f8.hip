// Probe of v_mfma_f32_16x16x128_f8f6f4 with FP8 (E4M3, OCP) operands and of v_cvt_pk_fp8_f32 on gfx950.
// Assumed layout (checked here against a host reference): lane l supplies row (l % 16) of its operand and the 32 k-bytes
// [32 * (l / 16), +32) as 8 VGPRs; D[i][j] = sum_k A[i][k] B[j][k]; lane l holds D[4 * (l / 16) + r][l % 16] in register r.
// Scale operands 0 must mean "no scaling".  Prints the max abs error against the host and the encodings of a few values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const uint8_t* A, const uint8_t* B, float* D, const float* vals, uint8_t* enc, int nvals) {
  const int l = threadIdx.x, row = l & 15, g = l >> 4;
  v8i a, b;
  memcpy(&a, A + row * 128 + g * 32, 32);
  memcpy(&b, B + row * 128 + g * 32, 32);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + row] = acc[r];
  for (int i = l; i < nvals; i += 64) {
    unsigned w = __builtin_amdgcn_cvt_pk_fp8_f32(vals[i], 0.f, 0u, false);
    enc[i] = (uint8_t)(w & 0xff);
  }
}
static float e4m3_to_float(uint8_t v) {   // OCP E4M3 (fn): bias 7, no infinities, 0x7f / 0xff = NaN
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}
int main() {
  uint8_t hA[16 * 128], hB[16 * 128];
  unsigned st = 12345;
  auto rnd = [&] { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (int i = 0; i < 16 * 128; ++i) { hA[i] = (uint8_t)(rnd() % 0x78) | ((rnd() & 1) << 7); hB[i] = (uint8_t)(rnd() % 0x78) | ((rnd() & 1) << 7); }
  const float hv[] = {0.f, 1.f, -1.f, 0.5f, 1.75f, 448.f, 447.f, 460.f, 500.f, 1e-3f, 2.f / 512.f, 0.3f, 3.3f, 240.f, 1e6f};
  const int nv = sizeof(hv) / 4;
  uint8_t *dA, *dB, *dE; float *dD, *dV;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 1024); hipMalloc(&dV, sizeof hv); hipMalloc(&dE, nv);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dV, hv, sizeof hv, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dV, dE, nv);
  float hD[256]; uint8_t hE[64];
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(hE, dE, nv, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double r = 0;
      for (int kk = 0; kk < 128; ++kk) r += (double)e4m3_to_float(hA[i * 128 + kk]) * (double)e4m3_to_float(hB[j * 128 + kk]);
      maxerr = fmax(maxerr, fabs(r - hD[i * 16 + j])); maxref = fmax(maxref, fabs(r));
    }
  printf("mfma 16x16x128 fp8: max |err| %.4g (max |ref| %.4g)  D[0][0]=%.4f D[3][5]=%.4f\n", maxerr, maxref, hD[0], hD[3 * 16 + 5]);
  for (int i = 0; i < nv; ++i) printf("cvt %12.6g -> 0x%02x = %g\n", hv[i], hE[i], e4m3_to_float(hE[i]));
  return 0;
}
