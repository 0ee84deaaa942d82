// Shared device helpers for the SPMM gfx950 kernels (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SPMM_OK 0
#define SPMM_ERR_SHAPE 1
#define SPMM_ERR_LAUNCH 2
#define SPMM_ERR_UNSUPPORTED 3

extern "C" void spmm_set_error(const char* fmt, ...);

#define SPMM_CHECK_SHAPE(cond, ...)                   \
  do {                                                \
    if (!(cond)) {                                    \
      spmm_set_error(__VA_ARGS__);                    \
      return SPMM_ERR_SHAPE;                          \
    }                                                 \
  } while (0)

#define SPMM_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      spmm_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
      return SPMM_ERR_LAUNCH;                                                \
    }                                                                        \
  } while (0)

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ---------------------------------------------------------------------------------------------
// Counter-based dropout RNG.  One 32-bit hash gives two 16-bit uniforms; element e of a tensor
// uses hash(seed, e>>1) half (e&1).  Forward and backward regenerate the same mask from
// (seed, element index), so no mask is ever stored.  p=0 short-circuits at the call sites.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // "lowbias32" finaliser
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}
// One finaliser round over (index xor seed): lowbias32 is built to decorrelate consecutive integers, which is all a
// dropout mask needs; a second round doubled the integer work of the attention kernels for no measurable change in
// the mask statistics (tests/test_kernels_gpu.py::test_attention_dropout_consistency, ::test_layernorm_dropout_masks_match).
// Effective seed of a call site: the per-step seed (a counter that advances by 1) and the per-call-site salt go through two
// splitmix64 rounds BEFORE they meet the element index, so consecutive steps / layers get unrelated masks (xor-ing a counter
// straight into the index would make mask_{n+1}(i) = mask_n(i ^ d)).  Once per thread, not per element.
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t seed_mix(const uint64_t* seed_ptr, uint64_t salt) { return splitmix64(splitmix64(*seed_ptr) + salt); }
__device__ __forceinline__ uint32_t rng_pair(uint64_t seed, uint64_t pair_idx) {
  uint32_t lo = (uint32_t)pair_idx, hi = (uint32_t)(pair_idx >> 32);
  uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  return mix32((lo ^ s0) + (hi ^ s1) * 0x9E3779B1u);
}
// Attention-probability dropout: the element (row, kv) uses half (kv & 1) of drop_pair(drop_rowkey(seed, row), kv >> 1).
// One 32-bit key per query row (hashed once per lane, full lowbias32) and then, per PAIR of elements, a Weyl step and a mixer made of
// full-rate instructions only: x = rowkey + pair * 0x9E3779B1 (the multiple of a compile-time pair index is a literal), two
// v_mul_u32_u24 with xor-shifts by 16 between them (v_mul_lo_u32, which lowbias32 needs twice, is quarter rate: the two of them
// were 8 of the 14 issue slots a pair cost, and the attention kernels are VALU-bound).  The 24-bit multiply drops the top byte of
// its input, which the preceding x ^= x >> 16 has already folded into bits 8-15.  Checked against lowbias32 on 300 000 rows x 64
// pairs and 60 000 x 384 (keep rate, variance of the per-row drop count, keep-bit correlation at lags 1-39 along a row and between
// rows 1-4 apart: all at the sampling noise, 2-4e-4; avalanche 0.499-0.502 per input bit); a single 24-bit multiply is NOT enough
// (lag correlations of 1e-3 to 2e-2 depending on the constant).
constexpr uint32_t DROP_WEYL = 0x9E3779B1u;
__device__ __forceinline__ uint32_t mix24(uint32_t x) {
  x ^= x >> 16; x = __umul24(x, 0xda8f81u); x ^= x >> 16; x = __umul24(x, 0x76dfb5u); x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_rowkey(uint64_t seed, uint64_t row) {
  return mix32((uint32_t)row ^ (uint32_t)seed) ^ (uint32_t)(seed >> 32) ^ ((uint32_t)(row >> 32) * 0x9E3779B1u);
}
__device__ __forceinline__ uint32_t drop_pair(uint32_t rowkey, uint32_t pair) { return mix24(rowkey + pair * DROP_WEYL); }
// keep-decision for element idx; thresh16 = round(p * 65536)
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh16) {
  uint32_t r = rng_pair(seed, idx >> 1);
  uint32_t u = (idx & 1) ? (r >> 16) : (r & 0xffffu);
  return u >= thresh16;
}
// Row-kernel dropout (LayerNorm / embedding kernels): element (row, col) uses half (col & 1) of drop_pair(rowkey, col >> 1) with
// rowkey = drop_rowkey(seed, row) hashed once per row (the 64-bit counter form made the LayerNorm kernels VALU-co-limited).  c % 4 == 0.
__device__ __forceinline__ void drop_keep4(uint32_t rowkey, int c, uint32_t thresh16, bool k[4]) {
  const uint32_t r0 = drop_pair(rowkey, (uint32_t)c >> 1), r1 = drop_pair(rowkey, ((uint32_t)c >> 1) + 1);
  k[0] = (r0 & 0xffffu) >= thresh16; k[1] = (r0 >> 16) >= thresh16;
  k[2] = (r1 & 0xffffu) >= thresh16; k[3] = (r1 >> 16) >= thresh16;
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t idx) {   // [0,1)
  return (float)(rng_pair(seed, idx) >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sums without the LDS crossbar: four DPP steps inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror), one ds_swizzle for the xor-16 step, v_readlane for the two halves.  Every lane of the (half-)wave must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum(float v) {   // over each group of 32 lanes, result in all of them
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {   // over the wave, result wave-uniform (scalar registers)
  v = half_sum(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// (v_rcp_f32 through the builtin: `__frcp_rn` and `1.f / x` expand to the ten-instruction IEEE division sequence.)
// erf by Abramowitz & Stegun 7.1.26 (|abs error| < 1.5e-7, far below the bf16 resolution of the GELU outputs): one v_rcp,
// one v_exp and a 5-term Horner chain instead of libm's erff in the GEMM epilogues.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float y = 1.0f - poly * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// gelu(x) and gelu'(x) together: exp(-x^2/2) serves both the erf (A&S 7.1.26: erf(z) = 1 - poly(t) exp(-z^2), z = x / sqrt 2)
// and the density term, so the derivative costs two extra FMAs in the forward epilogue and the backward epilogue becomes a
// plain multiply (SPMM_EPI_GELU_DERIV / SPMM_EPI_MUL).
// Two elements at a time on float2 values: every multiply / FMA becomes a v_pk_* instruction (two elements per issue slot);
// only the reciprocal, the exponential and the sign transfer stay per element.  Same formula as gelu_erf_both.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool DERIV>
__device__ __forceinline__ void gelu_erf_pair(f32x2 x, f32x2& g, f32x2& dg) {
  const f32x2 ax = __builtin_elementwise_abs(x);
  const f32x2 den = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
  f32x2 t;
  t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  const f32x2 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const f32x2 a = x * x * (-0.5f * 1.4426950408889634f);
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);
  const f32x2 h = 0.5f - 0.5f * (poly * e);                       // = Phi(|x|) - 1/2  in [0, 1/2]
  f32x2 hs;
  hs.x = copysignf(h.x, x.x); hs.y = copysignf(h.y, x.y);
  const f32x2 cdf = hs + 0.5f;
  g = x * cdf;
  if (DERIV) dg = cdf + x * (0.3989422804014327f * e);
}
__device__ __forceinline__ void gelu_erf_both(float x, float& g, float& dg) {
  const float az = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-0.5f * x * x);
  const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
  g = x * cdf;
  dg = cdf + x * (0.3989422804014327f * e);
}

__device__ __forceinline__ bf16x4 to_bf16x4(float a, float b, float c, float d) {
  bf16x4 r; r[0] = (bf16)a; r[1] = (bf16)b; r[2] = (bf16)c; r[3] = (bf16)d; return r;
}
