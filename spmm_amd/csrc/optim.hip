// Multi-tensor optimiser kernels over the flat fp32 parameter arena (gfx950, HBM-bound, 16-byte accesses).
//
//   sqnorm + adamw_prep : torch.nn.utils.clip_grad_norm_(params, 5.) SPMM_models.py:361
//   adamw               : torch.optim.AdamW(lr, weight_decay on ALL params) step, SPMM_models.py:340,362
//                         (decoupled decay, bias correction, eps outside the sqrt -- torch semantics),
//                         fused with the clip scale and the refresh of the bf16 compute shadow
//   ema                 : _momentum_update SPMM_models.py:266-269, fused with the momentum bf16 shadow refresh
// All scalars that change between steps (lr, step count, grad-norm, skip flag) live in device memory so the
// whole step can be replayed from a hipGraph.
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

struct AdamScalars {   // written by adamw_prep_kernel, read by adamw_kernel
  float clip;          // min(1, max_norm / (||g|| + 1e-6))
  float bc1, bc2;      // 1 - beta^t
  float skip;          // 1 -> leave everything untouched (non-finite gradient or NaN-guard step)
  float grad_norm;     // for logging
};

// Deterministic two-stage reduction (fixed summation order): data-parallel replicas hold bit-identical gradients after the
// all-reduce and must derive the SAME clip coefficient, otherwise their parameters drift apart in the last bits.
constexpr int SQN_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n4, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out += (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void adamw_prep_kernel(const float* __restrict__ normsq, float max_norm, float beta1, float beta2,
                                  int* __restrict__ step, const int* __restrict__ nan_flag, AdamScalars* __restrict__ out) {
  const float nsq = *normsq;
  const float norm = sqrtf(nsq);
  const bool bad = !(nsq == nsq) || isinf(nsq) || (nan_flag && *nan_flag);
  int t = *step;
  if (!bad) { t += 1; *step = t; }
  out->grad_norm = norm;
  out->clip = fminf(1.f, max_norm / (norm + 1e-6f));
  out->bc1 = 1.f - powf(beta1, (float)t);
  out->bc2 = 1.f - powf(beta2, (float)t);
  out->skip = bad ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16* __restrict__ shadow, long n4,
                                                    const float* __restrict__ lr_ptr, float beta1, float beta2, float eps, float wd,
                                                    const AdamScalars* __restrict__ sc) {
  if (sc->skip != 0.f) return;
  const float lr = *lr_ptr, clip = sc->clip;
  const float step_size = lr / sc->bc1, inv_sqrt_bc2 = rsqrtf(sc->bc2), decay = 1.f - lr * wd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 pp = *(f32x4*)(p + i * 4), mm = *(f32x4*)(m + i * 4), vv = *(f32x4*)(v + i * 4);
    const f32x4 gg = *(const f32x4*)(g + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * clip;
      pp[j] *= decay;
      mm[j] = beta1 * mm[j] + (1.f - beta1) * gr;
      vv[j] = beta2 * vv[j] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pp[j] -= step_size * (mm[j] / denom);
    }
    *(f32x4*)(p + i * 4) = pp; *(f32x4*)(m + i * 4) = mm; *(f32x4*)(v + i * 4) = vv;
    if (shadow) *(bf16x4*)(shadow + i * 4) = to_bf16x4(pp[0], pp[1], pp[2], pp[3]);
  }
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pm, const float* __restrict__ p, bf16* __restrict__ shadow,
                                                  long n4, float momentum) {
  const float a = momentum, b = 1.f - momentum;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 x = *(f32x4*)(pm + i * 4);
    const f32x4 y = *(const f32x4*)(p + i * 4);
    x[0] = x[0] * a + y[0] * b; x[1] = x[1] * a + y[1] * b; x[2] = x[2] * a + y[2] * b; x[3] = x[3] * a + y[3] * b;
    *(f32x4*)(pm + i * 4) = x;
    if (shadow) *(bf16x4*)(shadow + i * 4) = to_bf16x4(x[0], x[1], x[2], x[3]);
  }
}

__global__ void axpy_scalar_kernel(float* dst, const float* src, const float* scale_ptr, float scale, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i] * scale * (scale_ptr ? *scale_ptr : 1.f);
}

}  // namespace

extern "C" long spmm_adam_scalars_bytes(void) { return sizeof(AdamScalars); }

extern "C" long spmm_grad_sqnorm_workspace_bytes(void) { return SQN_BLOCKS * sizeof(float); }

extern "C" int spmm_grad_sqnorm(const float* g, long n, float* out_zeroed, float* workspace, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && n % 4 == 0, "spmm_grad_sqnorm: n=%ld must be a positive multiple of 4", n);
  SPMM_CHECK_SHAPE(workspace != nullptr, "spmm_grad_sqnorm: needs a workspace of spmm_grad_sqnorm_workspace_bytes()");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > SQN_BLOCKS) blocks = SQN_BLOCKS;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, stream, g, n / 4, workspace);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, stream, workspace, (int)blocks, out_zeroed);
  SPMM_LAUNCH_CHECK("spmm_grad_sqnorm");
  return SPMM_OK;
}

extern "C" int spmm_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, const float* lr_ptr,
                               float beta1, float beta2, float eps, float weight_decay, const float* normsq, float max_norm,
                               int* step, const int* nan_flag, void* scalars_ws, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && n % 4 == 0, "spmm_adamw_step: n=%ld must be a positive multiple of 4", n);
  hipLaunchKernelGGL(adamw_prep_kernel, dim3(1), dim3(1), 0, stream, normsq, max_norm, beta1, beta2, step, nan_flag, (AdamScalars*)scalars_ws);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, stream, p, g, m, v, (bf16*)bf16_shadow, n / 4, lr_ptr, beta1, beta2, eps,
                     weight_decay, (const AdamScalars*)scalars_ws);
  SPMM_LAUNCH_CHECK("spmm_adamw_step");
  return SPMM_OK;
}

extern "C" int spmm_ema_update(float* pm, const float* p, void* bf16_shadow, long n, float momentum, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && n % 4 == 0, "spmm_ema_update: n=%ld must be a positive multiple of 4", n);
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ema_kernel, dim3(blocks), dim3(256), 0, stream, pm, p, (bf16*)bf16_shadow, n / 4, momentum);
  SPMM_LAUNCH_CHECK("spmm_ema_update");
  return SPMM_OK;
}

extern "C" int spmm_axpy_scalar(float* dst, const float* src, const float* scale_ptr, float scale, int n, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0, "spmm_axpy_scalar: n=%d", n);
  hipLaunchKernelGGL(axpy_scalar_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dst, src, scale_ptr, scale, n);
  SPMM_LAUNCH_CHECK("spmm_axpy_scalar");
  return SPMM_OK;
}
