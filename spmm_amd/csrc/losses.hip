// Loss-side kernels of the SPMM pretraining step for gfx950 (all fp32 math, one wave or one workgroup per row).
//
//   l2norm_*        F.normalize(proj(cls)) SPMM_models.py:92,95,101,105 (+ bf16 hi/lo split for the sim GEMMs)
//   ita_rows        soft-target contrastive loss + its gradient, SPMM_models.py:113-131
//   sample_neg      hard-negative sampling, SPMM_models.py:154-178 (one inverse-CDF draw per row, no host sync)
//   lm_loss         next-token CE (PAD targets included) + momentum distillation, SPMM_models.py:233-238
//   itm_head        itm_head Linear(2H,2) + cross entropy, SPMM_models.py:201-206 (fwd and bwd in one pass)
//   mpm_head        property_mtr_head final Linear(H,1) + masked MSE (x5), SPMM_models.py:251-256
//   enqueue         _dequeue_and_enqueue SPMM_models.py:272-286 (also maintains the bf16 GEMM shadows of the queue)
//
// Every loss kernel adds its (already normalised) contribution into a small device array `losses`
// with atomicAdd; gradients are scaled by a device scalar so the whole step is graph-replayable.
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {   // blockDim = 256
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ---------------------------------------------------------------- l2 normalise (one wave per row)
// y = x / max(||x||, 1e-12); optionally emit the split-bf16 GEMM operand row [hi | lo | hi] (a3, for the A side)
// or [hi | hi | lo] (w3, for the W side) so that a3 . w3 = hi*hi + lo*hi + hi*lo ~ fp32 product.
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, float* __restrict__ nrm,
                                  bf16* __restrict__ a3, bf16* __restrict__ w3, bf16* __restrict__ yT, long ldt, int rows, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float ss = 0.f;
  for (int c = lane; c < E; c += 64) { const float v = x[(long)row * ldx + c]; ss += v * v; }
  const float n = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
  if (lane == 0 && nrm) nrm[row] = n;
  for (int c = lane; c < E; c += 64) {
    const float v = x[(long)row * ldx + c] / n;
    y[(long)row * E + c] = v;
    const bf16 hi = (bf16)v;
    const bf16 lo = (bf16)(v - (float)hi);
    if (a3) { a3[(long)row * 3 * E + c] = hi; a3[(long)row * 3 * E + E + c] = lo; a3[(long)row * 3 * E + 2 * E + c] = hi; }
    if (w3) { w3[(long)row * 3 * E + c] = hi; w3[(long)row * 3 * E + E + c] = hi; w3[(long)row * 3 * E + 2 * E + c] = lo; }
    if (yT) yT[(long)c * ldt + row] = hi;
  }
}
// dx = gscale * (dy - y * <y, dy>) / n   (bf16 out for the projection dgrad/wgrad GEMMs)
__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ nrm,
                                  const float* __restrict__ gscale, bf16* __restrict__ dx, int rows, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < E; c += 64) dot += y[(long)row * E + c] * dy[(long)row * E + c];
  dot = wave_sum(dot);
  const float s = (gscale ? *gscale : 1.f) / nrm[row];
  for (int c = lane; c < E; c += 64) dx[(long)row * E + c] = (bf16)((dy[(long)row * E + c] - y[(long)row * E + c] * dot) * s);
}

// ---------------------------------------------------------------- ITA rows
// S  [2B, J] fp32 : student sims of one feature bank (rows 0..B-1 and B..2B-1 are two different losses)
// SM [2B, J] fp32 : teacher sims in the same order.   J = B + Q real columns; ldj >= J row stride.
// loss_row = lse(s) - alpha * sum_j softmax(sm)_j s_j - (1-alpha) * s[diag]       (sum_j target_j = 1)
// dS_j     = (softmax(s)_j - alpha softmax(sm)_j - (1-alpha) [j==diag]) * rscale  with rscale = 1/(2B)
// dtemp   += sum_j dS_j * (-s_j / temp)     (s = raw/temp)
__global__ __launch_bounds__(256) void ita_rows_kernel(const float* __restrict__ S, const float* __restrict__ SM, long ldj, int B,
                                                       int J, const float* __restrict__ alpha_ptr, const float* __restrict__ temp_ptr,
                                                       bf16* __restrict__ dS, long ldd, int Jpad, float* __restrict__ losses,
                                                       int loss_slot, float* __restrict__ dtemp, int* __restrict__ nan_flag) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  const int diag = row % B;
  const float* s = S + (long)row * ldj;
  const float* sm = SM + (long)row * ldj;
  const float alpha = *alpha_ptr;
  // (one workgroup per row and only 2B rows: the three passes are latency-bound, so they read 16 bytes per lane where the row
  //  stride allows it -- J4 = the part of the row covered by whole float4's)
  const int J4 = (ldj % 4 == 0 && ((uintptr_t)S % 16 == 0) && ((uintptr_t)SM % 16 == 0)) ? (J & ~3) : 0;
  float mx = -INFINITY, mxm = -INFINITY;
  for (int j = threadIdx.x * 4; j < J4; j += 1024) {
    const f32x4 a = *(const f32x4*)(s + j), b = *(const f32x4*)(sm + j);
    mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
    mxm = fmaxf(fmaxf(mxm, fmaxf(b[0], b[1])), fmaxf(b[2], b[3]));
  }
  for (int j = J4 + threadIdx.x; j < J; j += 256) { mx = fmaxf(mx, s[j]); mxm = fmaxf(mxm, sm[j]); }
  mx = block_max(mx, sh);
  mxm = block_max(mxm, sh);
  float se = 0.f, sem = 0.f, dot = 0.f;
  for (int j = threadIdx.x * 4; j < J4; j += 1024) {
    const f32x4 a = *(const f32x4*)(s + j), b = *(const f32x4*)(sm + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float em = __expf(b[e] - mxm);
      se += __expf(a[e] - mx);
      sem += em;
      dot += em * a[e];
    }
  }
  for (int j = J4 + threadIdx.x; j < J; j += 256) {
    const float v = s[j];
    const float em = __expf(sm[j] - mxm);
    se += __expf(v - mx);
    sem += em;
    dot += em * v;
  }
  se = block_sum(se, sh);
  sem = block_sum(sem, sh);
  dot = block_sum(dot, sh);
  const float lse = mx + __logf(se);
  const float loss = lse - alpha * dot / sem - (1.f - alpha) * s[diag];
  const float rscale = 0.5f / B;
  const float inv_se = 1.f / se, inv_sem = alpha / sem;
  float dt = 0.f;
  bf16* d = dS + (long)row * ldd;
  const int J4d = (J4 && ldd % 4 == 0 && ((uintptr_t)dS % 8 == 0)) ? J4 : 0;
  for (int j = threadIdx.x * 4; j < J4d; j += 1024) {
    const f32x4 a = *(const f32x4*)(s + j), b = *(const f32x4*)(sm + j);
    float g4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g4[e] = (__expf(a[e] - mx) * inv_se - __expf(b[e] - mxm) * inv_sem - (j + e == diag ? 1.f - alpha : 0.f)) * rscale;
      dt += g4[e] * a[e];
    }
    *(bf16x4*)(d + j) = to_bf16x4(g4[0], g4[1], g4[2], g4[3]);
  }
  for (int j = J4d + threadIdx.x; j < Jpad; j += 256) {
    float g = 0.f;
    if (j < J) {
      const float v = s[j];
      g = (__expf(v - mx) * inv_se - __expf(sm[j] - mxm) * inv_sem - (j == diag ? 1.f - alpha : 0.f)) * rscale;
      dt += g * v;
    }
    d[j] = (bf16)g;
  }
  dt = block_sum(dt, sh);
  if (threadIdx.x == 0) {
    atomicAdd(losses + loss_slot, loss * rscale);
    atomicAdd(dtemp, -dt / *temp_ptr);
    if (!(loss == loss) && nan_flag) *nan_flag = 1;
  }
}

// ---------------------------------------------------------------- hard negatives
// weights = softmax(S[row, :B]) with the diagonal zeroed; one draw per row by inverse CDF (one wave per row).
// If `forced` is given its indices are copied instead (parity replay of the reference's recorded draws).
__global__ void sample_neg_kernel(const float* __restrict__ S, long ldj, int B, const long* __restrict__ forced,
                                  const uint64_t* __restrict__ seed_ptr, uint64_t salt, long* __restrict__ out, long out_offset) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  if (forced) {
    if (lane == 0) out[row] = forced[row] + out_offset;
    return;
  }
  const float* s = S + (long)row * ldj;
  float mx = -INFINITY;
  for (int j = lane; j < B; j += 64) mx = fmaxf(mx, s[j]);
  mx = wave_max(mx);
  float tot = 0.f;
  for (int j = lane; j < B; j += 64) tot += (j == row) ? 0.f : __expf(s[j] - mx);
  tot = wave_sum(tot);
  const float target = rng_uniform(seed_mix(seed_ptr, salt), (uint64_t)row) * tot;
  // sequential scan in chunks of 64 with a wave prefix sum
  float run = 0.f;
  int pick = -1;
  for (int j0 = 0; j0 < B && pick < 0; j0 += 64) {
    const int j = j0 + lane;
    const float w = (j < B && j != row) ? __expf(s[j] - mx) : 0.f;
    float pre = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(pre, o, 64); if (lane >= o) pre += t; }
    const bool hit = (w > 0.f) && (run + pre > target);
    const unsigned long long m = __ballot(hit);
    if (m) pick = j0 + __ffsll((long long)m) - 1;
    run += __shfl(pre, 63, 64);
  }
  if (pick < 0) {   // rounding left the target beyond the last bucket: take the last non-diagonal index
    pick = (row == B - 1) ? B - 2 : B - 1;
    if (pick < 0) pick = 0;
  }
  if (lane == 0) out[row] = pick + out_offset;
}

// ---------------------------------------------------------------- LM loss (one wave per token row)
// logits / logits_m: [nseq*L, ldl] fp32 (V real columns).  Position t predicts ids[b, t+1]; t = L-1 has no label.
// loss = (1-alpha) * mean_all CE + alpha * mean_{label != 0} distill.   dlogits (bf16, [rows, ldd], zero padded to Vpad).
__global__ __launch_bounds__(1024) void count_nonpad_kernel(const int* __restrict__ ids, long nseq, int L, int* __restrict__ out) {
  __shared__ float sh[16];                             // one workgroup of 16 waves (the count is needed before lm_loss starts)
  float c = 0.f;
  for (long i = threadIdx.x; i < nseq * L; i += 1024) {
    const int t = (int)(i % L);
    if (t >= 1 && ids[i] != 0) c += 1.f;
  }
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sh[w];
    *out = (int)(t + 0.5f);
  }
}
// One wave per row, workgroups stride over the rows and add ONE loss atomic each: 16 384 same-address atomics (one per row) serialise
// at ~13 ns apiece -- 213 us of a 215-us launch, measured.
__global__ __launch_bounds__(256) void lm_loss_kernel(const float* __restrict__ logits, const float* __restrict__ logits_m, long ldl,
                               const int* __restrict__ ids, long nseq, int L, int V, const float* __restrict__ alpha_ptr,
                               const int* __restrict__ n_nonpad, const float* __restrict__ gscale, bf16* __restrict__ dlogits,
                               long ldd, int Vpad, float* __restrict__ losses, int loss_slot) {
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float alpha = *alpha_ptr;
  const float n_all = (float)(nseq * (L - 1));
  const float n_nz = (float)max(*n_nonpad, 1);
  const float w_ce = (1.f - alpha) / n_all;
  const float g = gscale ? *gscale : 1.f;
  float lacc = 0.f;
  for (long row = (long)blockIdx.x * 4 + wave; row < nseq * L; row += (long)gridDim.x * 4) {
    const int t = (int)(row % L);
    bf16* d = dlogits ? dlogits + row * ldd : nullptr;
    if (t == L - 1) {
      if (d) for (int j = lane; j < Vpad; j += 64) d[j] = (bf16)0.f;
      continue;
    }
    const int label = ids[row + 1];
    const float* x = logits + row * ldl;
    const float* xm = logits_m + row * ldl;
    float mx = -INFINITY, mxm = -INFINITY;
    for (int j = lane; j < V; j += 64) { mx = fmaxf(mx, x[j]); mxm = fmaxf(mxm, xm[j]); }
    mx = wave_max(mx);
    mxm = wave_max(mxm);
    float se = 0.f, sem = 0.f, dot = 0.f;
    for (int j = lane; j < V; j += 64) {
      const float em = __expf(xm[j] - mxm);
      se += __expf(x[j] - mx);
      sem += em;
      dot += em * x[j];
    }
    se = wave_sum(se); sem = wave_sum(sem); dot = wave_sum(dot);
    const float lse = mx + __logf(se);
    const float w_ds = (label != 0) ? alpha / n_nz : 0.f;
    lacc += w_ce * (lse - x[label]) + w_ds * (lse - dot / sem);          // (wave-uniform)
    if (d) {
      for (int j = lane; j < Vpad; j += 64) {
        float v = 0.f;
        if (j < V) {
          const float sp = __expf(x[j] - mx) / se;
          v = g * (w_ce * (sp - (j == label ? 1.f : 0.f)) + w_ds * (sp - __expf(xm[j] - mxm) / sem));
        }
        d[j] = (bf16)v;
      }
    }
  }
  if (lane == 0) sh[wave] = lacc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(losses + loss_slot, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// ---------------------------------------------------------------- ITM head fwd+bwd (one wave per pair row)
// vl[i] = [xa[rowa(i)] | xb[rowb(i)]] (bf16 rows of width H), logits = vl W^T + b (W [2, 2H] fp32), CE vs label
// (1 for i < B else 0), mean over n = 3B rows.  Writes d xa / d xb rows (bf16) and accumulates dW, db.
template <typename TX>   // bf16, or float with the fp32 residual stream (EngineOptions.resid_fp32)
__global__ void itm_head_kernel(const TX* __restrict__ xa, long stride_a, const TX* __restrict__ xb, long stride_b, int H,
                                const float* __restrict__ W, const float* __restrict__ bias, int n, int B,
                                const float* __restrict__ gscale, float* __restrict__ losses, int loss_slot,
                                float* __restrict__ logits_out, bf16* __restrict__ dxa, bf16* __restrict__ dxb,
                                float* __restrict__ dW, float* __restrict__ db, int do_bwd) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const TX* a = xa + (long)i * stride_a;
  const TX* b = xb + (long)i * stride_b;
  float l0 = 0.f, l1 = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float va = (float)a[c], vb = (float)b[c];
    l0 += va * W[c] + vb * W[H + c];
    l1 += va * W[2 * H + c] + vb * W[3 * H + c];
  }
  l0 = wave_sum(l0) + bias[0];
  l1 = wave_sum(l1) + bias[1];
  const int label = i < B ? 1 : 0;
  const float mx = fmaxf(l0, l1);
  const float lse = mx + __logf(__expf(l0 - mx) + __expf(l1 - mx));
  if (lane == 0) {
    atomicAdd(losses + loss_slot, (lse - (label ? l1 : l0)) / n);
    if (logits_out) { logits_out[2 * i] = l0; logits_out[2 * i + 1] = l1; }
  }
  if (!do_bwd) return;
  const float g = (gscale ? *gscale : 1.f) / n;
  const float d0 = g * (__expf(l0 - lse) - (label ? 0.f : 1.f));
  const float d1 = g * (__expf(l1 - lse) - (label ? 1.f : 0.f));
  bf16* da = dxa + (long)i * stride_a;
  bf16* dbb = dxb + (long)i * stride_b;
  for (int c = lane; c < H; c += 64) {
    const float va = (float)a[c], vb = (float)b[c];
    da[c] = (bf16)(d0 * W[c] + d1 * W[2 * H + c]);
    dbb[c] = (bf16)(d0 * W[H + c] + d1 * W[3 * H + c]);
    atomicAdd(dW + c, d0 * va);
    atomicAdd(dW + H + c, d0 * vb);
    atomicAdd(dW + 2 * H + c, d1 * va);
    atomicAdd(dW + 3 * H + c, d1 * vb);
  }
  if (lane == 0) { atomicAdd(db, d0); atomicAdd(db + 1, d1); }
}

// ---------------------------------------------------------------- MPM head: pred = h . w + b ; masked MSE * 5
// h: [B*(Lp), H] bf16 rows of the LayerNorm'ed head activations laid out per sequence of Lp tokens; positions
// 0..Lp-2 predict property 0..Lp-2 (the [:, :-1, :] slice, SPMM_models.py:250).  mask: 1 = masked (excluded).
__global__ void count_keep_kernel(const float* __restrict__ mask, long n, int* __restrict__ out) {
  __shared__ float sh[4];
  float c = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) c += (mask[i] == 0.f) ? 1.f : 0.f;
  c = block_sum(c, sh);
  if (threadIdx.x == 0) *out = (int)(c + 0.5f);
}
// Workgroups stride over the rows; the loss, db and the H columns of dw are accumulated per wave in registers and leave as one
// atomic per workgroup (per column): one atomic per ROW on the same addresses serialised the launch (~13 ns each).
template <typename TX>
__global__ __launch_bounds__(256) void mpm_head_kernel(const TX* __restrict__ h, int Lp, int H, const float* __restrict__ w, const float* __restrict__ bias,
                                const float* __restrict__ target, const float* __restrict__ mask, int B, const int* __restrict__ n_keep,
                                const float* __restrict__ gscale, float* __restrict__ losses, int loss_slot, float* __restrict__ pred_out,
                                bf16* __restrict__ dh, float* __restrict__ dw, float* __restrict__ db, int do_bwd) {
  constexpr int MAXC = 16;                               // H <= 1024
  __shared__ float red[4][1024 + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float nk = (float)max(*n_keep, 1);
  const float gsc = gscale ? *gscale : 1.f;
  float lacc = 0.f, dbacc = 0.f, dwacc[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) dwacc[i] = 0.f;
  for (int r = blockIdx.x * 4 + wave; r < B * Lp; r += gridDim.x * 4) {   // r in [0, B*Lp)
    const int b = r / Lp, i = r - b * Lp;
    const TX* x = h + (long)r * H;
    bf16* d = dh ? dh + (long)r * H : nullptr;
    if (i == Lp - 1) {
      if (do_bwd && d) for (int c = lane; c < H; c += 64) d[c] = (bf16)0.f;
      continue;
    }
    float p = 0.f;
    for (int c = lane; c < H; c += 64) p += (float)x[c] * w[c];
    p = wave_sum(p) + bias[0];
    const long ti = (long)b * (Lp - 1) + i;
    const bool keep = mask[ti] == 0.f;
    const float err = p - target[ti];
    if (lane == 0 && pred_out) pred_out[ti] = p;
    if (keep) lacc += 5.f * err * err / nk;              // (wave-uniform)
    if (!do_bwd) continue;
    const float g = keep ? gsc * 10.f * err / nk : 0.f;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < H) {
        d[c] = (bf16)(g * w[c]);
        dwacc[k] += g * (float)x[c];
      }
    }
    dbacc += g;
  }
  // block reduction: loss, db, dw columns
#pragma unroll
  for (int k = 0; k < MAXC; ++k) { const int c = lane + 64 * k; if (c < H) red[wave][c] = dwacc[k]; }
  if (lane == 0) { red[wave][1024] = lacc; red[wave][1025] = dbacc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(losses + loss_slot, (red[0][1024] + red[1][1024]) + (red[2][1024] + red[3][1024]));
    if (do_bwd) atomicAdd(db, (red[0][1025] + red[1][1025]) + (red[2][1025] + red[3][1025]));
  }
  if (do_bwd)
    for (int c = threadIdx.x; c < H; c += 256) atomicAdd(dw + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
}

// ---------------------------------------------------------------- small dense heads for the inference tier
// out[r, n] = act(bias[n] + sum_k x[r, k] W[n, k]), fp32 weights / accumulation / output, x fp32 or bf16.  One wave per row.
// K >= 64: the lanes stride k (coalesced x and W rows) and every output is one wave reduction -- the projection, matching and
// regression heads (N in {1, 2, 256, H}, K = H or 2H) called on a handful of pooled rows (SPMM_models.py:36-43).  K < 64: the lanes
// stride n -- property_embed = Linear(1, H) (:36).
template <bool XBF>
__global__ __launch_bounds__(256) void rows_linear_kernel(const void* __restrict__ xv, long ldx, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                                          long rows, int N, int K, int act) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  auto X = [&](int k) -> float { return XBF ? (float)((const bf16*)xv)[r * ldx + k] : ((const float*)xv)[r * ldx + k]; };
  auto fin = [&](float v) -> float { return act == 1 ? 0.5f * v * (1.f + erff(v * 0.70710678118654752f)) : v; };
  if (K >= 64) {
    for (int n = 0; n < N; ++n) {
      float a = 0.f;
      for (int k = lane; k < K; k += 64) a += X(k) * W[(long)n * K + k];
      a = wave_sum(a);
      if (lane == 0) out[r * ldo + n] = fin(a + (bias ? bias[n] : 0.f));
    }
  } else {
    for (int n = lane; n < N; n += 64) {
      float a = bias ? bias[n] : 0.f;
      for (int k = 0; k < K; ++k) a += X(k) * W[(long)n * K + k];
      out[r * ldo + n] = fin(a);
    }
  }
}

// ---------------------------------------------------------------- queue
// feats [n, E] fp32 (all-gathered momentum features).  queue [E, Q] fp32 (reference layout, state_dict buffer),
// w3 [B + Q, 3E] bf16 split rows (sim-GEMM W operand; queue column j lives at row B + j),
// qT [E, ldt] bf16 (dfeat-GEMM W operand; queue column j lives at column B + j).  ptr advances by n mod Q.
__global__ void enqueue_kernel(const float* __restrict__ feats, int n, int E, float* __restrict__ queue, int Q,
                               bf16* __restrict__ w3, bf16* __restrict__ qT, long ldt, int Bloc, long* __restrict__ ptr,
                               const int* __restrict__ skip) {
  if (skip && *skip) return;                 // NaN step: the reference returns before _dequeue_and_enqueue (SPMM_models.py:132-134, :208)
  const long p0 = *ptr;
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const long col = (p0 + i) % Q;
    for (int c = threadIdx.x; c < E; c += blockDim.x) {
      const float v = feats[(long)i * E + c];
      queue[(long)c * Q + col] = v;
      const bf16 hi = (bf16)v;
      const bf16 lo = (bf16)(v - (float)hi);
      bf16* r = w3 + (Bloc + col) * 3L * E;
      r[c] = hi; r[E + c] = hi; r[2 * E + c] = lo;
      qT[(long)c * ldt + Bloc + col] = hi;
    }
  }
}
__global__ void advance_ptr_kernel(long* ptr, int n, int Q, const int* skip) {
  if (skip && *skip) return;
  *ptr = (*ptr + n) % Q;
}

// rebuild the bf16 shadows of the whole queue from the fp32 master (after load_state_dict)
__global__ void queue_shadow_kernel(const float* __restrict__ queue, int E, int Q, bf16* __restrict__ w3, bf16* __restrict__ qT,
                                    long ldt, int Bloc) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)E * Q; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i / Q);
    const long col = i - (long)c * Q;
    const float v = queue[i];
    const bf16 hi = (bf16)v;
    const bf16 lo = (bf16)(v - (float)hi);
    bf16* r = w3 + (Bloc + col) * 3L * E;
    r[c] = hi; r[E + c] = hi; r[2 * E + c] = lo;
    qT[(long)c * ldt + Bloc + col] = hi;
  }
}

__global__ void clamp_scalar_kernel(float* p, float lo, float hi) { *p = fminf(fmaxf(*p, lo), hi); }

}  // namespace

extern "C" int spmm_l2norm_fwd(const float* x, long ldx, float* y, float* nrm, void* a3, void* w3, void* yT, long ldt, int rows, int E,
                               hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && E > 0, "spmm_l2norm_fwd: rows=%d E=%d", rows, E);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, y, nrm, (bf16*)a3, (bf16*)w3, (bf16*)yT, ldt, rows, E);
  SPMM_LAUNCH_CHECK("spmm_l2norm_fwd");
  return SPMM_OK;
}
extern "C" int spmm_l2norm_bwd(const float* dy, const float* y, const float* nrm, const float* gscale, void* dx, int rows, int E,
                               hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && E > 0, "spmm_l2norm_bwd: rows=%d E=%d", rows, E);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, dy, y, nrm, gscale, (bf16*)dx, rows, E);
  SPMM_LAUNCH_CHECK("spmm_l2norm_bwd");
  return SPMM_OK;
}
extern "C" int spmm_ita_rows(const float* S, const float* SM, long ldj, int nrows, int B, int J, const float* alpha_ptr,
                             const float* temp_ptr, void* dS, long ldd, int Jpad, float* losses, int loss_slot, float* dtemp,
                             int* nan_flag, hipStream_t stream) {
  SPMM_CHECK_SHAPE(nrows > 0 && B > 0 && J >= B && Jpad >= J && ldd >= Jpad, "spmm_ita_rows: nrows=%d B=%d J=%d Jpad=%d", nrows, B, J, Jpad);
  hipLaunchKernelGGL(ita_rows_kernel, dim3(nrows), dim3(256), 0, stream, S, SM, ldj, B, J, alpha_ptr, temp_ptr, (bf16*)dS, ldd, Jpad,
                     losses, loss_slot, dtemp, nan_flag);
  SPMM_LAUNCH_CHECK("spmm_ita_rows");
  return SPMM_OK;
}
extern "C" int spmm_sample_neg(const float* S, long ldj, int B, const long* forced, const uint64_t* seed_ptr, uint64_t salt,
                               long* out, long out_offset, hipStream_t stream) {
  SPMM_CHECK_SHAPE(B > 1, "spmm_sample_neg: needs B >= 2 (B=%d)", B);
  SPMM_CHECK_SHAPE(forced || seed_ptr, "spmm_sample_neg: needs a device seed or forced indices");
  hipLaunchKernelGGL(sample_neg_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, S, ldj, B, forced, seed_ptr, salt, out, out_offset);
  SPMM_LAUNCH_CHECK("spmm_sample_neg");
  return SPMM_OK;
}
extern "C" int spmm_lm_loss(const float* logits, const float* logits_m, long ldl, const int* ids, long nseq, int L, int V,
                            const float* alpha_ptr, int* n_nonpad_ws, const float* gscale, void* dlogits, long ldd, int Vpad,
                            float* losses, int loss_slot, hipStream_t stream) {
  SPMM_CHECK_SHAPE(nseq > 0 && L > 1 && V > 0 && Vpad >= V, "spmm_lm_loss: nseq=%ld L=%d V=%d Vpad=%d", nseq, L, V, Vpad);
  hipLaunchKernelGGL(count_nonpad_kernel, dim3(1), dim3(1024), 0, stream, ids, nseq, L, n_nonpad_ws);
  hipLaunchKernelGGL(lm_loss_kernel, dim3((nseq * L + 3) / 4 < 1024 ? (nseq * L + 3) / 4 : 1024), dim3(256), 0, stream, logits, logits_m, ldl, ids, nseq, L, V, alpha_ptr,
                     n_nonpad_ws, gscale, (bf16*)dlogits, ldd, Vpad, losses, loss_slot);
  SPMM_LAUNCH_CHECK("spmm_lm_loss");
  return SPMM_OK;
}
extern "C" int spmm_itm_head(const void* xa, long stride_a, const void* xb, long stride_b, int H, const float* W, const float* bias,
                             int n, int B, const float* gscale, float* losses, int loss_slot, float* logits_out, void* dxa, void* dxb,
                             float* dW, float* db, int do_bwd, int x_is_f32, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && H > 0, "spmm_itm_head: n=%d H=%d", n, H);
  SPMM_CHECK_SHAPE(!do_bwd || (dxa && dxb && dW && db), "spmm_itm_head: backward outputs missing");
  if (x_is_f32)
    hipLaunchKernelGGL(itm_head_kernel<float>, dim3((n + 3) / 4), dim3(256), 0, stream, (const float*)xa, stride_a, (const float*)xb, stride_b, H, W,
                       bias, n, B, gscale, losses, loss_slot, logits_out, (bf16*)dxa, (bf16*)dxb, dW, db, do_bwd);
  else
    hipLaunchKernelGGL(itm_head_kernel<bf16>, dim3((n + 3) / 4), dim3(256), 0, stream, (const bf16*)xa, stride_a, (const bf16*)xb, stride_b, H, W,
                       bias, n, B, gscale, losses, loss_slot, logits_out, (bf16*)dxa, (bf16*)dxb, dW, db, do_bwd);
  SPMM_LAUNCH_CHECK("spmm_itm_head");
  return SPMM_OK;
}
extern "C" int spmm_mpm_head(const void* h, int Lp, int H, const float* w, const float* bias, const float* target, const float* mask,
                             int B, int* n_keep_ws, const float* gscale, float* losses, int loss_slot, float* pred_out, void* dh,
                             float* dw, float* db, int do_bwd, int x_is_f32, hipStream_t stream) {
  SPMM_CHECK_SHAPE(B > 0 && Lp > 1 && H > 0 && H <= 1024, "spmm_mpm_head: B=%d Lp=%d H=%d (H <= 1024)", B, Lp, H);
  SPMM_CHECK_SHAPE(!do_bwd || (dh && dw && db), "spmm_mpm_head: backward outputs missing");
  hipLaunchKernelGGL(count_keep_kernel, dim3(1), dim3(256), 0, stream, mask, (long)B * (Lp - 1), n_keep_ws);
  const dim3 mgrid((B * Lp + 3) / 4 < 512 ? (B * Lp + 3) / 4 : 512);
  if (x_is_f32)
    hipLaunchKernelGGL(mpm_head_kernel<float>, mgrid, dim3(256), 0, stream, (const float*)h, Lp, H, w, bias, target, mask, B, n_keep_ws, gscale, losses,
                       loss_slot, pred_out, (bf16*)dh, dw, db, do_bwd);
  else
    hipLaunchKernelGGL(mpm_head_kernel<bf16>, mgrid, dim3(256), 0, stream, (const bf16*)h, Lp, H, w, bias, target, mask, B, n_keep_ws, gscale, losses,
                       loss_slot, pred_out, (bf16*)dh, dw, db, do_bwd);
  SPMM_LAUNCH_CHECK("spmm_mpm_head");
  return SPMM_OK;
}
extern "C" int spmm_rows_linear(const void* x, int x_is_bf16, long ldx, const float* W, const float* bias, float* out, long ldo,
                                long rows, int N, int K, int act, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && N > 0 && K > 0 && ldx >= K && ldo >= N, "spmm_rows_linear: rows=%ld N=%d K=%d ldx=%ld ldo=%ld", rows, N, K, ldx, ldo);
  SPMM_CHECK_SHAPE(act == 0 || act == 1, "spmm_rows_linear: act %d (0 = none, 1 = erf-GELU)", act);
  if (x_is_bf16)
    hipLaunchKernelGGL(rows_linear_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, W, bias, out, ldo, rows, N, K, act);
  else
    hipLaunchKernelGGL(rows_linear_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, W, bias, out, ldo, rows, N, K, act);
  SPMM_LAUNCH_CHECK("spmm_rows_linear");
  return SPMM_OK;
}
extern "C" int spmm_enqueue(const float* feats, int n, int E, float* queue, int Q, void* w3, void* qT, long ldt, int Bloc, long* ptr,
                            int advance, const int* skip_flag, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && E > 0 && Q % n == 0, "spmm_enqueue: queue_size %d %% batch %d != 0 (SPMM_models.py:279)", Q, n);
  hipLaunchKernelGGL(enqueue_kernel, dim3(n < 256 ? n : 256), dim3(256), 0, stream, feats, n, E, queue, Q, (bf16*)w3, (bf16*)qT, ldt, Bloc, ptr, skip_flag);
  if (advance) hipLaunchKernelGGL(advance_ptr_kernel, dim3(1), dim3(1), 0, stream, ptr, n, Q, skip_flag);
  SPMM_LAUNCH_CHECK("spmm_enqueue");
  return SPMM_OK;
}
extern "C" int spmm_queue_shadow(const float* queue, int E, int Q, void* w3, void* qT, long ldt, int Bloc, hipStream_t stream) {
  SPMM_CHECK_SHAPE(E > 0 && Q > 0, "spmm_queue_shadow: E=%d Q=%d", E, Q);
  hipLaunchKernelGGL(queue_shadow_kernel, dim3(1024), dim3(256), 0, stream, queue, E, Q, (bf16*)w3, (bf16*)qT, ldt, Bloc);
  SPMM_LAUNCH_CHECK("spmm_queue_shadow");
  return SPMM_OK;
}
extern "C" int spmm_clamp_scalar(float* p, float lo, float hi, hipStream_t stream) {
  hipLaunchKernelGGL(clamp_scalar_kernel, dim3(1), dim3(1), 0, stream, p, lo, hi);
  SPMM_LAUNCH_CHECK("spmm_clamp_scalar");
  return SPMM_OK;
}
