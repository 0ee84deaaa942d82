// HBM-bound row kernels for gfx950: residual+dropout+LayerNorm (fwd/bwd), embeddings+LayerNorm (fwd/bwd),
// PV embedding/masking, bf16 transposes (with fused column sums for bias gradients), casts, gathers.
// One wave64 per row, 8-byte bf16 vector accesses, fp32 math, statistics via wave shuffles.
//
// Reference ops replaced: BertSelfOutput / BertOutput `LayerNorm(dropout(dense) + input)` xbert.py:369-373,
// :447-451; BertEmbeddings.forward :193-220; SPMM.forward PV embed/mask/CLS concat SPMM_models.py:82-88;
// BertPredictionHeadTransform LayerNorm :675; property_mtr_head LayerNorm SPMM_models.py:41.
#include "common.h"
#include <stdlib.h>
#include "../../include/spmm_hip.h"

namespace {

constexpr int MAXC = 4;   // 4-element chunks per lane: H <= 1024

struct RowVec {
  float v[MAXC][4];
};

__device__ __forceinline__ void load_row_bf16(const bf16* __restrict__ p, int H, int lane, RowVec& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const bf16x4 t = *(const bf16x4*)(p + c);
      r.v[i][0] = (float)t[0]; r.v[i][1] = (float)t[1]; r.v[i][2] = (float)t[2]; r.v[i][3] = (float)t[3];
    } else {
      r.v[i][0] = r.v[i][1] = r.v[i][2] = r.v[i][3] = 0.f;
    }
  }
}
__device__ __forceinline__ void store_row_bf16(bf16* __restrict__ p, int H, int lane, const RowVec& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) *(bf16x4*)(p + c) = to_bf16x4(r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]);
  }
}
__device__ __forceinline__ void load_row_f32(const float* __restrict__ p, int H, int lane, RowVec& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const f32x4 t = *(const f32x4*)(p + c);
      r.v[i][0] = t[0]; r.v[i][1] = t[1]; r.v[i][2] = t[2]; r.v[i][3] = t[3];
    } else {
      r.v[i][0] = r.v[i][1] = r.v[i][2] = r.v[i][3] = 0.f;
    }
  }
}
__device__ __forceinline__ float row_sum(const RowVec& r) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) s += (r.v[i][0] + r.v[i][1]) + (r.v[i][2] + r.v[i][3]);
  return wave_sum(s);
}

// normalise z in place given gamma/beta -> y; returns mean / rstd
__device__ __forceinline__ void ln_apply(RowVec& z, int H, int lane, const float* gamma, const float* beta, float eps,
                                         float& mean, float& rstd, RowVec& y) {
  mean = row_sum(z) / H;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = z.v[i][j] - mean; ss += d * d; }
    }
  }
  const float var = wave_sum(ss) / H;
  rstd = rsqrtf(var + eps);
  if (!(var + eps > 0.f)) rstd = 0.f;   // all-equal row with eps=1e-12 underflow guard (torch gives 0*inf-free result too)
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const f32x4 gm = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) y.v[i][j] = (z.v[i][j] - mean) * rstd * gm[j] + bt[j];
    }
  }
}

// ---------------------------------------------------------------- y = LN(dropout(x) + res)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16* __restrict__ y, bf16* __restrict__ zout, float* __restrict__ mean_o,
                                                     float* __restrict__ rstd_o, long rows, int H, float eps,
                                                     uint32_t thresh16, float dscale, const uint64_t* seed_ptr, uint64_t salt,
                                                     const int* __restrict__ rows_ptr) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_ptr) { const long r_ = *rows_ptr; rows = r_ < rows ? r_ : rows; }
  if (row >= rows) return;
  RowVec z, r, o;
  load_row_bf16(x + row * H, H, lane, z);
  if (thresh16) {
    const uint32_t rowkey = drop_rowkey(seed_mix(seed_ptr, salt), (uint64_t)row);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        bool k[4];
        drop_keep4(rowkey, c, thresh16, k);
#pragma unroll
        for (int j = 0; j < 4; ++j) z.v[i][j] = k[j] ? z.v[i][j] * dscale : 0.f;
      }
    }
  }
  if (res) {
    load_row_bf16(res + row * H, H, lane, r);
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) z.v[i][j] += r.v[i][j];
  }
  float mean, rstd;
  ln_apply(z, H, lane, gamma, beta, eps, mean, rstd, o);
  store_row_bf16(y + row * H, H, lane, o);
  if (zout) store_row_bf16(zout + row * H, H, lane, z);
  if (mean_o && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// fp32 residual stream (EngineOptions.resid_fp32): the residual arrives in fp32 and the normalised row leaves twice -- bf16 for the
// GEMMs that read it as an MFMA operand, fp32 for the next residual addition and the loss heads.  One wave per row, any H <= 1024.
__device__ __forceinline__ void store_row_f32(float* __restrict__ p, int H, int lane, const RowVec& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) { f32x4 t = {r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]}; *(f32x4*)(p + c) = t; }
  }
}
__global__ __launch_bounds__(256) void ln_fwd_r32_kernel(const bf16* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16* __restrict__ y, float* __restrict__ y32, bf16* __restrict__ zout,
                                                         float* __restrict__ mean_o, float* __restrict__ rstd_o, long rows, int H, float eps,
                                                         uint32_t thresh16, float dscale, const uint64_t* seed_ptr, uint64_t salt) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  RowVec z, r, o;
  load_row_bf16(x + row * H, H, lane, z);
  if (thresh16) {
    const uint32_t rowkey = drop_rowkey(seed_mix(seed_ptr, salt), (uint64_t)row);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        bool k[4];
        drop_keep4(rowkey, c, thresh16, k);
#pragma unroll
        for (int j = 0; j < 4; ++j) z.v[i][j] = k[j] ? z.v[i][j] * dscale : 0.f;
      }
    }
  }
  if (res) {
    load_row_f32(res + row * H, H, lane, r);
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) z.v[i][j] += r.v[i][j];
  }
  float mean, rstd;
  ln_apply(z, H, lane, gamma, beta, eps, mean, rstd, o);
  store_row_bf16(y + row * H, H, lane, o);
  if (y32) store_row_f32(y32 + row * H, H, lane, o);
  if (zout) store_row_bf16(zout + row * H, H, lane, z);
  if (mean_o && lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// The same for H == 256 * NC (768, 1024, ...): a row is owned by HALF a wave (32 lanes x NC chunks of 16 bytes), so a wave
// normalises two rows at once, every access is a full 16-byte lane access and there are no column guards.
template <int NC>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16* __restrict__ y, bf16* __restrict__ zout, float* __restrict__ mean_o,
                                                       float* __restrict__ rstd_o, long rows, float eps, uint32_t thresh16,
                                                       float dscale, const uint64_t* seed_ptr, uint64_t salt, const int* __restrict__ rows_ptr) {
  constexpr int H = 256 * NC;
  const int l32 = threadIdx.x & 31;
  if (rows_ptr) {                                  // device-side row count (<= rows): workgroups wholly past it leave at once
    const long r_ = *rows_ptr;
    rows = r_ < rows ? (r_ > 0 ? r_ : 1) : rows;
    if ((long)blockIdx.x * 8 >= rows) return;
  }
  long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool live = row < rows;
  if (!live) row = rows - 1;                       // the partner half-wave still needs every lane for the reductions
  const long base = row * H + l32 * 8;
  bf16x8 xv[NC], rv[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) xv[i] = *(const bf16x8*)(x + base + i * 256);
  if (res) {
#pragma unroll
    for (int i = 0; i < NC; ++i) rv[i] = *(const bf16x8*)(res + base + i * 256);
  }
  float z[NC][8];
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) z[i][j] = (float)xv[i][j];
  if (thresh16) {
    const uint32_t rowkey = drop_rowkey(seed_mix(seed_ptr, salt), (uint64_t)row);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bool k[4];
        drop_keep4(rowkey, l32 * 8 + i * 256 + h * 4, thresh16, k);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[i][h * 4 + j] = k[j] ? z[i][h * 4 + j] * dscale : 0.f;
      }
  }
  float s = 0.f;
  if (res) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) z[i][j] += (float)rv[i][j];
  }
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += z[i][j];
  const float mean = half_sum(s) * (1.f / H);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = z[i][j] - mean; ss += d * d; }
  const float var = half_sum(ss) * (1.f / H);
  float rstd = rsqrtf(var + eps);
  if (!(var + eps > 0.f)) rstd = 0.f;
  if (!live) return;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = l32 * 8 + i * 256;
    const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
    const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (bf16)((z[i][j] - mean) * rstd * g0[j] + b0[j]);
      o[4 + j] = (bf16)((z[i][4 + j] - mean) * rstd * g1[j] + b1[j]);
    }
    *(bf16x8*)(y + base + i * 256) = o;
    if (zout) {
      bf16x8 zb;
#pragma unroll
      for (int j = 0; j < 8; ++j) zb[j] = (bf16)z[i][j];
      *(bf16x8*)(zout + base + i * 256) = zb;
    }
  }
  if (mean_o && l32 == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// ---------------------------------------------------------------- LN backward
// dz = rstd * (dy*g - mean_H(dy*g) - xhat * mean_H(dy*g*xhat));  dgamma += sum_rows dy*xhat; dbeta += sum_rows dy
// dx (optional) = dropout-mask(dz).  Per-column partial sums are kept in registers over the block's rows,
// combined through LDS and flushed with one atomicAdd per column per block.
// NC = 8-byte chunks per lane (3 covers H <= 768, 4 covers H <= 1024).  The raw bf16 rows of the NEXT iteration are
// requested before the current row is reduced, so the wave always has loads in flight (the kernel is pure HBM streaming:
// 2-3 rows in, 1-2 rows out per row).
#ifndef LN_BWD_FROMY_WGS
#define LN_BWD_FROMY_WGS 3      // workgroups per CU the from-output variant is compiled for (4 spills 13 registers: 52 B of scratch per lane)
#endif
template <int NC>
struct RawRow { bf16x4 v[NC]; };

template <int NC, bool EXACT>
__device__ __forceinline__ void load_raw(const bf16* __restrict__ p, int H, int lane, RawRow<NC>& r) {
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (EXACT || c < H) r.v[i] = *(const bf16x4*)(p + c);
    else r.v[i] = to_bf16x4(0.f, 0.f, 0.f, 0.f);
  }
}

// EXACT: H == 256 * NC, no column guards.  HOT: the combination every encoder layer of a training step uses (one incoming
// gradient, dropout on x, dx + dgamma + dbeta + dxsum all wanted) with the run-time switches folded away.
// FROMY (round 6): `z` holds the LayerNorm's OUTPUT y instead of its input, and the normalised value is recovered as (y - beta) / gamma
// (a channel with gamma == 0 carries no information about it: xhat = 0 there).  The forward then need not store its pre-norm sum at all --
// one 131-MB write per residual LayerNorm at the benchmark shape -- and y is alive anyway as the next GEMM's operand.  Numerically the same
// as the z form for gamma of order 1 (both read one bf16-rounded tensor; EXPERIMENTS.md 4.8).
template <int NC, bool EXACT, bool HOT, bool FROMY>
__global__ __launch_bounds__(256, FROMY ? LN_BWD_FROMY_WGS : 4) void ln_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ dy2,
                                                     const bf16* __restrict__ z, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     bf16* __restrict__ dz_o, bf16* __restrict__ dx_o,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int H,
                                                     uint32_t thresh16, float dscale, const uint64_t* seed_ptr, uint64_t salt,
                                                     int drop_on_dy, float* __restrict__ dxsum, const int* __restrict__ rows_ptr) {
  __shared__ float red[4][NC * 256];
  if (rows_ptr) { const long r_ = *rows_ptr; rows = r_ < rows ? r_ : rows; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gsum[NC][4], bsum[NC][4], xsum[NC][4];   // xsum: column sums of dx = bias gradient of the dense layer that produced x
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) gsum[i][j] = bsum[i][j] = xsum[i][j] = 0.f;
  if (HOT) { dy2 = nullptr; drop_on_dy = 0; }
  const bool has2 = !HOT && dy2 != nullptr, has_dx = HOT || dx_o != nullptr, has_xs = HOT || dxsum != nullptr, drop = HOT || thresh16 != 0;
  const uint64_t seed = drop ? seed_mix(seed_ptr, salt) : 0;
  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  RawRow<NC> ng, ng2, nz;
  float nmean = 0.f, nrstd = 0.f;
  if (row < rows) {
    load_raw<NC, EXACT>(dy + row * H, H, lane, ng);
    if (has2) load_raw<NC, EXACT>(dy2 + row * H, H, lane, ng2);
    load_raw<NC, EXACT>(z + row * H, H, lane, nz);
    nmean = FROMY ? 0.f : mean_i[row]; nrstd = rstd_i[row];
  }
  for (; row < rows; row += stride) {
    RawRow<NC> cg = ng, cg2 = ng2, cz = nz;
    const float mean = nmean, rstd = nrstd;
    const long nrow = row + stride;
    if (nrow < rows) {
      load_raw<NC, EXACT>(dy + nrow * H, H, lane, ng);
      if (has2) load_raw<NC, EXACT>(dy2 + nrow * H, H, lane, ng2);
      load_raw<NC, EXACT>(z + nrow * H, H, lane, nz);
      nmean = FROMY ? 0.f : mean_i[nrow]; nrstd = rstd_i[nrow];
    }
    float g[NC][4], xh[NC][4];
    const uint32_t rowkey = drop ? drop_rowkey(seed, (uint64_t)row) : 0u;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) g[i][j] = (float)cg.v[i][j] + (has2 ? (float)cg2.v[i][j] : 0.f);
    if (drop && drop_on_dy) {   // y_out = dropout(LN(z)) (BertEmbeddings): mask the incoming gradient
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (EXACT || c < H) {
          bool k[4];
          drop_keep4(rowkey, c, thresh16, k);
#pragma unroll
          for (int j = 0; j < 4; ++j) g[i][j] = k[j] ? g[i][j] * dscale : 0.f;
        }
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (EXACT || c < H) {
        const f32x4 gm = *(const f32x4*)(gamma + c);      // L1-resident; keeping it in registers costs occupancy
        f32x4 bt = {0.f, 0.f, 0.f, 0.f}, ig = {0.f, 0.f, 0.f, 0.f};
        if constexpr (FROMY) {
          bt = *(const f32x4*)(beta + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) ig[j] = gm[j] != 0.f ? __builtin_amdgcn_rcpf(gm[j]) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = FROMY ? ((float)cz.v[i][j] - bt[j]) * ig[j] : ((float)cz.v[i][j] - mean) * rstd;
          const float dg = g[i][j] * gm[j];
          gsum[i][j] += g[i][j] * x;
          bsum[i][j] += g[i][j];
          if constexpr (!FROMY) xh[i][j] = x;             // (FROMY recomputes it from the raw row below: 12 registers less across the reduction,
                                                          //  which is what keeps this variant inside the 128-register bound without scratch)
          g[i][j] = dg;
          s1 += dg;
          s2 += dg * x;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { if constexpr (!FROMY) xh[i][j] = 0.f; g[i][j] = 0.f; }
      }
    }
    s1 = wave_sum_dpp(s1) / H;
    s2 = wave_sum_dpp(s2) / H;
    const bool mask_dx = has_dx && drop && !drop_on_dy;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (EXACT || c < H) {
        float o[4];
        if constexpr (FROMY) {
          const f32x4 gm = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = ((float)cz.v[i][j] - bt[j]) * (gm[j] != 0.f ? __builtin_amdgcn_rcpf(gm[j]) : 0.f);
            o[j] = rstd * (g[i][j] - s1 - x * s2);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
        }
        const bf16x4 ob = to_bf16x4(o[0], o[1], o[2], o[3]);
        *(bf16x4*)(dz_o + row * H + c) = ob;
        if (has_dx) {
          bf16x4 xb = ob;
          if (mask_dx) {
            bool k[4];
            drop_keep4(rowkey, c, thresh16, k);
            xb = to_bf16x4(k[0] ? o[0] * dscale : 0.f, k[1] ? o[1] * dscale : 0.f, k[2] ? o[2] * dscale : 0.f, k[3] ? o[3] * dscale : 0.f);
          }
          *(bf16x4*)(dx_o + row * H + c) = xb;
          if (has_xs) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xsum[i][j] += (float)xb[j];
          }
        } else if (has_xs) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xsum[i][j] += (float)ob[j];
        }
      }
    }
  }
  if (!HOT && dgamma == nullptr && dxsum == nullptr) return;
  // three block reductions through one LDS buffer: dgamma, dbeta, dxsum
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    float* dst = which == 0 ? dgamma : which == 1 ? dbeta : dxsum;
    if (!HOT && dst == nullptr) continue;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        red[wave][(lane + 64 * i) * 4 + j] = which == 0 ? gsum[i][j] : which == 1 ? bsum[i][j] : xsum[i][j];
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) atomicAdd(dst + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

// ---------------------------------------------------------------- embeddings
// mode 0: z = word[id] + type[0] + pos[l]                      (text, BertEmbeddings xbert.py:209-219)
// mode 1: z = PV-token + type[0] + pos[l]   where PV-token(b,0) = cls, PV-token(b,l>0) =
//         (x[b,l-1]*w + bias) * (1-m) + masktok * m                (SPMM_models.py:82-88 then xbert.py:209-219)
// then y = dropout(LN(z)).  zout keeps z (bf16) for backward.
struct EmbedP {
  const int* ids;            // [nseq, L]  (mode 0)
  const float* word;         // [V, H]
  const float* pos;          // [P, H]
  const float* type0;        // [H]
  const float* pv_x;         // [nseq_src, L-1] (mode 1); mode 2: inputs_embeds fp32 [nseq*L, H]
  const float* pv_mask;      // [nseq_src, L-1] 1 = masked
  const float* pv_w; const float* pv_b; const float* pv_cls; const float* pv_masktok;   // [H] each
  int src_mod;               // mode 1: source row = seq % src_mod (P1 and P11 share the batch)
  const float* gamma; const float* beta;
  bf16* y; bf16* zout; float* mean_o; float* rstd_o;
  long nseq; int L, H, mode; float eps;
  int pos0;                  // position of l = 0 (decode steps embed one token at position t)
  const int* pos_ptr;        // ... or read from device memory (graph replay)
  uint32_t thresh16; float dscale; const uint64_t* seed_ptr; uint64_t salt;
};

__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(EmbedP p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.nseq * p.L) return;
  const long seq = row / p.L;
  const int l = (int)(row - seq * p.L);
  RowVec z, t, o;
  load_row_f32(p.pos + (long)(l + (p.pos_ptr ? *p.pos_ptr : p.pos0)) * p.H, p.H, lane, z);
  load_row_f32(p.type0, p.H, lane, t);
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) z.v[i][j] += t.v[i][j];
  if (p.mode == 0) {
    const int id = p.ids[row];
    load_row_f32(p.word + (long)id * p.H, p.H, lane, t);
  } else if (p.mode == 2) {
    load_row_f32(p.pv_x + row * p.H, p.H, lane, t);
  } else if (l == 0) {
    load_row_f32(p.pv_cls, p.H, lane, t);
  } else {
    const long src = (seq % p.src_mod) * (p.L - 1) + (l - 1);
    const float m = p.pv_mask[src], xv = p.pv_x[src];
    RowVec w, b;
    load_row_f32(p.pv_w, p.H, lane, w);
    load_row_f32(p.pv_b, p.H, lane, b);
    load_row_f32(p.pv_masktok, p.H, lane, t);
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) t.v[i][j] = (xv * w.v[i][j] + b.v[i][j]) * (1.f - m) + t.v[i][j] * m;
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) z.v[i][j] += t.v[i][j];
  float mean, rstd;
  ln_apply(z, p.H, lane, p.gamma, p.beta, p.eps, mean, rstd, o);
  if (p.thresh16) {
    const uint32_t rowkey = drop_rowkey(seed_mix(p.seed_ptr, p.salt), (uint64_t)row);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < p.H) {
        bool k[4];
        drop_keep4(rowkey, c, p.thresh16, k);
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[i][j] = k[j] ? o.v[i][j] * p.dscale : 0.f;
      }
    }
  }
  store_row_bf16(p.y + row * p.H, p.H, lane, o);
  if (p.zout) store_row_bf16(p.zout + row * p.H, p.H, lane, z);
  if (p.mean_o && lane == 0) { p.mean_o[row] = mean; p.rstd_o[row] = rstd; }
}

// Backward of the embedding sum, given dz [nseq*L, H] (gradient w.r.t. the pre-LayerNorm sum, bf16).
// One workgroup per position l; threads stride over H; loop over sequences.  dpos[l] has a single owner,
// type0 / cls / masktok / w / b get one atomicAdd per column per workgroup, word rows are scattered with atomics
// (PAD id 0 skipped: nn.Embedding padding_idx, xbert.py:178).
struct EmbedBwdP {
  const bf16* dz; const int* ids; const float* pv_x; const float* pv_mask; int src_mod;
  float* dword; float* dpos; float* dtype0; float* d_w; float* d_b; float* d_cls; float* d_masktok;
  long nseq; int L, H, mode;
};
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbedBwdP p) {
  const int l = blockIdx.x;
  const long s_per = (p.nseq + gridDim.y - 1) / gridDim.y;
  const long s_beg = blockIdx.y * s_per, s_end = min(p.nseq, s_beg + s_per);
  for (int c = threadIdx.x; c < p.H; c += 256) {
    float acc = 0.f, accw = 0.f, accb = 0.f, accm = 0.f;
    for (long s = s_beg; s < s_end; ++s) {
      const long row = s * p.L + l;
      const float g = (float)p.dz[row * p.H + c];
      acc += g;
      if (p.mode == 0) {
        const int id = p.ids[row];
        if (id != 0) atomicAdd(p.dword + (long)id * p.H + c, g);
      } else if (l > 0) {
        const long src = (s % p.src_mod) * (p.L - 1) + (l - 1);
        const float m = p.pv_mask[src];
        accw += g * (1.f - m) * p.pv_x[src];
        accb += g * (1.f - m);
        accm += g * m;
      }
    }
    atomicAdd(p.dpos + (long)l * p.H + c, acc);
    atomicAdd(p.dtype0 + c, acc);
    if (p.mode == 1) {
      if (l == 0) {
        atomicAdd(p.d_cls + c, acc);
      } else {
        atomicAdd(p.d_w + c, accw);
        atomicAdd(p.d_b + c, accb);
        atomicAdd(p.d_masktok + c, accm);
      }
    }
  }
}

// ---------------------------------------------------------------- transposes
// out[C, Rpad] = in[R, C]^T (bf16), zero-filled for r in [R, Rpad); optional colsum[c] += sum_r in[r][c].
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ in, long ldi, bf16* __restrict__ out,
                                                             long ldo, int R, int C, int Rpad, float* __restrict__ colsum) {
  __shared__ bf16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float cs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i, c = c0 + tx;
    const bf16 v = (r < R && c < C) ? in[(long)r * ldi + c] : (bf16)0.f;
    tile[ty * 16 + i][tx] = v;
    cs += (float)v;
  }
  if (colsum) {
    __shared__ float part[4][64];
    part[ty][tx] = cs;
    __syncthreads();
    if (ty == 0 && c0 + tx < C) atomicAdd(colsum + c0 + tx, part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
  } else {
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty * 16 + i, r = r0 + tx;
    if (c < C && r < Rpad) out[(long)c * ldo + r] = tile[tx][ty * 16 + i];
  }
}

// fp32 master weight [R, C] -> bf16 shadow [R, C] and transposed bf16 shadow [C, R]
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ in, bf16* __restrict__ out,
                                                             bf16* __restrict__ outT, int R, int C) {
  __shared__ bf16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i, c = c0 + tx;
    bf16 v = (bf16)0.f;
    if (r < R && c < C) {
      v = (bf16)in[(long)r * C + c];
      if (out) out[(long)r * C + c] = v;
    }
    tile[ty * 16 + i][tx] = v;
  }
  __syncthreads();
  if (outT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = c0 + ty * 16 + i, r = r0 + tx;
      if (c < C && r < R) outT[(long)c * R + r] = tile[tx][ty * 16 + i];
    }
  }
}

// All transposed bf16 dgrad shadows in ONE launch: descriptor table (device) of {fp32 src, bf16 dstT, R, C, first tile};
// workgroup b finds its matrix by binary search over the tile prefix.
struct CtDesc { const float* src; bf16* dstT; int R, C; int tile0, ntc; };
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const CtDesc* __restrict__ descs, int ndesc) {
  __shared__ bf16 tile[64][66];
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CtDesc d = descs[lo];
  const int tt = blockIdx.x - d.tile0;
  const int r0 = (tt / d.ntc) * 64, c0 = (tt % d.ntc) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty * 16 + i, c = c0 + tx;
    tile[ty * 16 + i][tx] = (r < d.R && c < d.C) ? (bf16)d.src[(long)r * d.C + c] : (bf16)0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty * 16 + i, r = r0 + tx;
    if (c < d.C && r < d.R) d.dstT[(long)c * d.R + r] = tile[tx][ty * 16 + i];
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(in + i * 4);
    *(bf16x4*)(out + i * 4) = to_bf16x4(v[0], v[1], v[2], v[3]);
  }
}
__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ in, float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const bf16x4 v = *(const bf16x4*)(in + i * 4);
    f32x4 o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *(f32x4*)(out + i * 4) = o;
  }
}

// dst_f32[row_idx[r] or r] (+)= src_bf16[r]   (row length H, 4-element vectors; rows with a negative index are skipped)
__global__ void acc_rows_kernel(float* __restrict__ dst, long ldd, const bf16* __restrict__ src, long lds,
                                const long* __restrict__ idx, long rows, int H, int atomic) {
  const int h4 = H / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * h4; i += (long)gridDim.x * blockDim.x) {
    const long r = i / h4;
    const int c = (int)(i - r * h4) * 4;
    const long dr = idx ? idx[r] : r;
    if (dr < 0) continue;                      // a source row nobody owns (rows past a text negative's length)
    const bf16x4 v = *(const bf16x4*)(src + r * lds + c);
    float* d = dst + dr * ldd + c;
    if (atomic) {
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(d + j, (float)v[j]);
    } else {
      f32x4 o = *(f32x4*)d;
      o[0] += (float)v[0]; o[1] += (float)v[1]; o[2] += (float)v[2]; o[3] += (float)v[3];
      *(f32x4*)d = o;
    }
  }
}
// dst_bf16[r] = src_bf16[idx[r]]  (row length H elements, 8-element vectors)
__global__ void gather_rows_kernel(bf16* __restrict__ dst, const bf16* __restrict__ src, const long* __restrict__ idx,
                                   long rows, int H) {
  const int h8 = H / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * h8; i += (long)gridDim.x * blockDim.x) {
    const long r = i / h8;
    const int c = (int)(i - r * h8) * 8;
    *(bf16x8*)(dst + r * H + c) = *(const bf16x8*)(src + idx[r] * H + c);
  }
}

int grid_for(long work, int block) {
  long g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}


}  // namespace

extern "C" int spmm_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* zout,
                           float* mean, float* rstd, long rows, int H, float eps, float dropout_p,
                           const uint64_t* seed_ptr, uint64_t salt, const int* rows_dev, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && H > 0 && H % 4 == 0 && H <= 1024, "spmm_ln_fwd: rows=%ld H=%d (need H%%4==0, H<=1024)", rows, H);
  SPMM_CHECK_SHAPE(dropout_p == 0.f || seed_ptr, "spmm_ln_fwd: dropout needs a device seed");
  const uint32_t th = (uint32_t)(dropout_p * 65536.f + 0.5f);
  const float ds = 1.f / (1.f - dropout_p);
#define LN_FWD16(NC)                                                                                                          \
  hipLaunchKernelGGL(ln_fwd16_kernel<NC>, dim3((rows + 7) / 8), dim3(256), 0, stream, (const bf16*)x, (const bf16*)res, gamma, \
                     beta, (bf16*)y, (bf16*)zout, mean, rstd, rows, eps, th, ds, seed_ptr, salt, rows_dev)
  if (H == 768) LN_FWD16(3);
  else if (H == 1024) LN_FWD16(4);
  else if (H == 512) LN_FWD16(2);
  else if (H == 256) LN_FWD16(1);
  else
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (const bf16*)res, gamma, beta,
                       (bf16*)y, (bf16*)zout, mean, rstd, rows, H, eps, th, ds, seed_ptr, salt, rows_dev);
#undef LN_FWD16
  SPMM_LAUNCH_CHECK("spmm_ln_fwd");
  return SPMM_OK;
}

extern "C" int spmm_ln_fwd_r32(const void* x, const float* res32, const float* gamma, const float* beta, void* y, float* y32, void* zout,
                               float* mean, float* rstd, long rows, int H, float eps, float dropout_p, const uint64_t* seed_ptr,
                               uint64_t salt, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && H > 0 && H % 4 == 0 && H <= 1024, "spmm_ln_fwd_r32: rows=%ld H=%d (need H%%4==0, H<=1024)", rows, H);
  SPMM_CHECK_SHAPE(dropout_p == 0.f || seed_ptr, "spmm_ln_fwd_r32: dropout needs a device seed");
  SPMM_CHECK_SHAPE((mean == nullptr) == (rstd == nullptr), "spmm_ln_fwd_r32: mean and rstd come together");
  hipLaunchKernelGGL(ln_fwd_r32_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)x, res32, gamma, beta, (bf16*)y, y32,
                     (bf16*)zout, mean, rstd, rows, H, eps, (uint32_t)(dropout_p * 65536.f + 0.5f), 1.f / (1.f - dropout_p), seed_ptr, salt);
  SPMM_LAUNCH_CHECK("spmm_ln_fwd_r32");
  return SPMM_OK;
}

extern "C" int spmm_ln_bwd(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd,
                           const float* gamma, void* dz, void* dx, float* dgamma, float* dbeta, long rows, int H,
                           float dropout_p, const uint64_t* seed_ptr, uint64_t salt, int drop_on_dy, float* dxsum, const int* rows_dev,
                           const float* beta_from_y, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && H > 0 && H % 4 == 0 && H <= 1024, "spmm_ln_bwd: rows=%ld H=%d", rows, H);
  SPMM_CHECK_SHAPE(beta_from_y != nullptr || mean != nullptr, "spmm_ln_bwd: the row means are required unless `z` holds the LayerNorm output (beta_from_y)");
  SPMM_CHECK_SHAPE(beta_from_y == nullptr || !drop_on_dy, "spmm_ln_bwd: the output of a LayerNorm followed by dropout does not determine the normalised values");
  SPMM_CHECK_SHAPE(dropout_p == 0.f || seed_ptr, "spmm_ln_bwd: dropout needs a device seed");
  long g = (rows + 3) / 4;
  // Grid: every workgroup ends with 3 H same-address atomics (d gamma, d beta, bias-gradient column sums), so fewer, longer workgroups win until the
  // rows per wave get too few to keep loads in flight: rows / 64 workgroups, between 256 and 512 (round 6, tools/bench_ln.py: 85 k rows 110 -> 100 us,
  // 28.7 k 44 -> 33-35, 13.8 k 38 -> 21.5 against the fixed 1024 of rounds 2-5).  SPMM_LN_BWD_GRID overrides (tools only).
  static const long genv = getenv("SPMM_LN_BWD_GRID") ? atol(getenv("SPMM_LN_BWD_GRID")) : 0;
  long gmax = genv > 0 ? genv : rows / 64;
  if (genv <= 0) gmax = gmax < 256 ? 256 : (gmax > 512 ? 512 : gmax);
  if (g > gmax) g = gmax;
  const uint32_t th = (uint32_t)(dropout_p * 65536.f + 0.5f);
  const float ds = 1.f / (1.f - dropout_p);
  const bool hot = !dy2 && dx && dgamma && dbeta && dxsum && th && !drop_on_dy;
#define LN_BWD_LAUNCH(NC, EX, HOT, FY)                                                                                                \
  hipLaunchKernelGGL((ln_bwd_kernel<NC, EX, HOT, FY>), dim3(g), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)dy2, (const bf16*)z, \
                     mean, rstd, gamma, beta_from_y, (bf16*)dz, (bf16*)dx, dgamma, dbeta, rows, H, th, ds, seed_ptr, salt, drop_on_dy, dxsum, rows_dev)
  if (beta_from_y != nullptr) {
    if (H == 768 && hot) LN_BWD_LAUNCH(3, true, true, true);
    else if (H <= 768) LN_BWD_LAUNCH(3, false, false, true);
    else LN_BWD_LAUNCH(4, false, false, true);
  } else {
    if (H == 768 && hot) LN_BWD_LAUNCH(3, true, true, false);
    else if (H <= 768) LN_BWD_LAUNCH(3, false, false, false);
    else LN_BWD_LAUNCH(4, false, false, false);
  }
#undef LN_BWD_LAUNCH
  SPMM_LAUNCH_CHECK("spmm_ln_bwd");
  return SPMM_OK;
}

extern "C" int spmm_embed_ln_fwd(int mode, const int* ids, const float* word, const float* pos, const float* type0,
                                 const float* pv_x, const float* pv_mask, const float* pv_w, const float* pv_b,
                                 const float* pv_cls, const float* pv_masktok, int src_mod, const float* gamma,
                                 const float* beta, void* y, void* zout, float* mean, float* rstd, long nseq, int L, int H,
                                 float eps, float dropout_p, const uint64_t* seed_ptr, uint64_t salt, hipStream_t stream) {
  SPMM_CHECK_SHAPE(nseq > 0 && L > 0 && H % 4 == 0 && H <= 1024, "spmm_embed_ln_fwd: nseq=%ld L=%d H=%d", nseq, L, H);
  SPMM_CHECK_SHAPE(mode == 0 ? (ids && word) : mode == 2 ? (pv_x != nullptr) : (pv_x && pv_mask && pv_w && pv_b && pv_cls && pv_masktok && src_mod > 0),
                   "spmm_embed_ln_fwd: missing inputs for mode %d", mode);
  SPMM_CHECK_SHAPE(dropout_p == 0.f || seed_ptr, "spmm_embed_ln_fwd: dropout needs a device seed");
  EmbedP p = {};
  p.ids = ids; p.word = word; p.pos = pos; p.type0 = type0; p.pv_x = pv_x; p.pv_mask = pv_mask; p.pv_w = pv_w; p.pv_b = pv_b;
  p.pv_cls = pv_cls; p.pv_masktok = pv_masktok; p.src_mod = src_mod; p.gamma = gamma; p.beta = beta; p.y = (bf16*)y;
  p.zout = (bf16*)zout; p.mean_o = mean; p.rstd_o = rstd; p.nseq = nseq; p.L = L; p.H = H; p.mode = mode; p.eps = eps;
  p.thresh16 = (uint32_t)(dropout_p * 65536.f + 0.5f); p.dscale = 1.f / (1.f - dropout_p); p.seed_ptr = seed_ptr; p.salt = salt;
  hipLaunchKernelGGL(embed_ln_fwd_kernel, dim3((nseq * L + 3) / 4), dim3(256), 0, stream, p);
  SPMM_LAUNCH_CHECK("spmm_embed_ln_fwd");
  return SPMM_OK;
}

extern "C" int spmm_embed_step_ln_fwd(const int* ids, int pos_index, const int* pos_ptr, const float* word, const float* pos, const float* type0,
                                      const float* gamma, const float* beta, void* y, long rows, int H, float eps,
                                      hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && pos_index >= 0 && H % 4 == 0 && H <= 1024, "spmm_embed_step_ln_fwd: rows=%ld pos=%d H=%d", rows, pos_index, H);
  SPMM_CHECK_SHAPE(ids && word && pos && type0 && gamma && beta && y, "spmm_embed_step_ln_fwd: null argument");
  EmbedP p = {};
  p.ids = ids; p.word = word; p.pos = pos; p.type0 = type0; p.gamma = gamma; p.beta = beta; p.y = (bf16*)y;
  p.nseq = rows; p.L = 1; p.H = H; p.mode = 0; p.eps = eps; p.pos0 = pos_index; p.pos_ptr = pos_ptr; p.dscale = 1.f;
  hipLaunchKernelGGL(embed_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, p);
  SPMM_LAUNCH_CHECK("spmm_embed_step_ln_fwd");
  return SPMM_OK;
}

extern "C" int spmm_embed_bwd(int mode, const void* dz, const int* ids, const float* pv_x, const float* pv_mask, int src_mod,
                              float* dword, float* dpos, float* dtype0, float* d_w, float* d_b, float* d_cls,
                              float* d_masktok, long nseq, int L, int H, hipStream_t stream) {
  SPMM_CHECK_SHAPE(nseq > 0 && L > 0 && H > 0, "spmm_embed_bwd: nseq=%ld L=%d H=%d", nseq, L, H);
  EmbedBwdP p = {(const bf16*)dz, ids, pv_x, pv_mask, src_mod, dword, dpos, dtype0, d_w, d_b, d_cls, d_masktok, nseq, L, H, mode};
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(L, nseq >= 32 ? 16 : 1), dim3(256), 0, stream, p);
  SPMM_LAUNCH_CHECK("spmm_embed_bwd");
  return SPMM_OK;
}

namespace {
__global__ __launch_bounds__(256) void segment_sum_kernel(const bf16* __restrict__ src, const int* __restrict__ start,
                                                          const int* __restrict__ list, bf16* __restrict__ out, long W) {
  const int u = blockIdx.y;
  const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= W) return;
  const int k0 = start[u], k1 = start[u + 1];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; ++k) {
    const bf16x8 v = *(const bf16x8*)(src + (long)list[k] * W + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
  *(bf16x8*)(out + (long)u * W + c) = o;
}
}  // namespace

extern "C" int spmm_segment_sum_bf16(const void* src, const int* start, const int* list, void* out, int U, long W, hipStream_t stream) {
  SPMM_CHECK_SHAPE(U > 0 && U <= 65535 && W > 0 && W % 8 == 0, "spmm_segment_sum_bf16: U=%d W=%ld (W %% 8 == 0)", U, W);
  hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((W / 8 + 255) / 256), U), dim3(256), 0, stream, (const bf16*)src, start, list,
                     (bf16*)out, W);
  SPMM_LAUNCH_CHECK("spmm_segment_sum_bf16");
  return SPMM_OK;
}

extern "C" int spmm_transpose_bf16(const void* in, long ldi, void* out, long ldo, int R, int C, int Rpad, float* colsum,
                                   hipStream_t stream) {
  SPMM_CHECK_SHAPE(R > 0 && C > 0 && Rpad >= R && ldo >= Rpad, "spmm_transpose_bf16: R=%d C=%d Rpad=%d ldo=%ld", R, C, Rpad, ldo);
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (Rpad + 63) / 64), dim3(256), 0, stream, (const bf16*)in, ldi,
                     (bf16*)out, ldo, R, C, Rpad, colsum);
  SPMM_LAUNCH_CHECK("spmm_transpose_bf16");
  return SPMM_OK;
}

extern "C" int spmm_cast_transpose(const float* in, void* out, void* outT, int R, int C, hipStream_t stream) {
  SPMM_CHECK_SHAPE(R > 0 && C > 0, "spmm_cast_transpose: R=%d C=%d", R, C);
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, in, (bf16*)out, (bf16*)outT, R, C);
  SPMM_LAUNCH_CHECK("spmm_cast_transpose");
  return SPMM_OK;
}

extern "C" long spmm_cast_transpose_desc_bytes(void) { return sizeof(CtDesc); }
extern "C" int spmm_cast_transpose_multi(const void* descs_dev, int ndesc, int total_tiles, hipStream_t stream) {
  SPMM_CHECK_SHAPE(ndesc > 0 && total_tiles > 0, "spmm_cast_transpose_multi: ndesc=%d tiles=%d", ndesc, total_tiles);
  hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3(total_tiles), dim3(256), 0, stream, (const CtDesc*)descs_dev, ndesc);
  SPMM_LAUNCH_CHECK("spmm_cast_transpose_multi");
  return SPMM_OK;
}

extern "C" int spmm_cast_f32_bf16(const float* in, void* out, long n, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && n % 4 == 0, "spmm_cast_f32_bf16: n=%ld must be a positive multiple of 4", n);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, stream, in, (bf16*)out, n / 4);
  SPMM_LAUNCH_CHECK("spmm_cast_f32_bf16");
  return SPMM_OK;
}
extern "C" int spmm_cast_bf16_f32(const void* in, float* out, long n, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && n % 4 == 0, "spmm_cast_bf16_f32: n=%ld must be a positive multiple of 4", n);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, stream, (const bf16*)in, out, n / 4);
  SPMM_LAUNCH_CHECK("spmm_cast_bf16_f32");
  return SPMM_OK;
}

extern "C" int spmm_acc_rows(float* dst, long ldd, const void* src, long lds, const long* idx, long rows, int H, int atomic,
                             hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && H > 0 && H % 4 == 0, "spmm_acc_rows: rows=%ld H=%d", rows, H);
  hipLaunchKernelGGL(acc_rows_kernel, dim3(grid_for(rows * (H / 4), 256)), dim3(256), 0, stream, dst, ldd, (const bf16*)src, lds,
                     idx, rows, H, atomic);
  SPMM_LAUNCH_CHECK("spmm_acc_rows");
  return SPMM_OK;
}
extern "C" int spmm_gather_rows(void* dst, const void* src, const long* idx, long rows, int H, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && H > 0 && H % 8 == 0 && idx, "spmm_gather_rows: rows=%ld H=%d", rows, H);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * (H / 8), 256)), dim3(256), 0, stream, (bf16*)dst, (const bf16*)src,
                     idx, rows, H);
  SPMM_LAUNCH_CHECK("spmm_gather_rows");
  return SPMM_OK;
}
