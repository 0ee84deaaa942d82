// Attention core for gfx950: softmax(Q K^T / sqrt(64) + mask) -> dropout -> . V, forward and backward.
//
// Replaces BertSelfAttention.forward xbert.py:305-354 (scores :305, /sqrt(d) :323, additive mask :327,
// softmax :335, dropout :344, context :350, head merge :352-354) for both the self- and the cross-attention
// instantiation (:285-290), with the masks of get_extended_attention_mask :889-948 (0 / -10000, causal AND
// padding when is_decoder) and invert_attention_mask (0 / finfo.min, call site :1038-1043) folded in from the
// raw [nseq, Lkv] 0/1 mask -- no [B,1,L,L] tensor is ever materialised.
//
// One workgroup per (sequence, head); head_dim is 64; Lq, Lkv <= 128, so the whole K/V panel of a head sits
// in LDS and the score tile lives in MFMA accumulators: single pass, no online-softmax rescaling.
// Scores are computed TRANSPOSED (S^T = K Q^T via v_mfma_f32_32x32x16_bf16 with K rows as the A operand), so
// a lane owns one query row (lane&31) and its kv entries sit in its accumulator registers: the row max / sum
// need one cross-lane exchange (lane ^ 32) only.  The probabilities feed the second MFMA (O^T = V^T P^T)
// straight from registers as the B operand: MFMA sums over its k-slots in any order, so the k-slot <-> kv
// assignment is chosen to be exactly the accumulator layout (kv = 4g+{0..3, 8..11} per 16-wide step) and the
// V^T fragments are read from LDS in that same order -- no cross-lane shuffle of P at all.
// Backward recomputes P from the saved log-sum-exp and runs two phases: (A) a wave owns 32 query rows ->
// dQ; (B) a wave owns 32 kv rows -> dK, dV (scores recomputed un-transposed so that the reduction index q is
// the register index).  Dropout masks are regenerated from (seed, element index).
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

constexpr int HD = 64;            // head dim
constexpr int ROWB = 128;         // bytes per row of a row-major [L][64] bf16 LDS tile
constexpr int TILE = 128 * ROWB;  // 16 KiB

struct AttnP {
  const bf16* Q; long ldq;
  const bf16* K; long ldk;
  const bf16* V; long ldv;
  const int* kmask;               // [nseq, Lkv] 1 = attend, or null (all ones)
  const int* kv_seq;              // [nseq] key/value sequence read by query sequence s (null: s itself); dK/dV stay per query sequence
  // Packed (variable-length) layouts: sequence s owns q_len[s] <= Lq query rows starting at row q_row0[s]; key/value
  // source u owns kv_len[u] <= Lkv rows starting at kv_row0[u].  Null = dense layout (row = seq*L + pos).  LSE, kmask and
  // the dropout counter keep the dense [.., Lq, Lkv] indexing, so packed and padded runs draw the same masks.
  const int* q_row0; const int* q_len; const int* kv_row0; const int* kv_len;
  // Sequences longer than 128 are processed as <= 128-long chunks of queries and keys by several launches (ops.py): q_off /
  // kv_off are the chunk's positions inside the sequence (causal mask); d_mode 1 = only write D[q] = sum_kv P dP of this key
  // chunk to Dbuf, 2 = take D[q] (summed over all key chunks by the caller) from Dbuf instead of computing it.
  int q_off, kv_off, d_mode; float* Dbuf;
  bf16* O; long ldo;              // fwd output (unused by backward)
  float* LSE;                     // [nseq, nH, Lq]
  const bf16* dO; long lddo;
  bf16* dQ; long lddq;
  bf16* dK; long lddk;
  bf16* dV; long lddv;
  int nseq, nH, Lq, Lkv;
  int causal_from;                // sequences >= causal_from get the causal mask (self-attention only)
  float mask_neg;                 // -10000 (self) or -FLT_MAX (cross)
  uint32_t drop_thresh16;         // 0 = no dropout
  float drop_scale;               // 1/(1-p)
  const uint64_t* seed_ptr;       // device seed (graph-replay safe), may be null when no dropout
  uint64_t seed_salt;             // per-call-site salt
};

// Row-major [128][64] bf16 tile, 16-B chunk index XOR-swizzled with rotr3((row>>1)&7):
//  * ds_read_b128 of 16 rows (distinct mod 16) at one k-slot -> 16 distinct 16-B slots of the 256-B bank window;
//  * ds_read_b64_tr_b16 of 4 rows x 64 B per 32-lane half -> rows r,r+1 sit in different window halves and rows r,r+2 in
//    different 64-B spans (bit 2 of the chunk index flips with bit 1 of the row): conflict-free as well.
__device__ __forceinline__ int frot(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int swz(int row, int c16) { return row * ROWB + ((c16 ^ frot(row)) << 4); }

// reference mask arithmetic: (1 - causal*mask) * -10000 (self) / (1 - mask) * finfo.min (cross)
__device__ __forceinline__ float score_bias(int mask_kv, bool causal, int q, int kv, float mask_neg) {
  const bool ok = mask_kv && (!causal || kv <= q);
  return ok ? 0.f : mask_neg;
}

// LDS-DMA staging of the [L][64] head slice: global_load_lds_dwordx4 writes lane-linear, so the swizzle is applied on
// the SOURCE chunk.  Rows >= L re-read row L-1 (finite data; such rows are masked / never stored).
__device__ __forceinline__ void stage_head(const bf16* __restrict__ src, long ld, int L, char* tile, int tid, int nthreads,
                                           int rows = 128) {
  // only the 32-row tiles the MFMAs will touch are staged (`rows` = tiles * 32): a 54-token sequence moves 64 rows, not 128
  for (int id = tid; id < rows * 8; id += nthreads) {
    const int row = id >> 3, pc = id & 7;
    const int lc = pc ^ frot(row);
    const int grow = row < L ? row : L - 1;
    const bf16* g = src + (long)grow * ld + lc * 8;
    const int wave_base = __builtin_amdgcn_readfirstlane((id & ~63) * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(tile + wave_base), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 ld_rm(const char* tile, int row, int slot) { return *(const bf16x8*)(tile + swz(row, slot)); }

__device__ __forceinline__ bf16x8 join8(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// A-operand fragments X^T[d][rows] for both 32-wide d blocks, straight from the row-major tile with the LDS transpose read:
// lane l (d = dt*32 + (l&31), g = l>>5) receives X[rb + 4g + {0..3, 8..11}][d] -- the k-slot order the probabilities use.
__device__ __forceinline__ void ld_tr2(const char* tile, int rb, int lane, bf16x8 (&out)[2]) {
  const int i16 = lane & 15, j = lane >> 4;
  const int row0 = rb + 4 * (j >> 1) + (i16 >> 2);
  const int col = (j & 1) * 16 + (i16 & 3) * 4;
  const unsigned a00 = (unsigned)(size_t)(tile + swz(row0, col >> 3) + (col & 7) * 2);
  const unsigned a01 = (unsigned)(size_t)(tile + swz(row0 + 8, col >> 3) + (col & 7) * 2);
  const unsigned a10 = (unsigned)(size_t)(tile + swz(row0, (col + 32) >> 3) + (col & 7) * 2);
  const unsigned a11 = (unsigned)(size_t)(tile + swz(row0 + 8, (col + 32) >> 3) + (col & 7) * 2);
  bf16x4 r0, r1, r2, r3;
  asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a00), "v"(a01), "v"(a10), "v"(a11) : "memory");
  out[0] = join8(r0, r1);
  out[1] = join8(r2, r3);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int hf) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16)v[hf * 8 + e];
  return r;
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// ------------------------------------------------------------------------------------------ forward
constexpr int FWD_LDS = 2 * TILE + 128 * 4;

template <int NT>   // NT = ceil(Lkv / 32)
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + TILE;
  float* mb = (float*)(smem + 2 * TILE);   // mask value per kv (1/0), -1 = padding
  const int h = blockIdx.x, seq = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5, nthreads = blockDim.x;
  const long kvs = p.kv_seq ? (long)p.kv_seq[seq] : (long)seq;
  const int Lq = p.q_len ? p.q_len[seq] : p.Lq, Lkv = p.kv_len ? p.kv_len[kvs] : p.Lkv;
  const long qrow = p.q_row0 ? (long)p.q_row0[seq] : (long)seq * p.Lq;
  const long kvrow = p.kv_row0 ? (long)p.kv_row0[kvs] : kvs * p.Lkv;
  const bf16* Qg = p.Q + qrow * p.ldq + h * HD;
  const bf16* Kg = p.K + kvrow * p.ldk + h * HD;
  const bf16* Vg = p.V + kvrow * p.ldv + h * HD;
  stage_head(Kg, p.ldk, Lkv, Ks, tid, nthreads, NT * 32);
  stage_head(Vg, p.ldv, Lkv, Vs, tid, nthreads, NT * 32);
  for (int j = tid; j < 128; j += nthreads)
    mb[j] = j < Lkv ? (p.kmask ? (float)p.kmask[(long)seq * p.Lkv + j] : 1.f) : -1.f;
  // Q fragments straight from HBM (B operand: row = lane&31, 8 consecutive d at (kk*2+g)*8)
  const int q = wave * 32 + (lane & 31);
  const int qc = q < Lq ? q : Lq - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(Qg + (long)qc * p.ldq + (kk * 2 + g) * 8);
  __syncthreads();

  const bool causal = seq >= p.causal_from;
  f32x16 st[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    st[t] = zero16();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) st[t] = MFMA32(ld_rm(Ks, t * 32 + (lane & 31), kk * 2 + g), qf[kk], st[t]);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kv = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
      const float mv = mb[kv];
      const float s = mv < 0.f ? -INFINITY : st[t][r] * 0.125f + score_bias(mv > 0.5f, causal, q + p.q_off, kv + p.kv_off, p.mask_neg);
      st[t][r] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(st[t][r] - mx);
      st[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  if (p.LSE && q < Lq && g == 0) p.LSE[((long)seq * p.nH + h) * p.Lq + q] = mx + __logf(sum);
  if (p.drop_thresh16) {
    const uint32_t rowkey = drop_rowkey(seed_mix(p.seed_ptr, p.seed_salt), ((uint64_t)seq * p.nH + h) * p.Lq + q);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {                 // registers 4*gq .. 4*gq+3 hold 4 consecutive keys
        const uint32_t pr = (uint32_t)(t * 32 + 8 * gq + 4 * g) >> 1;
        const uint32_t r0 = drop_pair(rowkey, pr), r1 = drop_pair(rowkey, pr + 1);
        st[t][gq * 4 + 0] = (r0 & 0xffffu) >= p.drop_thresh16 ? st[t][gq * 4 + 0] * p.drop_scale : 0.f;
        st[t][gq * 4 + 1] = (r0 >> 16) >= p.drop_thresh16 ? st[t][gq * 4 + 1] * p.drop_scale : 0.f;
        st[t][gq * 4 + 2] = (r1 & 0xffffu) >= p.drop_thresh16 ? st[t][gq * 4 + 2] * p.drop_scale : 0.f;
        st[t][gq * 4 + 3] = (r1 >> 16) >= p.drop_thresh16 ? st[t][gq * 4 + 3] * p.drop_scale : 0.f;
      }
  }
  // O^T[d][q] = sum_kv V^T[d][kv] P^T[kv][q]
  f32x16 ot[2] = {zero16(), zero16()};
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const bf16x8 pf = pack8(st[t], hf);
      bf16x8 vf[2];
      ld_tr2(Vs, t * 32 + hf * 16, lane, vf);
      ot[0] = MFMA32(vf[0], pf, ot[0]);
      ot[1] = MFMA32(vf[1], pf, ot[1]);
    }
  if (q < Lq) {
    bf16* Og = p.O + (qrow + q) * p.ldo + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d = dt * 32 + 8 * gq + 4 * g;
        *(bf16x4*)(Og + d) = to_bf16x4(ot[dt][gq * 4] * inv, ot[dt][gq * 4 + 1] * inv, ot[dt][gq * 4 + 2] * inv,
                                       ot[dt][gq * 4 + 3] * inv);
      }
  }
}

// ------------------------------------------------------------------------------------------ backward
constexpr int BWD_LDS = 4 * TILE + 4 * 128 * 4;

template <int NT>   // NT = ceil(Lkv / 32)
__global__ __launch_bounds__(256, NT <= 3 ? 3 : 2) void attn_bwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS holds only the 32-row tiles in use (K, V: NT tiles; Q, dO: ceil(max Lq / 32) tiles): a 54 x 54 head takes 34 KiB
  // instead of 66, and with the 160-VGPR budget of the NT <= 2 instances three workgroups share a CU instead of two.
  const int kvb = NT * 32 * 128, qb = ((p.Lq + 31) >> 5) * 32 * 128;
  char* Ks = smem;
  char* Vs = smem + kvb;
  char* Qs = smem + 2 * kvb;
  char* dOs = Qs + qb;
  float* mb = (float*)(dOs + qb);
  float* lse = mb + 128;
  float* Dq = lse + 128;
  uint32_t* rk = (uint32_t*)(Dq + 128);      // dropout row keys of the 128 query rows
  const int h = blockIdx.x, seq = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 5;
  const long kvs = p.kv_seq ? (long)p.kv_seq[seq] : (long)seq;
  const int Lq = p.q_len ? p.q_len[seq] : p.Lq, Lkv = p.kv_len ? p.kv_len[kvs] : p.Lkv;
  const long qrow = p.q_row0 ? (long)p.q_row0[seq] : (long)seq * p.Lq;
  const long kvrow = p.kv_row0 ? (long)p.kv_row0[kvs] : kvs * p.Lkv;
  const long dkvrow = p.kv_seq ? (long)seq * p.Lkv : kvrow;     // shared sources: dK/dV per query sequence, dense
  const bf16* Qg = p.Q + qrow * p.ldq + h * HD;
  const bf16* Kg = p.K + kvrow * p.ldk + h * HD;
  const bf16* Vg = p.V + kvrow * p.ldv + h * HD;
  const bf16* dOg = p.dO + qrow * p.lddo + h * HD;
  const int qrows = ((Lq + 31) >> 5) * 32;
  stage_head(Kg, p.ldk, Lkv, Ks, tid, 256, NT * 32);
  stage_head(Vg, p.ldv, Lkv, Vs, tid, 256, NT * 32);
  stage_head(Qg, p.ldq, Lq, Qs, tid, 256, qrows);
  stage_head(dOg, p.lddo, Lq, dOs, tid, 256, qrows);
  if (tid < 128) {
    const int j = tid;
    mb[j] = j < Lkv ? (p.kmask ? (float)p.kmask[(long)seq * p.Lkv + j] : 1.f) : -1.f;
    lse[j] = j < Lq ? p.LSE[((long)seq * p.nH + h) * p.Lq + j] : 0.f;
    Dq[j] = 0.f;
  }
  __syncthreads();

  const bool causal = seq >= p.causal_from;
  const bool drop = p.drop_thresh16 != 0;
  const uint64_t seed = drop ? seed_mix(p.seed_ptr, p.seed_salt) : 0;
  const uint64_t headbase = ((uint64_t)seq * p.nH + h) * (uint64_t)p.Lq;
  if (drop && tid < 128) rk[tid] = drop_rowkey(seed, headbase + tid);
  const int NTq = (Lq + 31) >> 5;

  // ---- phase A: wave owns query tile `wave` -> D[q] = sum_kv P dP (fp32, exactly consistent with ds) and dQ.
  // D is NOT taken from rowsum(dO * O): O is bf16-rounded, and when dP is nearly constant over kv (real models)
  // ds = P (dP - D) is a small difference of large numbers that such a D would swamp.
  if (wave < NTq) {
    const int q = wave * 32 + (lane & 31);
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[kk] = ld_rm(Qs, q, kk * 2 + g);
      dof[kk] = ld_rm(dOs, q, kk * 2 + g);
    }
    const float lq = lse[q];
    const uint32_t rowkey = drop ? drop_rowkey(seed, headbase + q) : 0u;
    f32x16 st[NT], dp[NT];
    float dloc = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      st[t] = zero16();
      dp[t] = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        st[t] = MFMA32(ld_rm(Ks, t * 32 + (lane & 31), kk * 2 + g), qf[kk], st[t]);
        dp[t] = MFMA32(ld_rm(Vs, t * 32 + (lane & 31), kk * 2 + g), dof[kk], dp[t]);
      }
      uint32_t keepbits = 0xffffu;                     // bit r: probability (q, kv(r)) survived dropout in forward
      if (drop) {
        keepbits = 0;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const uint32_t pr2 = (uint32_t)(t * 32 + 8 * gq + 4 * g) >> 1;
          const uint32_t r0 = drop_pair(rowkey, pr2), r1 = drop_pair(rowkey, pr2 + 1);
          keepbits |= ((r0 & 0xffffu) >= p.drop_thresh16 ? 1u : 0u) << (gq * 4);
          keepbits |= ((r0 >> 16) >= p.drop_thresh16 ? 2u : 0u) << (gq * 4);
          keepbits |= ((r1 & 0xffffu) >= p.drop_thresh16 ? 4u : 0u) << (gq * 4);
          keepbits |= ((r1 >> 16) >= p.drop_thresh16 ? 8u : 0u) << (gq * 4);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kv = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        const float mv = mb[kv];
        float pr = 0.f, dpr = 0.f;
        if (mv >= 0.f && q < Lq) {
          const float s = st[t][r] * 0.125f + score_bias(mv > 0.5f, causal, q + p.q_off, kv + p.kv_off, p.mask_neg);
          pr = __expf(s - lq);
          dpr = dp[t][r];
          if (drop) dpr = ((keepbits >> r) & 1u) ? dpr * p.drop_scale : 0.f;
        }
        st[t][r] = pr;
        dp[t][r] = dpr;
        dloc += pr * dpr;
      }
    }
    dloc += __shfl_xor(dloc, 32, 64);
    if (p.d_mode == 1) {
      if (g == 0 && q < Lq) p.Dbuf[headbase + q] = dloc;
    } else if (p.d_mode == 2) {
      dloc = q < Lq ? p.Dbuf[headbase + q] : 0.f;
    }
    if (g == 0) Dq[q] = dloc;
    f32x16 dq[2] = {zero16(), zero16()};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = st[t][r] * (dp[t][r] - dloc);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 dsf = pack8(st[t], hf);
        bf16x8 kf[2];
        ld_tr2(Ks, t * 32 + hf * 16, lane, kf);
        dq[0] = MFMA32(kf[0], dsf, dq[0]);
        dq[1] = MFMA32(kf[1], dsf, dq[1]);
      }
    }
    if (q < Lq && p.d_mode != 1) {
      bf16* dQg = p.dQ + (qrow + q) * p.lddq + h * HD;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *(bf16x4*)(dQg + dt * 32 + 8 * gq + 4 * g) =
              to_bf16x4(dq[dt][gq * 4] * 0.125f, dq[dt][gq * 4 + 1] * 0.125f, dq[dt][gq * 4 + 2] * 0.125f,
                        dq[dt][gq * 4 + 3] * 0.125f);
    }
  }
  if (p.d_mode == 1) return;   // partial-D pass: nothing else is written
  __syncthreads();   // D[q] of every query tile is in LDS

  // ---- phase B: wave owns kv tile `wave` -> dK, dV   (S[q][kv]: lane = kv column, registers = q rows)
  if (wave < NT) {
    const int kv = wave * 32 + (lane & 31);
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kf[kk] = ld_rm(Ks, kv, kk * 2 + g);
      vf[kk] = ld_rm(Vs, kv, kk * 2 + g);
    }
    const float mv = mb[kv];
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    for (int qt = 0; qt < NTq; ++qt) {
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        s = MFMA32(ld_rm(Qs, qt * 32 + (lane & 31), kk * 2 + g), kf[kk], s);
        dp = MFMA32(ld_rm(dOs, qt * 32 + (lane & 31), kk * 2 + g), vf[kk], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        float pd = 0.f, ds = 0.f;
        if (mv >= 0.f && q < Lq) {
          const float sc = s[r] * 0.125f + score_bias(mv > 0.5f, causal, q + p.q_off, kv + p.kv_off, p.mask_neg);
          const float pr = __expf(sc - lse[q]);
          float dpr = dp[r];
          pd = pr;
          if (drop) {
            const uint32_t rr = drop_pair(rk[q], (uint32_t)kv >> 1);
            const bool keep = ((kv & 1) ? (rr >> 16) : (rr & 0xffffu)) >= p.drop_thresh16;
            pd = keep ? pr * p.drop_scale : 0.f;
            dpr = keep ? dpr * p.drop_scale : 0.f;
          }
          ds = pr * (dpr - Dq[q]);
        }
        s[r] = pd;
        dp[r] = ds;
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const bf16x8 pf = pack8(s, hf), dsf = pack8(dp, hf);
        bf16x8 dof[2], qf[2];
        ld_tr2(dOs, qt * 32 + hf * 16, lane, dof);
        ld_tr2(Qs, qt * 32 + hf * 16, lane, qf);
        dv[0] = MFMA32(dof[0], pf, dv[0]);
        dv[1] = MFMA32(dof[1], pf, dv[1]);
        dk[0] = MFMA32(qf[0], dsf, dk[0]);
        dk[1] = MFMA32(qf[1], dsf, dk[1]);
      }
    }
    if (kv < Lkv) {
      bf16* dKg = p.dK + (dkvrow + kv) * p.lddk + h * HD;
      bf16* dVg = p.dV + (dkvrow + kv) * p.lddv + h * HD;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int d = dt * 32 + 8 * gq + 4 * g;
          *(bf16x4*)(dKg + d) = to_bf16x4(dk[dt][gq * 4] * 0.125f, dk[dt][gq * 4 + 1] * 0.125f,
                                          dk[dt][gq * 4 + 2] * 0.125f, dk[dt][gq * 4 + 3] * 0.125f);
          *(bf16x4*)(dVg + d) = to_bf16x4(dv[dt][gq * 4], dv[dt][gq * 4 + 1], dv[dt][gq * 4 + 2], dv[dt][gq * 4 + 3]);
        }
    }
  }
}

int check_common(const char* name, int nseq, int nH, int Lq, int Lkv, long ldq, long ldk, long ldv) {
  SPMM_CHECK_SHAPE(nseq > 0 && nH > 0, "%s: empty problem", name);
  SPMM_CHECK_SHAPE(Lq >= 1 && Lq <= 128 && Lkv >= 1 && Lkv <= 128, "%s: Lq=%d Lkv=%d must be in [1,128]", name, Lq, Lkv);
  SPMM_CHECK_SHAPE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0, "%s: row strides must be multiples of 8", name);
  return SPMM_OK;
}

}  // namespace

// head_dim is fixed at 64 (config_bert.json: 768 / 12).  Tensors are token-major: row = seq*L + pos, head h at
// columns [h*64, h*64+64) of the given base pointer.
extern "C" int spmm_attn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                             const int* kmask, const int* kv_seq, const int* q_row0, const int* q_len, const int* kv_row0,
                             const int* kv_len, void* O, long ldo, float* LSE, int nseq, int nH, int Lq,
                             int Lkv, int causal_from, int is_cross, float dropout_p, const uint64_t* seed_ptr,
                             uint64_t seed_salt, int q_off, int kv_off, hipStream_t stream) {
  int rc = check_common("spmm_attn_fwd", nseq, nH, Lq, Lkv, ldq, ldk, ldv);
  if (rc) return rc;
  SPMM_CHECK_SHAPE(dropout_p == 0.f || seed_ptr != nullptr, "spmm_attn_fwd: dropout needs a device seed");
  SPMM_CHECK_SHAPE((q_row0 == nullptr) == (q_len == nullptr) && (kv_row0 == nullptr) == (kv_len == nullptr),
                   "spmm_attn_fwd: row0 and len arrays come in pairs");
  AttnP p = {};
  p.Q = (const bf16*)Q; p.ldq = ldq; p.K = (const bf16*)K; p.ldk = ldk; p.V = (const bf16*)V; p.ldv = ldv;
  p.kmask = kmask; p.kv_seq = kv_seq; p.q_row0 = q_row0; p.q_len = q_len; p.kv_row0 = kv_row0; p.kv_len = kv_len;
  p.q_off = q_off; p.kv_off = kv_off;
  p.O = (bf16*)O; p.ldo = ldo; p.LSE = LSE; p.nseq = nseq; p.nH = nH; p.Lq = Lq; p.Lkv = Lkv;
  p.causal_from = is_cross ? nseq : causal_from;
  p.mask_neg = is_cross ? -3.4028234663852886e38f : -10000.f;
  p.drop_thresh16 = (uint32_t)(dropout_p * 65536.f + 0.5f);
  p.drop_scale = 1.f / (1.f - dropout_p);
  p.seed_ptr = seed_ptr; p.seed_salt = seed_salt;
  const int nt = (Lkv + 31) / 32, nw = (Lq + 31) / 32;
  const size_t lds = FWD_LDS;
  dim3 grid(nH, nseq), block(64 * nw);
  switch (nt) {
    case 1: hipLaunchKernelGGL(attn_fwd_kernel<1>, grid, block, lds, stream, p); break;
    case 2: hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, block, lds, stream, p); break;
    case 3: hipLaunchKernelGGL(attn_fwd_kernel<3>, grid, block, lds, stream, p); break;
    default: hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, block, lds, stream, p); break;
  }
  SPMM_LAUNCH_CHECK("spmm_attn_fwd");
  return SPMM_OK;
}

extern "C" int spmm_attn_bwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                             const int* kmask, const int* kv_seq, const int* q_row0, const int* q_len, const int* kv_row0,
                             const int* kv_len, const void* O, long ldo, const float* LSE, const void* dO,
                             long lddo, void* dQ, long lddq, void* dK, long lddk, void* dV, long lddv, int nseq, int nH, int Lq,
                             int Lkv, int causal_from, int is_cross, float dropout_p, const uint64_t* seed_ptr,
                             uint64_t seed_salt, int q_off, int kv_off, int d_mode, float* Dbuf, hipStream_t stream) {
  int rc = check_common("spmm_attn_bwd", nseq, nH, Lq, Lkv, ldq, ldk, ldv);
  if (rc) return rc;
  SPMM_CHECK_SHAPE(dropout_p == 0.f || seed_ptr != nullptr, "spmm_attn_bwd: dropout needs a device seed");
  SPMM_CHECK_SHAPE((q_row0 == nullptr) == (q_len == nullptr) && (kv_row0 == nullptr) == (kv_len == nullptr),
                   "spmm_attn_bwd: row0 and len arrays come in pairs");
  SPMM_CHECK_SHAPE(d_mode == 0 || ((d_mode == 1 || d_mode == 2) && Dbuf != nullptr), "spmm_attn_bwd: d_mode=%d needs Dbuf", d_mode);
  // > 64 KiB dynamic LDS needs the opt-in: once per process (function-local static: initialised exactly once, thread-safe --
  // backward entries are called from autograd worker threads)
  static const hipError_t attr_rc = [] {
    const void* fns[4] = {(const void*)attn_bwd_kernel<1>, (const void*)attn_bwd_kernel<2>, (const void*)attn_bwd_kernel<3>,
                          (const void*)attn_bwd_kernel<4>};
    for (int i = 0; i < 4; ++i) {
      hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }();
  if (attr_rc != hipSuccess) {
    spmm_set_error("spmm_attn_bwd: cannot raise dynamic LDS to %d: %s", BWD_LDS, hipGetErrorString(attr_rc));
    return SPMM_ERR_LAUNCH;
  }
  AttnP p = {};
  p.Q = (const bf16*)Q; p.ldq = ldq; p.K = (const bf16*)K; p.ldk = ldk; p.V = (const bf16*)V; p.ldv = ldv;
  p.kmask = kmask; p.kv_seq = kv_seq; p.q_row0 = q_row0; p.q_len = q_len; p.kv_row0 = kv_row0; p.kv_len = kv_len;
  p.q_off = q_off; p.kv_off = kv_off; p.d_mode = d_mode; p.Dbuf = Dbuf;
  p.O = (bf16*)O; p.ldo = ldo; p.LSE = (float*)LSE; p.dO = (const bf16*)dO; p.lddo = lddo;
  p.dQ = (bf16*)dQ; p.lddq = lddq; p.dK = (bf16*)dK; p.lddk = lddk; p.dV = (bf16*)dV; p.lddv = lddv;
  p.nseq = nseq; p.nH = nH; p.Lq = Lq; p.Lkv = Lkv;
  p.causal_from = is_cross ? nseq : causal_from;
  p.mask_neg = is_cross ? -3.4028234663852886e38f : -10000.f;
  p.drop_thresh16 = (uint32_t)(dropout_p * 65536.f + 0.5f);
  p.drop_scale = 1.f / (1.f - dropout_p);
  p.seed_ptr = seed_ptr; p.seed_salt = seed_salt;
  const int nt_b = (Lkv + 31) / 32 > 4 ? 4 : (Lkv + 31) / 32;
  const size_t lds_b = (size_t)2 * nt_b * 32 * 128 + (size_t)2 * ((Lq + 31) / 32) * 32 * 128 + 4 * 128 * 4;
  switch ((Lkv + 31) / 32) {
    case 1: hipLaunchKernelGGL(attn_bwd_kernel<1>, dim3(nH, nseq), dim3(256), lds_b, stream, p); break;
    case 2: hipLaunchKernelGGL(attn_bwd_kernel<2>, dim3(nH, nseq), dim3(256), lds_b, stream, p); break;
    case 3: hipLaunchKernelGGL(attn_bwd_kernel<3>, dim3(nH, nseq), dim3(256), lds_b, stream, p); break;
    default: hipLaunchKernelGGL(attn_bwd_kernel<4>, dim3(nH, nseq), dim3(256), lds_b, stream, p); break;
  }
  SPMM_LAUNCH_CHECK("spmm_attn_bwd");
  return SPMM_OK;
}
