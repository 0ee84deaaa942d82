// Fused cross-attention block, forward, for gfx950 -- the unit BASELINE.json's metric names.
//
// Replaces, in ONE launch, what xbert.py does in BertSelfAttention.forward :305-354 (scores, /sqrt(64), additive encoder mask
// :327 from invert_attention_mask :1038-1043, softmax :335, dropout :344, context :350, head merge :352-354) followed by
// BertSelfOutput.forward :369-373 (dense, dropout, + residual, LayerNorm) for the cross-attention instantiation (:285-290):
//
//     Y = LayerNorm(dropout(softmax(Q K^T / 8 + mask) V . Wo^T + bo) + X)
//
// (Q = X Wq^T + bq and the shared K / V projections stay GEMM launches: K/V are projected once per unique source sequence,
// spmm_amd/engine.py KVSource, which a per-panel fused projection would undo.)
//
// Row-panel design.  A workgroup (4 waves, one per SIMD, 512 registers each) owns 64 query rows of ONE sequence and the FULL
// hidden width H = 128 * NCT, so the LayerNorm row is whole inside the workgroup: a 54-token property sequence is one panel, a
// 128-token SMILES sequence two.  The output projection's K dimension is walked head PAIR by head pair (128 wide, NS = H/128
// steps).  Per step:
//   (a) the LDS-DMA of the NEXT pair's K/V head tiles is issued (double-buffered; in flight under (b));
//   (b) the four waves each run the attention core of one (32-row tile, head): S^T = K Q^T in MFMA accumulators, softmax with
//       one lane per query row, P feeds O^T = V^T P^T from registers (the scheme of attention.hip), the 32x64 context tile
//       goes to LDS as bf16;
//   (c) every wave multiplies the 64x128 context panel with ITS quarter of the output columns:
//       acc[64 rows][WN = H/4 cols] += ctx[64][128] . Wo[cols][pair]^T, 192 fp32 accumulators per lane at H = 768.  Wo is read
//       from a FRAGMENT-ORDERED bf16 shadow (spmm_xattn_pack_wo): every wave-level load is 1 KiB contiguous, straight into the
//       MFMA A operand -- no LDS for the weights, which no two waves share.
// Epilogue: bias, hidden dropout (same counter hash as spmm_ln_fwd, so spmm_ln_bwd regenerates the mask), + residual (panel
// staged in LDS by DMA), row statistics across the four waves through LDS, LayerNorm; z (pre-norm, for the backward) and y leave
// through an LDS image as full 16-B-per-lane row-major stores.  The context panel is also written out (coalesced, from its LDS
// image) when the backward needs it (weight gradient of Wo, attention backward).
//
// All LDS accesses made while a DMA is in flight are inline asm: a compiler-visible LDS access (or __syncthreads) would make the
// compiler's waitcnt pass drain the DMA first.
#include "common.h"
#include "attn_tiles.h"
#include "../../include/spmm_hip.h"

namespace {

struct XattnP {
  const bf16* Q; long ldq;
  const bf16* K; long ldk;
  const bf16* V; long ldv;
  const int* kmask;                 // [nseq, Lkv] 1 = attend, or null
  const int* kv_seq;                // [nseq] key/value source of query sequence s (null: s)
  const int* q_row0; const int* q_len; const int* kv_row0; const int* kv_len;     // packed layouts (attention.hip)
  const bf16* WoF;                  // fragment-ordered output-projection weight (spmm_xattn_pack_wo)
  const float* bo;                  // [H]
  const bf16* R; long ldr;          // residual rows, indexed like Q
  const float* gamma; const float* beta; float eps;
  bf16* Y; long ldy;
  bf16* Z; long ldz;                // pre-LayerNorm sum (null: not kept)
  float* mean; float* rstd;         // [rows] (null: not kept)
  bf16* CTX; long ldc;              // attention context (null: not kept)
  float* LSE;                       // [nseq, nH, Lq] (null: not kept)
  int nseq, nH, Lq, Lkv;
  float mask_neg;
  uint32_t drop_a16; float scale_a; uint64_t salt_a;     // attention-probability dropout
  uint32_t drop_h16; float scale_h; uint64_t salt_h;     // hidden dropout on the dense output
  const uint64_t* seed_ptr;
  long row_base;                    // row of the group's first row in the whole token batch (hidden-dropout counter)
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define XA_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define XA_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)p; }

// four / one 16-byte LDS reads, no wait (the caller waits with XA_LGKM0 and fences with XA_SB before using the registers)
__device__ __forceinline__ void lds_rd4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, bf16x8 (&d)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
               : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
template <typename T>
__device__ __forceinline__ void lds_rd1(unsigned a, T& d) {
  asm volatile("ds_read_b128 %0, %1" : "=&v"(d) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_wr8(unsigned a, bf16x4 v) { asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory"); }

// the transpose reads of ld_tr2x2 (attn_tiles.h) without the wait: A operands V^T for two 16-row k-blocks
__device__ __forceinline__ void tr_issue(const char* tile, int rb, int lane, bf16x4 (&r)[8]) {
  const int i16 = lane & 15, j = lane >> 4;
  const int row0 = rb + 4 * (j >> 1) + (i16 >> 2);
  const int col = (j & 1) * 16 + (i16 & 3) * 4;
  unsigned a[8];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    a[b * 4 + 0] = lds_addr(tile + swz(row0 + 16 * b, col >> 3) + (col & 7) * 2);
    a[b * 4 + 1] = lds_addr(tile + swz(row0 + 16 * b + 8, col >> 3) + (col & 7) * 2);
    a[b * 4 + 2] = lds_addr(tile + swz(row0 + 16 * b, (col + 32) >> 3) + (col & 7) * 2);
    a[b * 4 + 3] = lds_addr(tile + swz(row0 + 16 * b + 8, (col + 32) >> 3) + (col & 7) * 2);
  }
  asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
               "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15"
               : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
}

// LDS image of a [64][H] bf16 panel (residual in, z / y out): 16-byte chunk c16 of row r sits at slot r * C16 + (c16 ^ (r & 15)).
template <int C16>
__device__ __forceinline__ int pslot(int row, int c16) { return row * C16 + (c16 ^ (row & 15)); }

template <int NT, int NCT>   // NT = ceil(Lkv / 32) key tiles; H = 128 * NCT
__global__ __launch_bounds__(256, 1) void xattn_fwd_kernel(XattnP p) {
  constexpr int H = 128 * NCT, NS = NCT, WN = 32 * NCT, C16 = H / 8;
  constexpr int KVT = NT * 32 * ROWB;            // one [NT*32][64] head tile
  constexpr int KVBUF = 4 * KVT;                 // K(h0) V(h0) K(h1) V(h1)
  constexpr int PANEL = 64 * H * 2;
  constexpr int R0 = 2 * KVBUF > PANEL ? 2 * KVBUF : PANEL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kvb = smem;                              // two K/V buffers; later the residual / z / y panel image
  char* ctxs = smem + R0;                        // [64][128] bf16 context panel of the current head pair
  float* mb = (float*)(ctxs + 64 * 256);         // additive score bias per key
  float* red = mb + 128;                         // [2][4][64] row partial sums

  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = wave & 1, hsel = wave >> 1;
  const int seq = blockIdx.y, r0p = blockIdx.x * 64;
  const long kvs = p.kv_seq ? (long)p.kv_seq[seq] : (long)seq;
  const int Lq = p.q_len ? p.q_len[seq] : p.Lq, Lkv = p.kv_len ? p.kv_len[kvs] : p.Lkv;
  if (r0p >= Lq) return;                         // workgroup-uniform
  const int nvalid = Lq - r0p < 64 ? Lq - r0p : 64;
  const long qrow = (p.q_row0 ? (long)p.q_row0[seq] : (long)seq * p.Lq) + r0p;      // the panel's first row
  const long kvrow = p.kv_row0 ? (long)p.kv_row0[kvs] : kvs * p.Lkv;
  const int myq = rt * 32 + l31;                 // panel row this lane owns in the attention core
  const int myqc = myq < nvalid ? myq : nvalid - 1;

  const float neg2 = fmaxf(p.mask_neg * LOG2E, -3.4028234e38f);
  for (int j = tid; j < 128; j += 256)
    mb[j] = j < Lkv ? ((p.kmask == nullptr || p.kmask[(long)seq * p.Lkv + j]) ? 0.f : neg2) : -INFINITY;

  bf16x8 qn[4];                                  // Q fragments of the NEXT step (B operand: row = lane&31, d = (kk*2+g)*8 ..)
  {
    const bf16* Qg = p.Q + (qrow + myqc) * p.ldq + hsel * HD;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qn[kk] = *(const bf16x8*)(Qg + (kk * 2 + g) * 8);
  }
  asm volatile("" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]));      // the loads are waited for HERE, before any DMA is in flight
  __syncthreads();                               // mb visible (no DMA pending yet: a plain barrier)

  const bf16* Kg = p.K + kvrow * p.ldk;
  const bf16* Vg = p.V + kvrow * p.ldv;
  auto stage_pair = [&](int s, int b) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      stage_head(Kg + (2 * s + hh) * HD, p.ldk, Lkv, kvb + b * KVBUF + (2 * hh) * KVT, tid, 256, NT * 32);
      stage_head(Vg + (2 * s + hh) * HD, p.ldv, Lkv, kvb + b * KVBUF + (2 * hh + 1) * KVT, tid, 256, NT * 32);
    }
  };
  stage_pair(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  XA_SB();

  f32x16 acc[2][NCT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[a][ct] = zero16();

  const uint64_t seed_a = p.drop_a16 ? seed_mix(p.seed_ptr, p.salt_a) : 0;

#pragma unroll 1
  for (int s = 0; s < NS; ++s) {                                     // (a real loop: unrolled, its six bodies spill)
    const int cur = s & 1;
    const int h = 2 * s + hsel;
    if (s + 1 < NS) stage_pair(s + 1, cur ^ 1);                      // (a) next pair's K/V: in flight under the attention core
    XA_SB();
    // ------------------------------------------------------------------ (b) attention core of (row tile rt, head h)
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = qn[kk];
    const char* Ks = kvb + cur * KVBUF + (2 * hsel) * KVT;
    const char* Vs = Ks + KVT;
    f32x16 st[NT];
    {
      bf16x8 kf[2][4];
      lds_rd4(lds_addr(Ks + swz(l31, g)), lds_addr(Ks + swz(l31, 2 + g)), lds_addr(Ks + swz(l31, 4 + g)), lds_addr(Ks + swz(l31, 6 + g)), kf[0]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        XA_LGKM0();
        XA_SB();
        if (t + 1 < NT) {
          const int r = (t + 1) * 32 + l31;
          lds_rd4(lds_addr(Ks + swz(r, g)), lds_addr(Ks + swz(r, 2 + g)), lds_addr(Ks + swz(r, 4 + g)), lds_addr(Ks + swz(r, 6 + g)), kf[(t + 1) & 1]);
        }
        st[t] = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) st[t] = MFMA32(kf[t & 1][kk], qf[kk], st[t]);
        XA_SB();
      }
    }
    // V^T fragments of key tile 0 start now and land under the softmax arithmetic
    bf16x4 vr[2][8];
    tr_issue(Vs, 0, lane, vr[0]);
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4 bias[4];
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) lds_rd1(lds_addr(mb + t * 32 + 8 * gq + 4 * g), bias[gq]);
      XA_LGKM0();
      XA_SB();
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sc = __builtin_fmaf(st[t][gq * 4 + j], 0.125f * LOG2E, bias[gq][j]);
          st[t][gq * 4 + j] = sc;
          mx = fmaxf(mx, sc);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[t][r] - mx);
        st[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    const int qpos = r0p + myq;                                      // position inside the sequence
    if (p.LSE && myq < nvalid && g == 0) p.LSE[((long)seq * p.nH + h) * p.Lq + qpos] = mx * LN2 + __logf(sum);
    if (p.drop_a16) {
      const uint32_t rowkey = drop_rowkey(seed_a, ((uint64_t)seq * p.nH + h) * p.Lq + qpos);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const uint32_t pr = (uint32_t)(t * 32 + 8 * gq + 4 * g) >> 1;
          const uint32_t r0 = drop_pair(rowkey, pr), r1 = drop_pair(rowkey, pr + 1);
          st[t][gq * 4 + 0] = (r0 & 0xffffu) >= p.drop_a16 ? st[t][gq * 4 + 0] : 0.f;
          st[t][gq * 4 + 1] = (r0 >> 16) >= p.drop_a16 ? st[t][gq * 4 + 1] : 0.f;
          st[t][gq * 4 + 2] = (r1 & 0xffffu) >= p.drop_a16 ? st[t][gq * 4 + 2] : 0.f;
          st[t][gq * 4 + 3] = (r1 >> 16) >= p.drop_a16 ? st[t][gq * 4 + 3] : 0.f;
        }
    }
    f32x16 ot[2] = {zero16(), zero16()};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bf16x8 pf0 = pack8(st[t], 0), pf1 = pack8(st[t], 1);
      XA_LGKM0();
      XA_SB();
      if (t + 1 < NT) tr_issue(Vs, (t + 1) * 32, lane, vr[(t + 1) & 1]);
      const bf16x4(&v)[8] = vr[t & 1];
      ot[0] = MFMA32(join8(v[0], v[1]), pf0, ot[0]);
      ot[1] = MFMA32(join8(v[2], v[3]), pf0, ot[1]);
      ot[0] = MFMA32(join8(v[4], v[5]), pf1, ot[0]);
      ot[1] = MFMA32(join8(v[6], v[7]), pf1, ot[1]);
      XA_SB();
    }
    {
      const float osc = p.drop_a16 ? inv * p.scale_a : inv;
      const unsigned rowb = lds_addr(ctxs + myq * 256) + g * 8;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ch = hsel * 8 + dt * 4 + gq;                      // 16-B chunk of the [64][128] panel: d = dt*32 + 8*gq + 4*g ..
          lds_wr8(rowb + ((ch ^ (myq & 15)) << 4), to_bf16x4(ot[dt][gq * 4] * osc, ot[dt][gq * 4 + 1] * osc, ot[dt][gq * 4 + 2] * osc,
                                                             ot[dt][gq * 4 + 3] * osc));
        }
    }
    XA_LGKM0();
    __builtin_amdgcn_s_barrier();                                    // barrier A: the context panel of this pair is complete
    XA_SB();
    // ------------------------------------------------------------------ (c) output projection, K slice [128 s, 128 s + 128)
    if (p.CTX) {                                                     // the panel leaves row-major, 16 B per lane, 256 B per row
      u32x4 cv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i, row = id >> 4, ch = id & 15;
        lds_rd1(lds_addr(ctxs + row * 256) + ((ch ^ (row & 15)) << 4), cv[i]);
      }
      XA_LGKM0();
      XA_SB();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i, row = id >> 4, ch = id & 15;
        if (row < nvalid) *(u32x4*)(p.CTX + (qrow + row) * p.ldc + s * 128 + ch * 8) = cv[i];
      }
    } else {
      XA_LGKM0();
      XA_SB();
    }
    if (s + 1 < NS) {                                                // Q fragments of the next head (ordinary loads: used next step)
      const bf16* Qg = p.Q + (qrow + myqc) * p.ldq + (h + 2) * HD;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qn[kk] = *(const bf16x8*)(Qg + (kk * 2 + g) * 8);
    }
    if (s == NS - 1) {
      // K/V buffers are free from here on (barrier A): the residual panel streams into them under the last projection step.
      // Rows past the sequence re-read its last valid row (finite, never stored).
      for (int id = tid; id < 64 * C16; id += 256) {
        const int row = id / C16, c16 = (id % C16) ^ (row & 15);
        const int rr = row < nvalid ? row : nvalid - 1;
        const bf16* gsrc = p.R + (qrow + rr) * p.ldr + c16 * 8;
        const int wave_base = __builtin_amdgcn_readfirstlane((id & ~63) * 16);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)(kvb + wave_base), 16, 0, 0);
      }
    }
    {
      // B operands: row = a*32 + l31, k = g*64 + kk*8 .. (head g of the pair), four k-steps at a time (32 registers instead of 64)
      const bf16x8* wb = (const bf16x8*)p.WoF + ((long)(s * 4 + wave) * NCT * 8) * 64 + lane;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        bf16x8 cf[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int row = a * 32 + l31;
          const unsigned rb2 = lds_addr(ctxs + row * 256);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) lds_rd1(rb2 + (((g * 8 + kh * 4 + kk) ^ (row & 15)) << 4), cf[a][kk]);
        }
        XA_LGKM0();
        XA_SB();
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          bf16x8 wf[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) wf[kk] = wb[(ct * 8 + kh * 4 + kk) * 64];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            acc[0][ct] = MFMA32(wf[kk], cf[0][kk], acc[0][ct]);
            acc[1][ct] = MFMA32(wf[kk], cf[1][kk], acc[1][ct]);
          }
        }
      }
    }
    if (s + 1 < NS) asm volatile("" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]));      // waited for before the next DMA is issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's share of the next pair's K/V (or the residual) has landed
    XA_LGKM0();
    __builtin_amdgcn_s_barrier();                                    // barrier B
    XA_SB();
  }

  // ---------------------------------------------------------------------------------------------------- epilogue
  // accumulator layout: lane (row = a*32 + l31, g) holds columns wave*WN + ct*32 + 8*gq + 4*g + {0..3} in acc[a][ct][gq*4 ..]
  const char* rs = kvb;                                              // residual image; z and then y overwrite it in place
  const uint64_t seed_h = p.drop_h16 ? seed_mix(p.seed_ptr, p.salt_h) : 0;
  float rsum[2] = {0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int row = a * 32 + l31;
    const uint32_t rowkey = p.drop_h16 ? drop_rowkey(seed_h, (uint64_t)(p.row_base + qrow + row)) : 0u;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c = wave * WN + ct * 32 + 8 * gq + 4 * g;
        const f32x4 b4 = *(const f32x4*)(p.bo + c);
        const bf16x4 r4 = *(const bf16x4*)(rs + pslot<C16>(row, c >> 3) * 16 + g * 8);
        bool keep[4] = {true, true, true, true};
        if (p.drop_h16) drop_keep4(rowkey, c, p.drop_h16, keep);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[a][ct][gq * 4 + j] + b4[j];
          if (p.drop_h16) v = keep[j] ? v * p.scale_h : 0.f;
          v += (float)r4[j];
          acc[a][ct][gq * 4 + j] = v;
          rsum[a] += v;
        }
      }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    rsum[a] += __shfl_xor(rsum[a], 32, 64);
    if (g == 0) red[wave * 64 + a * 32 + l31] = rsum[a];
  }
  __syncthreads();
  float mean[2], rstd[2], rss[2] = {0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int row = a * 32 + l31;
    mean[a] = (red[row] + red[64 + row] + red[128 + row] + red[192 + row]) * (1.f / H);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[a][ct][r] - mean[a]; rss[a] += d * d; }
    rss[a] += __shfl_xor(rss[a], 32, 64);
    if (g == 0) red[256 + wave * 64 + row] = rss[a];
  }
  if (p.Z) {                                                         // z image over the residual image (each lane rewrites what it read)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = a * 32 + l31;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int c = wave * WN + ct * 32 + 8 * gq + 4 * g;
          *(bf16x4*)(kvb + pslot<C16>(row, c >> 3) * 16 + g * 8) =
              to_bf16x4(acc[a][ct][gq * 4], acc[a][ct][gq * 4 + 1], acc[a][ct][gq * 4 + 2], acc[a][ct][gq * 4 + 3]);
        }
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int row = a * 32 + l31;
    const float var = (red[256 + row] + red[320 + row] + red[384 + row] + red[448 + row]) * (1.f / H);
    rstd[a] = rsqrtf(var + p.eps);
    if (!(var + p.eps > 0.f)) rstd[a] = 0.f;
    if (p.mean && wave == 0 && g == 0 && row < nvalid) { p.mean[qrow + row] = mean[a]; p.rstd[qrow + row] = rstd[a]; }
  }
  if (p.Z) {
    for (int id = tid; id < 64 * C16; id += 256) {
      const int row = id / C16, c16 = id % C16;
      if (row < nvalid) *(u32x4*)(p.Z + (qrow + row) * p.ldz + c16 * 8) = *(const u32x4*)(kvb + pslot<C16>(row, c16) * 16);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int row = a * 32 + l31;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c = wave * WN + ct * 32 + 8 * gq + 4 * g;
        const f32x4 gm = *(const f32x4*)(p.gamma + c), bt = *(const f32x4*)(p.beta + c);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (acc[a][ct][gq * 4 + j] - mean[a]) * rstd[a] * gm[j] + bt[j];
        *(bf16x4*)(kvb + pslot<C16>(row, c >> 3) * 16 + g * 8) = to_bf16x4(o[0], o[1], o[2], o[3]);
      }
  }
  __syncthreads();
  for (int id = tid; id < 64 * C16; id += 256) {
    const int row = id / C16, c16 = id % C16;
    if (row < nvalid) *(u32x4*)(p.Y + (qrow + row) * p.ldy + c16 * 8) = *(const u32x4*)(kvb + pslot<C16>(row, c16) * 16);
  }
}

// Wo [H, H] bf16 row-major ([out, in]) -> fragment order: element e of the 16-byte piece ((s*4 + w)*NCT + ct)*8 + kk of lane l is
// Wo[w*WN + ct*32 + (l & 31)][s*128 + (l >> 5)*64 + kk*8 + e]: the A operand of MFMA 32x32x16 for k-slots (head l>>5 of pair s,
// d = kk*8 ..), so that a wave-level fragment load is 1 KiB contiguous.
__global__ void xattn_pack_wo_kernel(const bf16* __restrict__ W, long ldw, bf16x8* __restrict__ out, int NCT) {
  const int H = 128 * NCT, WN = 32 * NCT;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte piece per thread
  const long total = (long)H * H / 8;
  if (idx >= total) return;
  const int l = idx & 63;
  long f = idx >> 6;
  const int kk = f & 7; f >>= 3;
  const int ct = f % NCT; f /= NCT;
  const int w = f & 3;
  const int s = (int)(f >> 2);
  const int col = w * WN + ct * 32 + (l & 31);
  const int k = s * 128 + (l >> 5) * 64 + kk * 8;
  out[idx] = *(const bf16x8*)(W + (long)col * ldw + k);
}

template <int NT, int NCT>
constexpr int xattn_lds() {
  constexpr int H = 128 * NCT, KVBUF = 4 * NT * 32 * ROWB, PANEL = 64 * H * 2;
  return (2 * KVBUF > PANEL ? 2 * KVBUF : PANEL) + 64 * 256 + 128 * 4 + 512 * 4;
}

template <int NT, int NCT>
int launch_xattn(const XattnP& p, dim3 grid, hipStream_t stream) {
  static const hipError_t attr_rc =
      hipFuncSetAttribute((const void*)xattn_fwd_kernel<NT, NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, xattn_lds<NT, NCT>());
  if (attr_rc != hipSuccess) {
    spmm_set_error("spmm_xattn_fwd: cannot raise dynamic LDS to %d: %s", xattn_lds<NT, NCT>(), hipGetErrorString(attr_rc));
    return SPMM_ERR_LAUNCH;
  }
  constexpr int lds = xattn_lds<NT, NCT>();
  xattn_fwd_kernel<NT, NCT><<<grid, dim3(256), lds, stream>>>(p);
  return SPMM_OK;
}

}  // namespace

extern "C" int spmm_xattn_pack_wo(const void* W, long ldw, void* out, int H, hipStream_t stream) {
  SPMM_CHECK_SHAPE(H == 128 || H == 256 || H == 768, "spmm_xattn_pack_wo: H=%d (supported: 128, 256, 768)", H);
  SPMM_CHECK_SHAPE(ldw >= H && ldw % 8 == 0, "spmm_xattn_pack_wo: ldw=%ld", ldw);
  const long total = (long)H * H / 8;
  hipLaunchKernelGGL(xattn_pack_wo_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)W, ldw, (bf16x8*)out, H / 128);
  SPMM_LAUNCH_CHECK("spmm_xattn_pack_wo");
  return SPMM_OK;
}

extern "C" int spmm_xattn_supported(int H, int nH, int Lq, int Lkv) {
  return (H == 128 || H == 256 || H == 768) && nH * 64 == H && Lq >= 1 && Lkv >= 1 && Lkv <= 128;
}

extern "C" int spmm_xattn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const int* kmask,
                              const int* kv_seq, const int* q_row0, const int* q_len, const int* kv_row0, const int* kv_len,
                              const void* WoF, const float* bo, const void* R, long ldr, const float* gamma, const float* beta, float eps,
                              void* Y, long ldy, void* Z, long ldz, float* mean, float* rstd, void* CTX, long ldc, float* LSE,
                              int nseq, int nH, int Lq, int Lkv, float attn_dropout_p, uint64_t salt_a, float hidden_dropout_p,
                              uint64_t salt_h, const uint64_t* seed_ptr, long row_base, hipStream_t stream) {
  const int H = nH * 64;
  SPMM_CHECK_SHAPE(spmm_xattn_supported(H, nH, Lq, Lkv), "spmm_xattn_fwd: H=%d nH=%d Lq=%d Lkv=%d unsupported (H in {128,256,768}, Lkv <= 128)", H,
                   nH, Lq, Lkv);
  SPMM_CHECK_SHAPE(nseq > 0 && Q && K && V && WoF && bo && R && gamma && beta && Y, "spmm_xattn_fwd: null argument");
  SPMM_CHECK_SHAPE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldr % 8 == 0 && ldy % 8 == 0 && (!Z || ldz % 8 == 0) && (!CTX || ldc % 8 == 0),
                   "spmm_xattn_fwd: row strides must be multiples of 8 elements");
  SPMM_CHECK_SHAPE((attn_dropout_p == 0.f && hidden_dropout_p == 0.f) || seed_ptr != nullptr, "spmm_xattn_fwd: dropout needs a device seed");
  SPMM_CHECK_SHAPE((q_row0 == nullptr) == (q_len == nullptr) && (kv_row0 == nullptr) == (kv_len == nullptr),
                   "spmm_xattn_fwd: row0 and len arrays come in pairs");
  SPMM_CHECK_SHAPE((mean == nullptr) == (rstd == nullptr), "spmm_xattn_fwd: mean and rstd come together");
  XattnP p = {};
  p.Q = (const bf16*)Q; p.ldq = ldq; p.K = (const bf16*)K; p.ldk = ldk; p.V = (const bf16*)V; p.ldv = ldv;
  p.kmask = kmask; p.kv_seq = kv_seq; p.q_row0 = q_row0; p.q_len = q_len; p.kv_row0 = kv_row0; p.kv_len = kv_len;
  p.WoF = (const bf16*)WoF; p.bo = bo; p.R = (const bf16*)R; p.ldr = ldr; p.gamma = gamma; p.beta = beta; p.eps = eps;
  p.Y = (bf16*)Y; p.ldy = ldy; p.Z = (bf16*)Z; p.ldz = ldz; p.mean = mean; p.rstd = rstd; p.CTX = (bf16*)CTX; p.ldc = ldc; p.LSE = LSE;
  p.nseq = nseq; p.nH = nH; p.Lq = Lq; p.Lkv = Lkv;
  p.mask_neg = -3.4028234663852886e38f;
  p.drop_a16 = (uint32_t)(attn_dropout_p * 65536.f + 0.5f); p.scale_a = 1.f / (1.f - attn_dropout_p); p.salt_a = salt_a;
  p.drop_h16 = (uint32_t)(hidden_dropout_p * 65536.f + 0.5f); p.scale_h = 1.f / (1.f - hidden_dropout_p); p.salt_h = salt_h;
  p.seed_ptr = seed_ptr; p.row_base = row_base;
  const dim3 grid((Lq + 63) / 64, nseq);
  const int nt = (Lkv + 31) / 32;
  int rc = SPMM_OK;
#define XA_CASE(NT_, NCT_) rc = launch_xattn<NT_, NCT_>(p, grid, stream)
#define XA_NT(NCT_)                                      \
  switch (nt) {                                          \
    case 1: XA_CASE(1, NCT_); break;                     \
    case 2: XA_CASE(2, NCT_); break;                     \
    case 3: XA_CASE(3, NCT_); break;                     \
    default: XA_CASE(4, NCT_); break;                    \
  }
  switch (H) {
    case 128: XA_NT(1); break;
    case 256: XA_NT(2); break;
    default: XA_NT(6); break;
  }
  if (rc) return rc;
  SPMM_LAUNCH_CHECK("spmm_xattn_fwd");
  return SPMM_OK;
}
