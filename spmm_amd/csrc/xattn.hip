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
#include <type_traits>
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
#define XA_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// The wait that makes an asynchronously loaded register valid takes that register as a read-write operand: every use of the loaded value
// is then a use of THIS statement's output, so no compiler-inserted copy, spill or re-ordered use of the register can sit ahead of the wait
// (the register's only other appearance is as the output of the load statement itself).
#define XA_VM_TIE(N, R) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(R) : "n"(N) : "memory")
#define XA_LGKM0_TIE(R) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R)::"memory")
#define XA_SB() __builtin_amdgcn_sched_barrier(0)

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Inline-asm memory operations: the compiler's waitcnt pass neither sees them nor waits for a pending LDS-DMA because of them;
// the caller places the waits (XA_LGKM0 / XA_VM) and fences the uses with XA_SB.
template <int OFF, typename T>
__device__ __forceinline__ void ds_rd(T& d, uint32_t addr) {                   // 16-byte LDS read
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ds_rd_tr(bf16x4& d, uint32_t addr) {           // transposing 8-byte LDS read
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ds_wr8(uint32_t addr, bf16x4 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// LDS-DMA of 16 bytes per lane: SGPR base + 32-bit VGPR byte offset -> wave-uniform LDS destination (M0) + lane * 16
__device__ __forceinline__ void glds16(uint32_t voff, const void* sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// stores in the same SGPR-base form (a 64-bit per-lane pointer kept live across the loop gets spilled, and every reload of it is
// a vmcnt(0) that drains the prefetch ring)
__device__ __forceinline__ void gl_st16(uint32_t voff, const void* sbase, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gl_st4(uint32_t voff, const void* sbase, float v) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
// asynchronous 16-byte global load into registers: SGPR base + VGPR byte offset + immediate
template <int IMM>
__device__ __forceinline__ void gl_ld16(bf16x8& d, uint32_t voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// ... into ACCUMULATOR registers (gfx950: one 512-entry file, loads may target either half, MFMA reads A operands from either): the
// prefetch ring lives beside the 192 accumulators and leaves the 256 architectural VGPRs to the attention core
template <int IMM>
__device__ __forceinline__ void gl_ld16a(bf16x8& d, uint32_t voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&a"(d) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}

// LDS image of a [64][H] bf16 panel (residual in, z / y out): 16-byte chunk c16 of row r sits at slot r * C16 + (c16 ^ (r & 15)).
template <int C16>
__device__ __forceinline__ int pslot(int row, int c16) { return row * C16 + (c16 ^ (row & 15)); }

template <int NT, int NCT>
struct XaLds {
  static constexpr int H = 128 * NCT;
  static constexpr int KVT = NT * 32 * ROWB;          // one [NT*32][64] head tile
  static constexpr int KVBUF = 4 * KVT;               // K(h0) V(h0) K(h1) V(h1)
  static constexpr int PANEL = 64 * H * 2;            // residual / z / y image
  static constexpr int R0 = 2 * KVBUF > PANEL ? 2 * KVBUF : PANEL;
  static constexpr int O_CTX = R0;                    // [64][128] bf16 context panel of the current head pair
  static constexpr int O_MB = O_CTX + 64 * 256;       // additive score bias per key (128 floats)
  static constexpr int O_RED = O_MB + 128 * 4;        // [2][4][64] row partial sums
  static constexpr int O_VEC = O_RED + 512 * 4;       // bias | gamma | beta, fp32 [3][H]
  static constexpr int TOTAL = O_VEC + 3 * H * 4;
};

// tools/ only (make XA_PROFILE=1): per-phase s_memtime stamps of wave 0 go to the LSE buffer instead of the log-sum-exp values
#ifdef XA_PROFILE
#define XA_STAMP(K) do { if (prof && tid == 0) prof[K] = __builtin_readcyclecounter(); } while (0)
#else
#define XA_STAMP(K) do { } while (0)
#endif

template <int NT, int NCT>   // NT = ceil(Lkv / 32) key tiles; H = 128 * NCT
__global__ __launch_bounds__(256, 1) void xattn_fwd_kernel(XattnP p) {
  using L = XaLds<NT, NCT>;
#ifdef XA_PROFILE
  unsigned long long* prof = p.LSE ? (unsigned long long*)p.LSE + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 : nullptr;
  p.LSE = nullptr;
  { const int tid = threadIdx.x; XA_STAMP(0); }
#endif
  constexpr int H = 128 * NCT, NS = NCT, WN = 32 * NCT, C16 = H / 8;
  constexpr int KVT = L::KVT, KVBUF = L::KVBUF;
  constexpr int NF = 8 * NCT;                    // Wo fragments per wave and step (1 KiB each)
  constexpr int D = NF < 12 ? NF : 12;           // ... of which D are in flight, in ACCUMULATOR registers (16 spill inside the loop: every reload drains the DMA)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
  char* kvb = smem;
  float* mb = (float*)(smem + L::O_MB);

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

  // Start-up order: the first head pair's K/V DMA and the first Q fragments are issued BEFORE the mask / bias vectors are fetched, so the
  // three round trips to memory overlap (the vectors' plain loads make the compiler wait for everything outstanding, which the barrier
  // below needs anyway).  Step 0's Q fragments are plain loads for the same reason: nothing of theirs is in flight past the barrier.
  bf16x8 qn[4];
  const char* Qb = (const char*)(p.Q + qrow * p.ldq + hsel * HD);
  const uint32_t qvo = (uint32_t)((myqc * p.ldq) * 2 + g * 16);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qn[kk] = *(const bf16x8*)(Qb + qvo + kk * 32);
  // Q fragments of the NEXT step (B operand: row = lane&31, d = (kk*2+g)*8 ..), by asm loads like everything else that is in
  // flight across the loop (a compiler-visible load would be waited for with vmcnt(0), draining the prefetch ring with it)
  auto load_q = [&](int s) {
    const char* qb = Qb + s * (2 * HD * 2);
    gl_ld16<0>(qn[0], qvo, qb); gl_ld16<32>(qn[1], qvo, qb); gl_ld16<64>(qn[2], qvo, qb); gl_ld16<96>(qn[3], qvo, qb);
  };

  // ---- K/V staging: lane-linear LDS image, swizzle on the source chunk (attn_tiles.h stage_head), as SGPR-base DMA
  const char* Kg = (const char*)(p.K + kvrow * p.ldk);
  const char* Vg = (const char*)(p.V + kvrow * p.ldv);
  uint32_t koff[NT], voff[NT];
  {
    const int lc = (tid & 7) ^ frot(tid >> 3);   // frot(row + 32 it) == frot(row)
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int row = (tid >> 3) + 32 * it;
      const int grow = row < Lkv ? row : Lkv - 1;
      koff[it] = (uint32_t)((grow * p.ldk + lc * 8) * 2);
      voff[it] = (uint32_t)((grow * p.ldv + lc * 8) * 2);
    }
  }
  auto stage_pair = [&](int s, int b) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const char* sk = Kg + (2 * s + hh) * (HD * 2);
      const char* sv = Vg + (2 * s + hh) * (HD * 2);
      const uint32_t dk = lds0 + b * KVBUF + (2 * hh) * KVT + wave * 1024;
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        glds16(koff[it], sk, dk + it * 4096);
        glds16(voff[it], sv, dk + KVT + it * 4096);
      }
    }
  };
  stage_pair(0, 0);
  const float neg2 = fmaxf(p.mask_neg * LOG2E, -3.4028234e38f);
  for (int j = tid; j < 128; j += 256)
    mb[j] = j < Lkv ? ((p.kmask == nullptr || p.kmask[(long)seq * p.Lkv + j]) ? 0.f : neg2) : -INFINITY;
  // Wo fragment ring: A operands from the fragment-ordered image, D x 1 KiB in flight per wave.  Fragment j = (kh*NCT + ct)*4 + kk of a
  // step feeds two MFMAs (row tiles 0 / 1).  The first D fragments of a step are issued right after the attention core's last MFMA
  // (they land under the context-panel exchange).  They are NOT carried across the loop's back edge: an asm-loaded register
  // that the compiler copies before its data has landed (phi copies) holds garbage -- seen with D = 12, tests caught it.
  bf16x8 ring[D];
  const uint32_t wvo = lane * 16;
  const char* Wb = (const char*)p.WoF + (long)wave * NF * 1024;
  auto issue = [&](auto jc, const char* wbase) {
    constexpr int j = decltype(jc)::value;
    constexpr int kh = j / (NCT * 4), ct = (j / 4) % NCT, kk = j % 4;
    gl_ld16<kk * 1024>(ring[j % D], wvo, wbase + (ct * 2 + kh) * 4096);
  };
  XA_VM(0);
  XA_LGKM0();                                    // the mask vector's LDS writes
  __builtin_amdgcn_s_barrier();
  XA_SB();
  XA_STAMP(1);

  // ---- loop-invariant LDS address parts of the attention core
  const int fr = frot(l31);
  uint32_t kb[4], vb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) kb[kk] = (uint32_t)(l31 * ROWB + (((kk * 2 + g) ^ fr) << 4));
  {
    const int i16 = lane & 15, j = lane >> 4;
    const int row0 = 4 * (j >> 1) + (i16 >> 2);
    const int col = (j & 1) * 16 + (i16 & 3) * 4;
    vb[0] = (uint32_t)(swz(row0, col >> 3) + (col & 7) * 2);
    vb[1] = (uint32_t)(swz(row0 + 8, col >> 3) + (col & 7) * 2);
    vb[2] = (uint32_t)(swz(row0, (col + 32) >> 3) + (col & 7) * 2);
    vb[3] = (uint32_t)(swz(row0 + 8, (col + 32) >> 3) + (col & 7) * 2);
  }
  const uint32_t mba = lds0 + L::O_MB + 16 * g;
  const uint32_t cva = lds0 + L::O_CTX + (tid >> 4) * 256 + (((tid & 15) ^ ((tid >> 4) & 15)) << 4);      // context panel, row-major reader

  // The accumulators start from the projection bias (lane (row, g) holds columns wave*WN + ct*32 + 8*gq + 4*g + {0..3} in
  // acc[a][ct][gq*4 ..]): the epilogue then only rounds them.
  f32x16 acc[2][NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x4 b4 = *(const f32x4*)(p.bo + wave * WN + ct * 32 + 8 * gq + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[0][ct][gq * 4 + j] = b4[j]; acc[1][ct][gq * 4 + j] = b4[j]; }
    }

  uint64_t seed_a = p.drop_a16 ? seed_mix(p.seed_ptr, p.salt_a) : 0;
  seed_a = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(seed_a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)seed_a);   // scalar registers

#pragma unroll 1
  for (int s = 0; s < NS; ++s) {                                     // (a real loop: unrolled, its bodies spill)
    const int cur = s & 1;
    const int h = 2 * s + hsel;
    if (s + 1 < NS) stage_pair(s + 1, cur ^ 1);                      // (a) next pair's K/V: in flight under the attention core
    XA_SB();
    // ------------------------------------------------------------------ (b) attention core of (row tile rt, head h)
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = qn[kk];
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));      // (real copies: qn is reloaded right away)
    if (s + 1 < NS) load_q(s + 1);
    XA_SB();
    const uint32_t kt = lds0 + cur * KVBUF + (2 * hsel) * KVT;
    uint32_t ka[4], va[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ka[i] = kb[i] + kt; va[i] = vb[i] + kt + KVT; }
    f32x16 st[NT];
    {
      bf16x8 kf[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) ds_rd<0>(kf[0][kk], ka[kk]);
      static_for<NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        XA_LGKM0();
        XA_SB();
        if constexpr (t + 1 < NT) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) ds_rd<(t + 1) * 4096>(kf[(t + 1) & 1][kk], ka[kk]);
        }
        st[t] = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) st[t] = MFMA32(kf[t & 1][kk], qf[kk], st[t]);
        XA_SB();
      });
    }
    // V^T fragments of key tile 0 start now and land under the softmax arithmetic
    bf16x4 vr[2][8];
    ds_rd_tr<0>(vr[0][0], va[0]); ds_rd_tr<0>(vr[0][1], va[1]); ds_rd_tr<0>(vr[0][2], va[2]); ds_rd_tr<0>(vr[0][3], va[3]);
    ds_rd_tr<2048>(vr[0][4], va[0]); ds_rd_tr<2048>(vr[0][5], va[1]); ds_rd_tr<2048>(vr[0][6], va[2]); ds_rd_tr<2048>(vr[0][7], va[3]);
    float mx = -INFINITY;
    static_for<NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      f32x4 bias[4];
      ds_rd<(t * 32 + 0) * 4>(bias[0], mba); ds_rd<(t * 32 + 8) * 4>(bias[1], mba);
      ds_rd<(t * 32 + 16) * 4>(bias[2], mba); ds_rd<(t * 32 + 24) * 4>(bias[3], mba);
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
    });
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
    if (p.LSE && myq < nvalid && g == 0) gl_st4((uint32_t)(myq * 4), p.LSE + ((long)seq * p.nH + h) * p.Lq + r0p, mx * LN2 + __logf(sum));
    if (p.drop_a16) {
      const uint32_t rowkey = drop_rowkey(seed_a, ((uint64_t)seq * p.nH + h) * p.Lq + qpos) + (uint32_t)(2 * g) * DROP_WEYL;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const uint32_t r0 = drop_pair(rowkey, t * 16 + 4 * gq), r1 = drop_pair(rowkey, t * 16 + 4 * gq + 1);   // (the lane's 2g is inside rowkey)
          st[t][gq * 4 + 0] = (r0 & 0xffffu) >= p.drop_a16 ? st[t][gq * 4 + 0] : 0.f;
          st[t][gq * 4 + 1] = (r0 >> 16) >= p.drop_a16 ? st[t][gq * 4 + 1] : 0.f;
          st[t][gq * 4 + 2] = (r1 & 0xffffu) >= p.drop_a16 ? st[t][gq * 4 + 2] : 0.f;
          st[t][gq * 4 + 3] = (r1 >> 16) >= p.drop_a16 ? st[t][gq * 4 + 3] : 0.f;
        }
    }
    f32x16 ot[2] = {zero16(), zero16()};
    static_for<NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const bf16x8 pf0 = pack8(st[t], 0), pf1 = pack8(st[t], 1);
      XA_LGKM0();
      XA_SB();
      if constexpr (t + 1 < NT) {
        constexpr int o = (t + 1) * 4096;
        bf16x4(&n)[8] = vr[(t + 1) & 1];
        ds_rd_tr<o>(n[0], va[0]); ds_rd_tr<o>(n[1], va[1]); ds_rd_tr<o>(n[2], va[2]); ds_rd_tr<o>(n[3], va[3]);
        ds_rd_tr<o + 2048>(n[4], va[0]); ds_rd_tr<o + 2048>(n[5], va[1]); ds_rd_tr<o + 2048>(n[6], va[2]); ds_rd_tr<o + 2048>(n[7], va[3]);
      }
      const bf16x4(&v)[8] = vr[t & 1];
      ot[0] = MFMA32(join8(v[0], v[1]), pf0, ot[0]);
      ot[1] = MFMA32(join8(v[2], v[3]), pf0, ot[1]);
      ot[0] = MFMA32(join8(v[4], v[5]), pf1, ot[0]);
      ot[1] = MFMA32(join8(v[6], v[7]), pf1, ot[1]);
      XA_SB();
    });
    const char* wbase = Wb + (long)s * (4 * NF * 1024);
    static_for<D>([&](auto jc) { issue(jc, wbase); });
    XA_SB();
    {
      const float osc = p.drop_a16 ? inv * p.scale_a : inv;
      int x = myq & 15;
      asm volatile("" : "+v"(x));                                    // (recomputed per step: eight hoisted addresses would stay live across the loop)
      const uint32_t rowb = lds0 + L::O_CTX + myq * 256 + g * 8;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int ch = hsel * 8 + dt * 4 + gq;                      // 16-B chunk of the [64][128] panel: d = dt*32 + 8*gq + 4*g ..
          ds_wr8(rowb + ((ch ^ x) << 4), to_bf16x4(ot[dt][gq * 4] * osc, ot[dt][gq * 4 + 1] * osc, ot[dt][gq * 4 + 2] * osc, ot[dt][gq * 4 + 3] * osc));
        }
    }
    XA_LGKM0();
    __builtin_amdgcn_s_barrier();                                    // barrier A: the context panel of this pair is complete
    XA_SB();
    XA_STAMP(2 + 2 * s);
    // ------------------------------------------------------------------ (c) output projection, K slice [128 s, 128 s + 128)
    // B operands: row = a*32 + l31, k = g*64 + kh*32 + kk*8 .. (head g of the pair) from the context panel.
    {
      int x = l31 & 15;
      asm volatile("" : "+v"(x));
      const uint32_t cfb = lds0 + L::O_CTX + l31 * 256;
      bf16x8 cf[2][4];
      auto load_cf = [&](int kh) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const uint32_t a0 = cfb + (((g * 8 + kh * 4 + kk) ^ x) << 4);
          ds_rd<0>(cf[0][kk], a0);
          ds_rd<8192>(cf[1][kk], a0);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { XA_LGKM0_TIE(cf[0][kk]); XA_LGKM0_TIE(cf[1][kk]); }
        XA_SB();
      };
      load_cf(0);
      static_for<NF>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int ct = (j / 4) % NCT, kk = j % 4;
        if constexpr (j == NCT * 4) load_cf(1);
        constexpr int younger = NF - 1 - j < D - 1 ? NF - 1 - j : D - 1;
        // j == 0: everything older has to be in anyway -- the next pair's K/V (issued a whole attention core ago), the next Q
        // fragments, this step's first D ring fragments; later waits count the ring fragments issued after the one needed
        // (the context stores / the residual DMA below are younger too: they only make some waits stricter)
        if constexpr (j == 0) XA_VM_TIE(0, ring[j % D]); else XA_VM_TIE(younger, ring[j % D]);
        XA_SB();
        acc[0][ct] = MFMA32(ring[j % D], cf[0][kk], acc[0][ct]);
        acc[1][ct] = MFMA32(ring[j % D], cf[1][kk], acc[1][ct]);
        XA_SB();
        if constexpr (j + D < NF) issue(std::integral_constant<int, j + D>{}, wbase);
        if constexpr (j == 3) {
          if (p.CTX) {                                               // the panel leaves row-major, 16 B per lane, 256 B per row
            // (read HERE, not earlier: a register filled by an asynchronous asm read must not stay live across other asynchronous
            // loads -- with the reads ahead of the fragment ring the first of the four came out clobbered)
            u32x4 cv[4];
            ds_rd<0>(cv[0], cva); ds_rd<4096>(cv[1], cva); ds_rd<8192>(cv[2], cva); ds_rd<12288>(cv[3], cva);
            XA_LGKM0_TIE(cv[0]); XA_LGKM0_TIE(cv[1]); XA_LGKM0_TIE(cv[2]); XA_LGKM0_TIE(cv[3]);
            XA_SB();
            const char* cb = (const char*)(p.CTX + qrow * p.ldc + s * 128);
            int tl = tid;
            asm volatile("" : "+v"(tl));                             // (offsets recomputed here: hoisted out of the loop they get spilled)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int row = (tl >> 4) + 16 * i;
              if (row < nvalid) gl_st16((uint32_t)((row * p.ldc + (tl & 15) * 8) * 2), cb, cv[i]);
            }
          }
          XA_SB();
        }
      });
    }
    XA_LGKM0();
    __builtin_amdgcn_s_barrier();                                    // barrier B
    XA_SB();
    XA_STAMP(3 + 2 * s);
  }

  // ---------------------------------------------------------------------------------------------------- epilogue
  // accumulator layout: lane (row = a*32 + l31, g) holds columns wave*WN + ct*32 + 8*gq + 4*g + {0..3} in acc[a][ct][gq*4 ..]
  //
  // Two phases.  (1) x (the accumulators started from the bias), rounded to bf16 exactly as the projection GEMM of the composite rounds
  // it, goes to a row-major LDS image over the dead K/V buffers; the accumulators are free after that.  (2) The LayerNorm row kernel's
  // body (rowops.hip ln_fwd16_kernel) on that image: half a wave owns a row (32 lanes x NCH 16-byte chunks), a wave walks its 16 rows
  // two at a time; hidden dropout (the counter hash of spmm_ln_fwd: spmm_ln_bwd regenerates the mask), + residual straight from HBM
  // (whole rows; all of a wave's residual loads are issued before phase 1 and land under it), two-pass statistics with DPP reductions
  // inside the half wave, z and y stored as 512-byte row segments.  One wave per SIMD: this phase is VALU-issue-bound, so its
  // arithmetic is written on float pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two elements per issue slot).
  constexpr int NCH = (C16 + 31) / 32;                               // 16-byte chunks of a row per lane: chunk l + 32 k
  constexpr bool RAGGED = C16 % 32 != 0;                             // H = 128: lanes 16..31 of a half wave own no chunk
  const int hw = lane >> 5;                                          // which of the wave's two rows
  u32x4 rres[8][NCH];
  {
    const bf16* Rg = p.R + qrow * p.ldr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = wave * 16 + 2 * i + hw, rr = row < nvalid ? row : nvalid - 1;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int ch = l31 + 32 * k;
        rres[i][k] = *(const u32x4*)(Rg + (long)rr * p.ldr + (ch < C16 ? ch : 0) * 8);
      }
    }
  }
  f32x2 gm[NCH][4], bt[NCH][4];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = ((l31 + 32 * k) < C16 ? (l31 + 32 * k) : 0) * 8;
#pragma unroll
    for (int m = 0; m < 4; ++m) { gm[k][m] = *(const f32x2*)(p.gamma + c + 2 * m); bt[k][m] = *(const f32x2*)(p.beta + c + 2 * m); }
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int c = wave * WN + ct * 32 + 8 * gq + 4 * g;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = a * 32 + l31;
        *(bf16x4*)(kvb + pslot<C16>(row, c >> 3) * 16 + g * 8) =
            to_bf16x4(acc[a][ct][gq * 4], acc[a][ct][gq * 4 + 1], acc[a][ct][gq * 4 + 2], acc[a][ct][gq * 4 + 3]);
      }
    }
  // (a raw barrier: __syncthreads would also wait for the residual loads, which phase 2 wants in flight)
  XA_LGKM0();
  __builtin_amdgcn_s_barrier();
  XA_SB();
  XA_STAMP(20);
  const uint64_t seed_h = p.drop_h16 ? seed_mix(p.seed_ptr, p.salt_h) : 0;
  const uint32_t lw = (uint32_t)(l31 * 4) * DROP_WEYL;               // the lane's part of the pair index (column >> 1) = (l + 32 k) * 4 + m
#pragma unroll                                                       // (unrolled: rres[i] must stay in registers)
  for (int i = 0; i < 8; ++i) {
    if (wave * 16 + 2 * i >= nvalid) break;                          // wave-uniform: both rows of the pair lie past the sequence
    const int row = wave * 16 + 2 * i + hw;
    f32x2 z[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int ch = l31 + 32 * k;
      const u32x4 xv = *(const u32x4*)(kvb + pslot<C16>(row, ch < C16 ? ch : 0) * 16);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x2 x2 = {__builtin_bit_cast(float, xv[m] << 16), __builtin_bit_cast(float, xv[m] & 0xffff0000u)};
        const f32x2 r2 = {__builtin_bit_cast(float, rres[i][k][m] << 16), __builtin_bit_cast(float, rres[i][k][m] & 0xffff0000u)};
        z[k][m] = p.drop_h16 ? x2 : x2 + r2;                         // (with dropout the residual joins after the mask below)
      }
    }
    if (p.drop_h16) {
      const uint32_t rk = drop_rowkey(seed_h, (uint64_t)(p.row_base + qrow + row)) + lw;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const uint32_t r = drop_pair(rk, 128 * k + m);
          const f32x2 keep = {(r & 0xffffu) >= p.drop_h16 ? p.scale_h : 0.f, (r >> 16) >= p.drop_h16 ? p.scale_h : 0.f};
          const f32x2 r2 = {__builtin_bit_cast(float, rres[i][k][m] << 16), __builtin_bit_cast(float, rres[i][k][m] & 0xffff0000u)};
          z[k][m] = z[k][m] * keep + r2;                             // one packed FMA: dropout scale / zero and the residual
        }
    }
    f32x2 sm2 = {0.f, 0.f};
    u32x4 zb[NCH];                                                   // z as stored for the backward (rounded before it is centred)
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if (RAGGED && l31 + 32 * k >= C16) {
#pragma unroll
        for (int m = 0; m < 4; ++m) z[k][m] = f32x2{0.f, 0.f};
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) { sm2 += z[k][m]; zb[k][m] = pk2(z[k][m].x, z[k][m].y); }
    }
    const float mean = half_sum(sm2.x + sm2.y) * (1.f / H);
    const f32x2 mean2 = {mean, mean};
    f32x2 ss2 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        z[k][m] -= mean2;                                            // centred from here on
        const f32x2 d = (RAGGED && l31 + 32 * k >= C16) ? f32x2{0.f, 0.f} : z[k][m];
        ss2 += d * d;
      }
    const float var = half_sum(ss2.x + ss2.y) * (1.f / H);
    float rstd = rsqrtf(var + p.eps);
    if (!(var + p.eps > 0.f)) rstd = 0.f;
    const f32x2 rstd2 = {rstd, rstd};
    if (row < nvalid) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int ch = l31 + 32 * k;
        if (!RAGGED || ch < C16) {
          u32x4 o;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const f32x2 y2 = z[k][m] * rstd2 * gm[k][m] + bt[k][m];
            o[m] = pk2(y2.x, y2.y);
          }
          *(u32x4*)(p.Y + (qrow + row) * p.ldy + ch * 8) = o;
          if (p.Z) *(u32x4*)(p.Z + (qrow + row) * p.ldz + ch * 8) = zb[k];
        }
      }
      if (p.mean && l31 == 0) { p.mean[qrow + row] = mean; p.rstd[qrow + row] = rstd; }
    }
  }
  XA_STAMP(23);
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
constexpr int xattn_lds() { return XaLds<NT, NCT>::TOTAL; }

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
