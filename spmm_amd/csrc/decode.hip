// Single-query attention over a key/value cache for autoregressive PV -> SMILES beam decoding.
// Replaces, for the newest position of every beam, xbert.py:305-354 (scores, 1/sqrt(d), softmax, context, head merge) of
// both BertSelfAttention flavours: self-attention reads the per-layer cache of the beam's own ancestry, cross-attention
// reads the PV keys/values computed once per molecule and shared by its k beams.
//
// Shape of the work: R rows (molecules x beams) x nH heads, one query each, Lkv <= 256 keys, d = 64.  It is an HBM-bound
// gather (2 x Lkv x 128 B per row-head), so there is no MFMA here: one wave64 per (row, head); 8 lanes share a key (each
// reads 16 B of its 128-B row: whole cache lines per wave instruction), fp32 dot products and softmax with wave shuffles,
// scores parked in LDS between the two passes over the keys.  Beams are never physically reordered: `anc[r, j]` names the cache row that holds position j of
// row r's hypothesis (updated by the host-side beam bookkeeping with one small gather per step).
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

struct DecAttnP {
  const bf16* q; long ldq;
  const bf16* K; const bf16* V; long seq_stride, tok_stride, head_stride;   // element (s, j, h, d) at s*seq_stride + j*tok_stride + h*head_stride + d
  const int* anc; int anc_ld; int kv_div; int group; int nblocks;
  bf16* out; long ldo;
  int R, nH, Lkv; float scale;
  const int* t_ptr;             // optional device step index: the cache holds positions 0..*t_ptr, i.e. Lkv = *t_ptr + 1 (graph replay)
  const int* rowmap;            // optional: cache row that row r's newest position is written to (identity when null)
  const bf16* knew; const bf16* vnew; long ldn;   // optional: key / value of the NEWEST position (Lkv - 1) of every row, not yet in the cache: read from
  //                                                 here (row stride ldn) and written to the cache row by the wave that owns the (row, head)
};

__global__ __launch_bounds__(256) void decode_attn_kernel(DecAttnP p) {
  __shared__ float ssc[4][256];                      // scores of the wave's (row, head), one wave per LDS row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Beams of one molecule share most of their keys/values (all of them in cross-attention), so their waves are placed
  // together: consecutive logical waves are the `group` beams of one (molecule, head), and consecutive logical blocks go to
  // the same XCD (blockIdx round-robins over the 8 XCDs), so the shared lines hit in that XCD's L2 / the CU's L1.
  const int per_xcd = (p.nblocks + 7) >> 3;
  const long lb = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const long gw = lb * 4 + wave;
  if (lb >= p.nblocks || gw >= (long)p.R * p.nH) return;
  const int per_mol = p.nH * p.group;
  const int n = (int)(gw / per_mol), rem = (int)(gw - (long)n * per_mol);
  const int h = rem / p.group, r = n * p.group + (rem - h * p.group);
  // 8 lanes share one key: lane (g, c) reads the 16-B chunk c of the head's 128-B K/V row of key 8*i + g, so one wave
  // instruction covers 8 whole cache lines (a lane-per-key layout touches 64 lines for the same bytes).
  const int g = lane >> 3, c = lane & 7;
  float qf[8];
  {
    const bf16x8 qv = *(const bf16x8*)(p.q + (long)r * p.ldq + h * 64 + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e] * p.scale;
  }
  const int* anc_row = p.anc ? p.anc + (long)r * p.anc_ld : nullptr;
  const long own = (long)(r / max(p.kv_div, 1)) * p.seq_stride + h * p.head_stride + c * 8;
  const int Lkv = p.t_ptr ? min(*p.t_ptr + 1, p.Lkv) : p.Lkv;
  const int niter = (Lkv + 7) >> 3;
  float mx = -INFINITY;
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    float part = 0.f;
    if (j < Lkv) {
      const long off = (anc_row ? (long)anc_row[j] * p.seq_stride + h * p.head_stride + c * 8 : own) + (long)j * p.tok_stride;
      const bf16x8 kv = *(const bf16x8*)(p.K + off);
#pragma unroll
      for (int e = 0; e < 8; ++e) part += (float)kv[e] * qf[e];
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (j < Lkv) {
      if (c == 0) ssc[wave][j] = part;
      mx = fmaxf(mx, part);
    }
  }
  mx = wave_max(mx);
  __builtin_amdgcn_wave_barrier();
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float sum = 0.f;
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    if (j < Lkv) {
      const float e = __expf(ssc[wave][j] - mx);
      sum += e;
      const float pe = (float)(bf16)e;               // the tiled training kernel feeds bf16 probabilities to the PV MFMA
      const long off = (anc_row ? (long)anc_row[j] * p.seq_stride + h * p.head_stride + c * 8 : own) + (long)j * p.tok_stride;
      const bf16x8 vv = *(const bf16x8*)(p.V + off);
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] += pe * (float)vv[d];
    }
  }
  // combine the 8 key slots (lanes with equal c): xor 8, 16, 32
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    sum += __shfl_xor(sum, o, 64);
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] += __shfl_xor(acc[d], o, 64);
  }
  if (g == 0) {
    const float inv = 1.f / sum;
    bf16x8 o;
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = (bf16)(acc[d] * inv);
    *(bf16x8*)(p.out + (long)r * p.ldo + h * 64 + c * 8) = o;
  }
}

// The same attention with ONE wave per (molecule, head) serving all G beams of the molecule.  In cross-attention the beams read the
// same keys/values (kv_div = G); in self-attention their ancestries coincide except for the last few positions (beam search
// coalesces), so a key row is loaded once -- for beam 0 -- and reused by every beam whose ancestor at that position is the same cache
// row; only the lanes of differing ancestors issue their own load.  The per-beam kernel left that sharing to the caches: same HBM
// bytes, G times the load instructions and L1 / L2 requests.  Ancestor indices are staged in LDS first (coalesced), so the key loads
// do not wait for a dependent index load.  Arithmetic per (row, head) is the per-beam kernel's (fp32 dots over 8 lanes x 8 elements,
// two passes, bf16-rounded probabilities in front of V).
// Round 4: the kernel waits (SQ_WAIT_ANY 0.66 of wave cycles), so its loads are issued in batches: the whole prologue (query rows and
// ancestry entries) before anything is consumed, and -- where some beam of an iteration sits on another cache row -- every beam's row
// back to back behind ONE wave-uniform branch instead of a load behind its own divergent branch per beam (one memory latency each).
// 4 waves per SIMD is both what the LDS arrays allow and what the register budget is held to (amdgpu_waves_per_eu: left alone the batched
// form takes 168 registers, three waves, and loses what the batching wins).  5 000 rows x 12 heads, last six positions on own rows:
// 60.5 -> 48.1 us at 25 keys, 83.7 -> 64.5 at 50, 130 -> 105 at 100 (tools/bench_decode_attn.py).
template <int G, bool ALLSAME>   // ALLSAME: no ancestry table, every beam of the molecule reads the same key/value rows (cross-attention)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G <= 5 ? 4 : 3, 4))) void decode_attn_group_kernel(DecAttnP p) {
  __shared__ float ssc[4][G][256];                   // scores of the wave's G (row, head) pairs
  __shared__ int sanc[4][G][256];                    // cache row of position j for each beam (16 waves per CU at G = 5; sized by Lkv with
  //                                                    run-time strides -- 32 waves per CU -- it measured SLOWER: 99 vs 84 us per launch,
  //                                                    and 8 waves per CU 134)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_xcd = (p.nblocks + 7) >> 3;
  const long lb = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const long gw = lb * 4 + wave;
  const int nmol = p.R / G;
  if (lb >= p.nblocks || gw >= (long)nmol * p.nH) return;
  const int n = (int)(gw / p.nH), h = (int)(gw - (long)n * p.nH);
  const int g = lane >> 3, c = lane & 7;
  const int Lkv = p.t_ptr ? min(*p.t_ptr + 1, p.Lkv) : p.Lkv;
  const int niter = (Lkv + 7) >> 3;
  float qf[G][8];
  {
    // every load of the prologue is issued before the first one is consumed (query rows, then the ancestry entries lane, lane + 64, ...
    // of every beam: a loop that loads and stores one entry at a time waits out a memory latency per entry -- 5 to 10 of them in a row)
    bf16x8 qv[G];
    int av[ALLSAME ? 1 : G][4];
#pragma unroll
    for (int b = 0; b < G; ++b) qv[b] = *(const bf16x8*)(p.q + (long)(n * G + b) * p.ldq + h * 64 + c * 8);
#pragma unroll
    for (int b = 0; b < (ALLSAME ? 1 : G); ++b)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u, r = n * G + b;
        av[b][u] = p.anc ? ((j < Lkv) ? p.anc[(long)r * p.anc_ld + j] : 0) : r / max(p.kv_div, 1);
      }
#pragma unroll
    for (int b = 0; b < (ALLSAME ? 1 : G); ++b)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < niter * 8) sanc[wave][b][lane + 64 * u] = av[b][u];
#pragma unroll
    for (int b = 0; b < G; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[b][e] = (float)qv[b][e] * p.scale;
  }
  __builtin_amdgcn_wave_barrier();
  const long hoff = h * p.head_stride + c * 8;
  float mx[G];
#pragma unroll
  for (int b = 0; b < G; ++b) mx[b] = -INFINITY;
  // one key's partial dot products for beam B from row KB (8 lanes x 8 elements, reduced over the key's 8 lanes)
#define DA_SCORE(B, KB)                                                                                          \
  do {                                                                                                           \
    float part = 0.f;                                                                                            \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) part += (float)(KB)[e] * qf[B][e];                            \
    part += dpp_f<0xB1>(part); part += dpp_f<0x4E>(part); part += dpp_f<0x141>(part);                            \
    if (valid) {                                                                                                 \
      if (c == 0) ssc[wave][B][j] = part;                                                                        \
      mx[B] = fmaxf(mx[B], part);                                                                                \
    }                                                                                                            \
  } while (0)
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    const bool valid = j < Lkv;
    const long joff = (long)j * p.tok_stride + hoff;
    int ab[G];
#pragma unroll
    for (int b = 0; b < G; ++b) ab[b] = (ALLSAME && b > 0) ? 0 : sanc[wave][b][j];
    const int a0 = ab[0];
    bool diff = false;
    if (!ALLSAME) {
#pragma unroll
      for (int b = 1; b < G; ++b) diff |= ab[b] != a0;
    }
    bf16x8 k0;
#pragma unroll
    for (int e = 0; e < 8; ++e) k0[e] = (bf16)0.f;
    if (valid) k0 = *(const bf16x8*)(p.K + (long)a0 * p.seq_stride + joff);
    if (!ALLSAME && __any(valid && diff)) {
      // some beam of some key of this iteration sits on another cache row (the last few positions of a hypothesis): every beam's row is
      // loaded, back to back with no control flow in between (a load per differing beam behind its own branch waits out one memory
      // latency per beam; the rows that do not differ hit the line beam 0 just fetched)
      bf16x8 kb[G];
#pragma unroll
      for (int b = 1; b < G; ++b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) kb[b][e] = (bf16)0.f;
      }
      if (valid) {
#pragma unroll
        for (int b = 1; b < G; ++b) kb[b] = *(const bf16x8*)(p.K + (long)ab[b] * p.seq_stride + joff);
      }
      DA_SCORE(0, k0);
#pragma unroll
      for (int b = 1; b < G; ++b) DA_SCORE(b, kb[b]);
    } else {
#pragma unroll
      for (int b = 0; b < G; ++b) DA_SCORE(b, k0);
    }
  }
#undef DA_SCORE
#pragma unroll
  for (int b = 0; b < G; ++b) mx[b] = wave_max(mx[b]);
  __builtin_amdgcn_wave_barrier();
  float acc[G][8], sum[G];
#pragma unroll
  for (int b = 0; b < G; ++b) {
    sum[b] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[b][d] = 0.f;
  }
#define DA_ACC(B, VB)                                                                                            \
  do {                                                                                                           \
    const float e_ = valid ? __expf(ssc[wave][B][j] - mx[B]) : 0.f;                                              \
    sum[B] += e_;                                                                                                \
    const float pe_ = (float)(bf16)e_;             /* the tiled training kernel feeds bf16 probabilities to the PV MFMA */ \
    _Pragma("unroll") for (int d = 0; d < 8; ++d) acc[B][d] += pe_ * (float)(VB)[d];                            \
  } while (0)
#pragma unroll 1
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    const bool valid = j < Lkv;
    const long joff = (long)j * p.tok_stride + hoff;
    int ab[G];
#pragma unroll
    for (int b = 0; b < G; ++b) ab[b] = (ALLSAME && b > 0) ? 0 : sanc[wave][b][j];
    const int a0 = ab[0];
    bool diff = false;
    if (!ALLSAME) {
#pragma unroll
      for (int b = 1; b < G; ++b) diff |= ab[b] != a0;
    }
    bf16x8 v0;
#pragma unroll
    for (int d = 0; d < 8; ++d) v0[d] = (bf16)0.f;
    if (valid) v0 = *(const bf16x8*)(p.V + (long)a0 * p.seq_stride + joff);
    if (!ALLSAME && __any(valid && diff)) {
      bf16x8 vb[G];
#pragma unroll
      for (int b = 1; b < G; ++b) {
#pragma unroll
        for (int d = 0; d < 8; ++d) vb[b][d] = (bf16)0.f;
      }
      if (valid) {
#pragma unroll
        for (int b = 1; b < G; ++b) vb[b] = *(const bf16x8*)(p.V + (long)ab[b] * p.seq_stride + joff);
      }
      DA_ACC(0, v0);
#pragma unroll
      for (int b = 1; b < G; ++b) DA_ACC(b, vb[b]);
    } else {
#pragma unroll
      for (int b = 0; b < G; ++b) DA_ACC(b, v0);
    }
  }
#undef DA_ACC
#pragma unroll
  for (int b = 0; b < G; ++b) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {               // combine the 8 key slots (lanes with equal c)
      sum[b] += __shfl_xor(sum[b], o, 64);
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[b][d] += __shfl_xor(acc[b][d], o, 64);
    }
    if (g == 0) {
      const float inv = 1.f / sum[b];
      bf16x8 o;
#pragma unroll
      for (int d = 0; d < 8; ++d) o[d] = (bf16)(acc[b][d] * inv);
      *(bf16x8*)(p.out + (long)(n * G + b) * p.ldo + h * 64 + c * 8) = o;
    }
  }
}

// The same kernel with beam 0's row of every iteration -- the row most beams of most iterations read -- arriving through a per-wave LDS ring
// filled by LDS-DMA DA_DEPTH iterations ahead (global_load_lds: a gather of 16 B per lane that needs no registers, so the bytes a wave keeps
// in flight are bounded by LDS, not by the register budget that holds the plain kernel at one or two loads per wave).  Score / index arrays
// for at most 128 keys (the decoder's 103-position caches) so that the ring fits beside them at four workgroups per CU (64-bit per-lane
// DMA addresses: after a compaction the ancestry table names cache rows beyond the launch's own row count).  Rows of differing beams still come by ordinary loads (rare: the last few positions).
constexpr int DA_DEPTH = 4;
template <int G, bool ALLSAME>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G <= 5 ? 4 : 3, 4))) void decode_attn_dma_kernel(DecAttnP p) {
  __shared__ float ssc[4][G][128];                   // scores of the wave's G (row, head) pairs
  __shared__ int sanc[4][G][128];                    // cache row of position j for each beam
  __shared__ __attribute__((aligned(16))) char sring[4][DA_DEPTH][1024];   // beam 0's rows of DA_DEPTH iterations: lane l's 16 bytes at l * 16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_xcd = (p.nblocks + 7) >> 3;
  const long lb = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const long gw = lb * 4 + wave;
  const int nmol = p.R / G;
  if (lb >= p.nblocks || gw >= (long)nmol * p.nH) return;
  const int n = (int)(gw / p.nH), h = (int)(gw - (long)n * p.nH);
  const int g = lane >> 3, c = lane & 7;
  const int Lkv = p.t_ptr ? min(*p.t_ptr + 1, p.Lkv) : p.Lkv;
  const int niter = (Lkv + 7) >> 3;
  float qf[G][8];
  {
    // every load of the prologue is issued before the first one is consumed (query rows, then the ancestry entries lane, lane + 64, ...
    // of every beam: a loop that loads and stores one entry at a time waits out a memory latency per entry -- 5 to 10 of them in a row)
    bf16x8 qv[G];
    int av[ALLSAME ? 1 : G][2];
#pragma unroll
    for (int b = 0; b < G; ++b) qv[b] = *(const bf16x8*)(p.q + (long)(n * G + b) * p.ldq + h * 64 + c * 8);
#pragma unroll
    for (int b = 0; b < (ALLSAME ? 1 : G); ++b)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = lane + 64 * u, r = n * G + b;
        av[b][u] = p.anc ? ((j < Lkv) ? p.anc[(long)r * p.anc_ld + j] : 0) : r / max(p.kv_div, 1);
      }
#pragma unroll
    for (int b = 0; b < (ALLSAME ? 1 : G); ++b)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (lane + 64 * u < niter * 8) sanc[wave][b][lane + 64 * u] = av[b][u];
#pragma unroll
    for (int b = 0; b < G; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[b][e] = (float)qv[b][e] * p.scale;
  }
  __builtin_amdgcn_wave_barrier();
  const long hoff = h * p.head_stride + c * 8;
  const uint32_t ring0 = (uint32_t)(uintptr_t)(LDS_AS char*)&sring[wave][0][0];
  // LDS-DMA of beam 0's row of iteration I (keys past the end re-fetch the last valid one; they are masked where they are used)
#define DA_DMA(I, SRC)                                                                                           \
  do {                                                                                                           \
    int j_ = (I) * 8 + g;                                                                                        \
    j_ = j_ < Lkv ? j_ : Lkv - 1;                                                                                \
    const bf16* src_ = (SRC) + ((long)sanc[wave][0][j_] * p.seq_stride + (long)j_ * p.tok_stride + hoff);        \
    const uint32_t dst_ = __builtin_amdgcn_readfirstlane(ring0 + (uint32_t)(((I) % DA_DEPTH) * 1024));           \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_), "s"(dst_) : "memory", "m0"); \
  } while (0)
  // iteration I's row has landed: at most min(DA_DEPTH - 1, niter - 1 - I) newer DMAs may still be in flight (vmcnt retires in order)
#define DA_LANDED(I)                                                                                             \
  do {                                                                                                           \
    const int newer_ = niter - 1 - (I);                                                                          \
    if (newer_ >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                            \
    else if (newer_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                       \
    else if (newer_ == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                       \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
  } while (0)
  static_assert(DA_DEPTH == 4, "DA_LANDED counts up to three newer loads");
  for (int i = 0; i < DA_DEPTH && i < niter; ++i) DA_DMA(i, p.K);
  float mx[G];
#pragma unroll
  for (int b = 0; b < G; ++b) mx[b] = -INFINITY;
  // one key's partial dot products for beam B from row KB (8 lanes x 8 elements, reduced over the key's 8 lanes)
#define DA_SCORE(B, KB)                                                                                          \
  do {                                                                                                           \
    float part = 0.f;                                                                                            \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) part += (float)(KB)[e] * qf[B][e];                            \
    part += dpp_f<0xB1>(part); part += dpp_f<0x4E>(part); part += dpp_f<0x141>(part);                            \
    if (valid) {                                                                                                 \
      if (c == 0) ssc[wave][B][j] = part;                                                                        \
      mx[B] = fmaxf(mx[B], part);                                                                                \
    }                                                                                                            \
  } while (0)
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    const bool valid = j < Lkv;
    const long joff = (long)j * p.tok_stride + hoff;
    int ab[G];
#pragma unroll
    for (int b = 0; b < G; ++b) ab[b] = (ALLSAME && b > 0) ? 0 : sanc[wave][b][j];
    const int a0 = ab[0];
    bool diff = false;
    if (!ALLSAME) {
#pragma unroll
      for (int b = 1; b < G; ++b) diff |= ab[b] != a0;
    }
    DA_LANDED(i);
    const bf16x8 k0 = *(const bf16x8*)&sring[wave][i % DA_DEPTH][lane * 16];
    if (!ALLSAME && __any(valid && diff)) {
      // some beam of some key of this iteration sits on another cache row (the last few positions of a hypothesis): every beam's row is
      // loaded, back to back with no control flow in between (a load per differing beam behind its own branch waits out one memory
      // latency per beam; the rows that do not differ hit the line beam 0 just fetched)
      bf16x8 kb[G];
#pragma unroll
      for (int b = 1; b < G; ++b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) kb[b][e] = (bf16)0.f;
      }
      if (valid) {
#pragma unroll
        for (int b = 1; b < G; ++b) kb[b] = *(const bf16x8*)(p.K + (long)ab[b] * p.seq_stride + joff);
      }
      DA_SCORE(0, k0);
#pragma unroll
      for (int b = 1; b < G; ++b) DA_SCORE(b, kb[b]);
    } else {
#pragma unroll
      for (int b = 0; b < G; ++b) DA_SCORE(b, k0);
    }
    if (i + DA_DEPTH < niter) DA_DMA(i + DA_DEPTH, p.K);      // (the slot's row is in registers: k0 was consumed above)
  }
#undef DA_SCORE
  for (int i = 0; i < DA_DEPTH && i < niter; ++i) DA_DMA(i, p.V);   // the ring is free: every key row has been consumed
#pragma unroll
  for (int b = 0; b < G; ++b) mx[b] = wave_max(mx[b]);
  __builtin_amdgcn_wave_barrier();
  float acc[G][8], sum[G];
#pragma unroll
  for (int b = 0; b < G; ++b) {
    sum[b] = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[b][d] = 0.f;
  }
#define DA_ACC(B, VB)                                                                                            \
  do {                                                                                                           \
    const float e_ = valid ? __expf(ssc[wave][B][j] - mx[B]) : 0.f;                                              \
    sum[B] += e_;                                                                                                \
    const float pe_ = (float)(bf16)e_;             /* the tiled training kernel feeds bf16 probabilities to the PV MFMA */ \
    _Pragma("unroll") for (int d = 0; d < 8; ++d) acc[B][d] += pe_ * (float)(VB)[d];                            \
  } while (0)
#pragma unroll 1
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    const bool valid = j < Lkv;
    const long joff = (long)j * p.tok_stride + hoff;
    int ab[G];
#pragma unroll
    for (int b = 0; b < G; ++b) ab[b] = (ALLSAME && b > 0) ? 0 : sanc[wave][b][j];
    const int a0 = ab[0];
    bool diff = false;
    if (!ALLSAME) {
#pragma unroll
      for (int b = 1; b < G; ++b) diff |= ab[b] != a0;
    }
    DA_LANDED(i);
    const bf16x8 v0 = *(const bf16x8*)&sring[wave][i % DA_DEPTH][lane * 16];
    if (!ALLSAME && __any(valid && diff)) {
      bf16x8 vb[G];
#pragma unroll
      for (int b = 1; b < G; ++b) {
#pragma unroll
        for (int d = 0; d < 8; ++d) vb[b][d] = (bf16)0.f;
      }
      if (valid) {
#pragma unroll
        for (int b = 1; b < G; ++b) vb[b] = *(const bf16x8*)(p.V + (long)ab[b] * p.seq_stride + joff);
      }
      DA_ACC(0, v0);
#pragma unroll
      for (int b = 1; b < G; ++b) DA_ACC(b, vb[b]);
    } else {
#pragma unroll
      for (int b = 0; b < G; ++b) DA_ACC(b, v0);
    }
    if (i + DA_DEPTH < niter) DA_DMA(i + DA_DEPTH, p.V);
  }
#undef DA_ACC
#undef DA_DMA
#undef DA_LANDED
#pragma unroll
  for (int b = 0; b < G; ++b) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {               // combine the 8 key slots (lanes with equal c)
      sum[b] += __shfl_xor(sum[b], o, 64);
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[b][d] += __shfl_xor(acc[b][d], o, 64);
    }
    if (g == 0) {
      const float inv = 1.f / sum[b];
      bf16x8 o;
#pragma unroll
      for (int d = 0; d < 8; ++d) o[d] = (bf16)(acc[b][d] * inv);
      *(bf16x8*)(p.out + (long)(n * G + b) * p.ldo + h * 64 + c * 8) = o;
    }
  }
}

// the newest position's key / value rows into the cache, in front of the attention launch (one launch for both; folding it into the
// attention kernel itself -- the rows read from the projection output, written by the wave that owns the (row, head) -- was built and
// measured: 40 more live registers or an exposed load between the two passes, 1.61 -> 1.78-1.92 ms of decode_attn per position)
__global__ __launch_bounds__(256) void cache_write_kernel(DecAttnP p, int jn) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;                 // 16-byte chunk of a row
  const int per_row = p.nH * 8;
  if (i >= (long)p.R * per_row) return;
  const long r = i / per_row;
  const int ch = (int)(i - r * per_row), h = ch >> 3, c = (ch & 7) * 8;
  const int j = p.t_ptr ? min(*p.t_ptr, p.Lkv - 1) : jn;
  const long dst = (long)(p.rowmap ? p.rowmap[r] : (int)r) * p.seq_stride + (long)j * p.tok_stride + h * p.head_stride + c;
  *(bf16x8*)((bf16*)p.K + dst) = *(const bf16x8*)(p.knew + r * p.ldn + h * 64 + c);
  *(bf16x8*)((bf16*)p.V + dst) = *(const bf16x8*)(p.vnew + r * p.ldn + h * 64 + c);
}

template <int G>
void launch_group(DecAttnP p, hipStream_t stream) {
  const long waves = (long)(p.R / G) * p.nH;
  p.nblocks = (int)((waves + 3) / 4);
  static const bool no_dma = getenv("SPMM_DECODE_NO_DMA") != nullptr;          // (debugging aid, like SPMM_DECODE_PER_BEAM)
  if (!no_dma && p.Lkv <= 128) {
    if (p.anc) hipLaunchKernelGGL((decode_attn_dma_kernel<G, false>), dim3((unsigned)((p.nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((decode_attn_dma_kernel<G, true>), dim3((unsigned)((p.nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
    return;
  }
  if (p.anc) hipLaunchKernelGGL((decode_attn_group_kernel<G, false>), dim3((unsigned)((p.nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((decode_attn_group_kernel<G, true>), dim3((unsigned)((p.nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
}

}  // namespace

extern "C" int spmm_decode_attn(const void* q, long ldq, const void* K, const void* V, long seq_stride, long tok_stride, long head_stride,
                                const int* anc, int anc_ld, int kv_div, int group, void* out, long ldo, int R, int nH, int Lkv,
                                float scale, const int* t_ptr, const void* knew, const void* vnew, long ldn, const int* rowmap,
                                hipStream_t stream) {
  SPMM_CHECK_SHAPE((knew == nullptr) == (vnew == nullptr) && (knew == nullptr || (anc != nullptr && ldn >= (long)nH * 64 && ldn % 8 == 0)),
                   "spmm_decode_attn: knew / vnew come together, with an ancestry table, ldn=%ld a multiple of 8 >= nH*64", ldn);
  SPMM_CHECK_SHAPE(R > 0 && nH > 0 && Lkv > 0 && Lkv <= 256, "spmm_decode_attn: R=%d nH=%d Lkv=%d (Lkv <= 256)", R, nH, Lkv);
  SPMM_CHECK_SHAPE((anc != nullptr && anc_ld >= Lkv) || (anc == nullptr && kv_div > 0), "spmm_decode_attn: anc_ld=%d kv_div=%d", anc_ld, kv_div);
  SPMM_CHECK_SHAPE(head_stride >= 64 && head_stride % 8 == 0, "spmm_decode_attn: head_stride=%ld (a multiple of 8, >= 64)", head_stride);
  SPMM_CHECK_SHAPE(seq_stride % 8 == 0 && tok_stride % 8 == 0 && ldq >= (long)nH * 64 && ldo >= (long)nH * 64 && ldq % 8 == 0 && ldo % 8 == 0,
                   "spmm_decode_attn: strides must keep 16-B alignment (seq %ld tok %ld)", seq_stride, tok_stride);
  SPMM_CHECK_SHAPE(group > 0 && R % group == 0, "spmm_decode_attn: R=%d must be a multiple of group=%d", R, group);
  const long waves = (long)R * nH;
  const int nblocks = (int)((waves + 3) / 4);
  DecAttnP p = {(const bf16*)q, ldq, (const bf16*)K, (const bf16*)V, seq_stride, tok_stride, head_stride, anc, anc_ld, kv_div, group, nblocks,
                (bf16*)out, ldo, R, nH, Lkv, scale, t_ptr, rowmap, (const bf16*)knew, (const bf16*)vnew, ldn};
  // beams of a molecule on one wave whenever the K/V rows of a group are (mostly) shared: cross-attention (kv_div == group) and
  // self-attention through an ancestry table
  if (knew) hipLaunchKernelGGL(cache_write_kernel, dim3((unsigned)(((long)R * nH * 8 + 255) / 256)), dim3(256), 0, stream, p, Lkv - 1);
  static const bool per_beam = getenv("SPMM_DECODE_PER_BEAM") != nullptr;       // (debugging aid: the one-wave-per-row kernel)
  const bool grouped = !per_beam && group >= 2 && group <= 6 && (anc != nullptr || kv_div == group);
  if (grouped) {
    switch (group) {
      case 2: launch_group<2>(p, stream); break;
      case 3: launch_group<3>(p, stream); break;
      case 4: launch_group<4>(p, stream); break;
      case 5: launch_group<5>(p, stream); break;
      default: launch_group<6>(p, stream); break;
    }
  } else {
    hipLaunchKernelGGL(decode_attn_kernel, dim3((unsigned)((nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
  }
  SPMM_LAUNCH_CHECK("spmm_decode_attn");
  return SPMM_OK;
}

// ---- one position of the k-beam search for N molecules at once (d_pv2smiles_batched.py:36-50 + the top-k branch of
// `generate`, d_pv2smiles_single.py:41-44), as ONE launch instead of ~95 tensor-library launches: next-token softmax and the k
// best successors of every beam, the k*k candidates, candidates ending in [SEP] moved to the molecule's finals in row-major order
// and struck out with -1e5, stop once a molecule holds >= k finals, the k best survivors, their token histories and the K/V
// ancestry table.  One wave per molecule; lane c (< k*k) owns candidate c; everything wave-uniform is made so by butterfly
// reductions (every lane ends with the result), so no LDS and no barrier.
namespace {

struct BeamP {
  const float* logits; long ldl; int N, k, V, Lmax, F;
  int t; const int* t_ptr; int t_off;          // tokens held by every live beam: *t_ptr + t_off when t_ptr is given (graph replay), else t
  int* tokens; float* cur_p; float* fin_p; int* fin_len; int* fin_tok; int* fin_n; unsigned char* done;
  int* anc; int anc_ld; int* ids_out; int* parent_out; int* n_done;
  const int* mol;              // optional: state index (tokens, scores, finals, flags) of compact molecule i -- the live subset after a compaction
  const int* rowmap;           // optional: K/V cache row of compact beam row i*k + b (the "own row" written into the ancestry table)
};
constexpr int BEAM_KMAX = 8, BEAM_SEP = 3;

__device__ __forceinline__ void wave_argmax(float& v, int& i) {        // ties -> the lower index; result in every lane
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

template <int VJ>
__global__ __launch_bounds__(256) void beam_step_kernel(BeamP p) {
  const int lane = threadIdx.x & 63;
  const int ci = blockIdx.x * 4 + (threadIdx.x >> 6);                   // compact molecule index: rows ci*k .. of logits / ids / anc
  if (ci >= p.N) return;                                                // wave-uniform
  const int n = p.mol ? p.mol[ci] : ci;                                 // index of the molecule's state
  const int k = p.k, kk2 = k * k, L = p.Lmax, F1 = p.F + 1;
  const int t = p.t_ptr ? *p.t_ptr + p.t_off : p.t;
  const bool was_done = p.done[n] != 0;
  const float NEG = -__builtin_inff();
  // ---- candidates: log softmax of the k most probable next tokens of every beam, on top of the beam's score
  float my_lp = NEG;
  int my_tok = 0;
  for (int b = 0; b < k; ++b) {
    const float* row = p.logits + (long)(ci * k + b) * p.ldl;
    float v[VJ];
#pragma unroll
    for (int j = 0; j < VJ; ++j) v[j] = lane + 64 * j < p.V ? row[lane + 64 * j] : NEG;
    float m = v[0];
#pragma unroll
    for (int j = 1; j < VJ; ++j) m = fmaxf(m, v[j]);
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VJ; ++j) s += lane + 64 * j < p.V ? expf(v[j] - m) : 0.f;
    s = wave_sum(s);
    const float cp = p.cur_p[n * k + b];
    for (int r = 0; r < k; ++r) {
      float bv = NEG;
      int bi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < VJ; ++j)
        if (v[j] > bv) { bv = v[j]; bi = lane + 64 * j; }
      wave_argmax(bv, bi);
#pragma unroll
      for (int j = 0; j < VJ; ++j)
        if (lane + 64 * j == bi) v[j] = NEG;
      if (lane == b * k + r) { my_lp = cp + logf(expf(bv - m) / s); my_tok = bi; }      // log(softmax(x)[i]) as the reference forms it
    }
  }
  // ---- finals, in row-major candidate order
  int fin_n = p.fin_n[n];
  if (!was_done) {
    for (int c = 0; c < kk2; ++c) {
      if (__shfl(my_tok, c, 64) != BEAM_SEP) continue;                  // wave-uniform
      const float lp = __shfl(my_lp, c, 64);
      const int slot = fin_n < p.F ? fin_n : p.F;                       // slot F: write-only dump (never reached with F = 2k)
      const int* src = p.tokens + ((long)n * k + c / k) * L;
      int* dst = p.fin_tok + ((long)n * F1 + slot) * L;
      for (int pos = lane; pos < L; pos += 64) dst[pos] = pos == t ? BEAM_SEP : src[pos];
      if (lane == 0) { p.fin_p[(long)n * F1 + slot] = lp; p.fin_len[(long)n * F1 + slot] = t + 1; }
      ++fin_n;
      if (lane == c) my_lp = -1e5f;
    }
  }
  // ---- the k best of the k*k candidates
  float cv = lane < kk2 ? my_lp : NEG;
  float new_p[BEAM_KMAX];
  int flat[BEAM_KMAX];
#pragma unroll
  for (int r = 0; r < BEAM_KMAX; ++r) {
    new_p[r] = 0.f; flat[r] = 0;
    if (r < k) {
      float bv = cv;
      int bi = lane;
      wave_argmax(bv, bi);
      new_p[r] = bv; flat[r] = bi;
      if (lane == bi) cv = NEG;
    }
  }
  const bool now_done = was_done || fin_n >= k;                         // a molecule that just reached k finals breaks before this update
  if (!now_done) {
    int tk[BEAM_KMAX][4], an[BEAM_KMAX][4];
#pragma unroll
    for (int r = 0; r < BEAM_KMAX; ++r)
      if (r < k) {
        const long prow = (long)n * k + flat[r] / k, crow = (long)ci * k + flat[r] / k;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pos = lane + 64 * q;
          tk[r][q] = pos < L ? p.tokens[prow * L + pos] : 0;
          an[r][q] = (p.anc && pos < L) ? p.anc[crow * p.anc_ld + pos] : 0;
        }
      }
#pragma unroll
    for (int r = 0; r < BEAM_KMAX; ++r)
      if (r < k) {
        const long row = (long)n * k + r, crow = (long)ci * k + r;
        const int own = p.rowmap ? p.rowmap[crow] : (int)crow;
        const int tokv = __shfl(my_tok, flat[r], 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pos = lane + 64 * q;
          if (pos < L) {
            p.tokens[row * L + pos] = pos == t ? tokv : tk[r][q];
            if (p.anc) p.anc[crow * p.anc_ld + pos] = pos < t ? an[r][q] : own;         // positions < t inherited, the rest the row's own
          }
        }
        if (lane == 0) {
          p.cur_p[row] = new_p[r];
          p.ids_out[crow] = tokv;
          if (p.parent_out) p.parent_out[crow] = flat[r] / k;
        }
      }
  } else if (lane < k) {
    p.ids_out[(long)ci * k + lane] = 0;                                 // a finished molecule keeps decoding [PAD]s nobody reads
    if (p.parent_out) p.parent_out[(long)ci * k + lane] = lane;
  }
  if (lane == 0) {
    p.fin_n[n] = fin_n;
    if (now_done && !was_done) {
      p.done[n] = 1;
      if (p.n_done) atomicAdd(p.n_done, 1);
    }
  }
}

}  // namespace

extern "C" int spmm_beam_step(const float* logits, long ldl, int N, int k, int V, int Lmax, int F, int t, const int* t_ptr, int t_off,
                              int* tokens, float* cur_p, float* fin_p, int* fin_len, int* fin_tok, int* fin_n, unsigned char* done,
                              int* anc, int anc_ld, int* ids_out, int* parent_out, int* n_done, const int* mol, const int* rowmap,
                              hipStream_t stream) {
  SPMM_CHECK_SHAPE(N > 0 && k >= 1 && k <= BEAM_KMAX && V >= k && V <= 512 && Lmax >= 3 && Lmax <= 256 && F >= k && ldl >= V,
                   "spmm_beam_step: N=%d k=%d V=%d Lmax=%d F=%d (k <= 8, k <= V <= 512, Lmax <= 256, F >= k)", N, k, V, Lmax, F);
  SPMM_CHECK_SHAPE(anc == nullptr || anc_ld >= Lmax, "spmm_beam_step: anc_ld=%d < Lmax=%d", anc_ld, Lmax);
  SPMM_CHECK_SHAPE(t_ptr != nullptr || (t >= 1 && t < Lmax), "spmm_beam_step: t=%d outside [1, Lmax)", t);
  BeamP p = {logits, ldl, N, k, V, Lmax, F, t, t_ptr, t_off, tokens, cur_p, fin_p, fin_len, fin_tok, fin_n, done, anc, anc_ld, ids_out, parent_out, n_done, mol, rowmap};
  const dim3 grid((unsigned)((N + 3) / 4));
  if (V <= 320) hipLaunchKernelGGL(beam_step_kernel<5>, grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(beam_step_kernel<8>, grid, dim3(256), 0, stream, p);
  SPMM_LAUNCH_CHECK("spmm_beam_step");
  return SPMM_OK;
}
