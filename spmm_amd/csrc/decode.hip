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

// ---- One wave per (molecule, head) serving all G beams of the molecule (round 5: MFMA form).  In cross-attention the beams read the same
// keys / values (kv_div = G); in self-attention their ancestries coincide except for the last positions (beam search coalesces: 6-12
// positions in the configs[3] bench, tools/README.md), so a shared key row is loaded ONCE for the molecule.
// Rounds 3-4 did this on the VALU (per-beam dot products over 8 lanes x 8 elements, two passes over the keys, scores parked in LDS): those
// kernels spent their time waiting, not moving bytes -- a wave walked its keys eight at a time with one or two loads in flight, a full
// memory latency per step, ~145 VALU instructions per step for five beams; the marginal cost per key was ~7 TB/s but ~35 us of every launch
// was fixed (three batches of waves x {ancestry, first keys, first values, reductions}).  Here
//  * the beams are the N dimension of an MFMA: S^T[16 keys x 16 beams] = K-rows[16 x 64] . Q^T (two v_mfma_f32_16x16x32_bf16) and
//    O^T[64 d x 16 beams] += V^T . P^T (four v_mfma_f32_16x16x16_bf16: the probabilities are already in the B operand's registers -- the C
//    layout of the first product IS the B layout of the second -- and V^T comes from the row-major LDS image through ds_read_b64_tr_b16);
//  * the softmax is online (running maximum and sum per beam, rescaled accumulators), so any number of key blocks streams through a ring
//    of D slots: a block is 16 keys = 2 KiB of key rows + 2 KiB of value rows, both by LDS-DMA, every slot requested before the first
//    block is consumed.  (A first version loaded the key rows straight into the A operand's registers with counted waits: the compiler
//    merged the loop-carried registers through copies placed before the waits -- copies of registers whose loads had not landed; 1-100 %
//    of the launches wrong depending on the shape.  A DMA has no register destination: nothing to copy or reuse early.)  Key rows are
//    stored with their 16-B chunks XOR-ed by (key & 7), applied on the DMA's source address, so that the A-operand reads of 16 rows spread
//    over the banks;
//  * ancestries that differ are VIRTUAL KEYS: positions below the first one where some beam sits on another cache row than beam 0
//    (`s`) are shared keys; every later position j contributes G keys (b, j), b's own row, each visible to beam b alone (masked to
//    -inf for the others).  One uniform loop, no divergent branch; unrelated ancestries everywhere (G x Lkv keys) are just more blocks.
//    (One key per DISTINCT row of a position -- 1.1-1.7 of them in the bench's last 6-12 positions, not 5 -- was built and measured: the
//    G^2 compares, the scan and the list it needs in front of the first load cost more than the 1-2 blocks it saves: 2.64 -> 2.74 ms
//    per position.)
// One wave per workgroup: the LDS a wave needs (D x 4 KiB + the ancestry table) then packs the CU without rounding to four waves
// (D = 2: 15 waves per CU at 103 positions; D = 3 and 4 measured 5-15 % slower at 25-50 keys, equal at 100).
// Arithmetic: fp32 scores from bf16 products, p = exp(s - running max) rounded to bf16 in front of V (the tiled training kernel's
// convention), the row sum over the unrounded p.
// 5 000 rows x 12 heads, last six positions on own rows (tools/bench_decode_attn.py): 49.8 -> 29.4 us at 25 keys, 63.4 -> 39.6 at 50,
// 107 -> 76 at 100; all beams on the same rows: 47.5 -> 19.6, 58.9 -> 28.9, 92.9 -> 57.3 (5.4 TB/s).
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int DM_SLOT = 4096;           // one block: 16 keys x 128 B of key rows, then 16 x 128 B of value rows

constexpr int DM_DEPTH = 2;
template <int G, bool ALLSAME>
__global__ __launch_bounds__(64) void decode_attn_mfma_kernel(DecAttnP p, int LD) {
  constexpr int D = DM_DEPTH;
  extern __shared__ __attribute__((aligned(16))) char dm_smem[];
  const int lane = threadIdx.x;
  const int per_xcd = (p.nblocks + 7) >> 3;          // (nblocks = (molecule, head) pairs: consecutive ones on one XCD)
  const long gw = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int nmol = p.R / G;
  if (gw >= (long)nmol * p.nH) return;
  const int n = (int)(gw / p.nH), h = (int)(gw - (long)n * p.nH);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int Lkv = p.t_ptr ? min(*p.t_ptr + 1, p.Lkv) : p.Lkv;
  const uint32_t ring0 = (uint32_t)(uintptr_t)(LDS_AS char*)dm_smem;
  int* sanc = (int*)(dm_smem + D * DM_SLOT);         // cache row of position j for each beam (not ALLSAME)

  // ---- prologue: the queries (B operand: beam r16, elements kk*32 + 8*q4 ..) and the ancestry table, every load issued before any is used
  bf16x8 qf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[kk][e] = (bf16)0.f;
  }
  if (r16 < G) {
    const bf16* qp = p.q + (long)(n * G + r16) * p.ldq + h * 64 + q4 * 8;
    qf[0] = *(const bf16x8*)qp;
    qf[1] = *(const bf16x8*)(qp + 32);
  }
  int s = Lkv;                                       // first position where some beam's cache row differs from beam 0's
  if constexpr (!ALLSAME) {
    int av[4][G];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (64 * u < Lkv) {                            // (wave-uniform; entries past the end re-read the last one: no load behind a lane mask)
        const int j = min(lane + 64 * u, Lkv - 1);
#pragma unroll
        for (int b = 0; b < G; ++b) av[u][b] = p.anc[(long)(n * G + b) * p.anc_ld + j];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (64 * u < Lkv) {
        const int j = lane + 64 * u;
        bool diff = false;
#pragma unroll
        for (int b = 1; b < G; ++b) diff |= av[u][b] != av[u][0];
        if (j < Lkv) {
#pragma unroll
          for (int b = 0; b < G; ++b) sanc[b * LD + j] = av[u][b];
        }
        const unsigned long long m = __ballot(diff && j < Lkv);
        if (m != 0ull) s = min(s, 64 * u + (int)__builtin_ctzll(m));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  const bool has_new = !ALLSAME && p.knew != nullptr;
  if (has_new) {
    // The newest position's key / value rows are still in the projection output: this wave reads them THERE (position Lkv - 1 is every
    // beam's own: G virtual keys) and copies its G x 2 rows of this head into the cache for the positions to come -- the separate copy
    // launch in front of every attention launch (12 per position, ~5 us each) is gone.
    s = min(s, Lkv - 1);
    for (int idx = lane; idx < G * 16; idx += 64) {
      const int b = idx >> 4, isv = (idx >> 3) & 1, ch = idx & 7, r = n * G + b;
      const bf16* src = (isv ? p.vnew : p.knew) + (long)r * p.ldn + h * 64 + ch * 8;
      bf16* dst = (bf16*)(isv ? p.V : p.K) + (long)(p.rowmap ? p.rowmap[r] : r) * p.seq_stride + (long)(Lkv - 1) * p.tok_stride + (long)h * p.head_stride + ch * 8;
      *(bf16x8*)dst = *(const bf16x8*)src;
    }
  }
  // the compiler's own loads are complete before the first DMA is issued: its wait-count bookkeeping does not see the DMA below, and a
  // load it still believed pending would cost a full drain at its first use INSIDE the loop, on every trip
  asm volatile("" : "+v"(qf[0]), "+v"(qf[1])::"memory");
  const int nv = s + G * (Lkv - s);                 // virtual keys: s shared ones, then G per position
  const int nblk = (nv + 15) >> 4;
  const int row_same = (n * G) / max(p.kv_div, 1);
  const long hbase = (long)h * p.head_stride;
  const int tok = (int)p.tok_stride;                // (Lkv <= 256 positions: j * tok_stride fits 32 bits for any cache this launcher accepts)
  // element offset of virtual key v's row of this head (keys past the end re-read the last one: masked where they are used, and a
  // value row that is read must hold finite numbers)
  auto vk_ptr = [&](int v, bool shared, const bf16*& kp, const bf16*& vp) {
    long off;
    if (ALLSAME) {
      off = (long)row_same * p.seq_stride + (long)(min(v, nv - 1) * tok) + hbase;
    } else if (shared) {                             // a block of keys every beam reads from beam 0's rows
      off = (long)sanc[v] * p.seq_stride + (long)(v * tok) + hbase;
    } else {
      v = v < nv ? v : nv - 1;
      int j = v, b = 0;
      if (v >= s) {
        const int w = v - s, jj = w / G;
        b = w - jj * G;
        j = s + jj;
      }
      if (has_new && j == Lkv - 1) {                 // not in the cache yet (this wave's own copy above may not have landed)
        const long o2 = (long)(n * G + b) * p.ldn + h * 64;
        kp = p.knew + o2;
        vp = p.vnew + o2;
        return;
      }
      off = (long)sanc[b * LD + j] * p.seq_stride + (long)(j * tok) + hbase;
    }
    kp = p.K + off;
    vp = p.V + off;
  };

  // block I into ring slot SL: four LDS-DMA operations (the waits below count them).  DMA u: lane -> key u*8 + (lane >> 3), LDS chunk lane & 7
#define DM_DMA(SRC, DST)                                                                                                       \
  do {                                                                                                                         \
    const uint32_t dst_ = __builtin_amdgcn_readfirstlane(DST);                                                                 \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(SRC), "s"(dst_) : "memory", "m0");     \
  } while (0)
#define DM_ISSUE(I, SL)                                                                                                        \
  do {                                                                                                                         \
    const bool sh_ = (I) * 16 + 16 <= s;                                                                                       \
    _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) {                                                                         \
      const bf16 *kp_, *vp_;                                                                                                   \
      vk_ptr((I) * 16 + u_ * 8 + (lane >> 3), sh_, kp_, vp_);                                                                  \
      kp_ += ((lane & 7) ^ ((lane >> 3) & 7)) * 8;                                                                             \
      vp_ += (lane & 7) * 8;                                                                                                   \
      DM_DMA(kp_, ring0 + (uint32_t)((SL) * DM_SLOT + u_ * 1024));                                                             \
      DM_DMA(vp_, ring0 + (uint32_t)((SL) * DM_SLOT + 2048 + u_ * 1024));                                                      \
    }                                                                                                                          \
  } while (0)
#define DM_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  static_assert(D >= 2 && D <= 4, "the landed-wait counts up to three newer blocks");
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nblk) DM_ISSUE(d, d);

  float mrun = -INFINITY, lrun = 0.f;               // running maximum / (this lane's part of the) sum of beam r16
  f32x4 acc[4];                                     // O^T: d = db*16 + 4*q4 + i, beam r16
#pragma unroll
  for (int db = 0; db < 4; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t ka0 = ring0 + (uint32_t)(r16 * 128 + ((q4 ^ (r16 & 7)) << 4));           // A operand: key r16, chunk q4 (kk = 0) ...
  const uint32_t ka1 = ring0 + (uint32_t)(r16 * 128 + (((4 + q4) ^ (r16 & 7)) << 4));     // ... and 4 + q4 (kk = 1)
  const uint32_t va0 = ring0 + (uint32_t)(2048 + (4 * q4 + (r16 >> 2)) * 128 + (r16 & 3) * 8);   // ds_read_b64_tr_b16: row 4*q4 + (i>>2), 4 columns at 4*(i&3)

  for (int base = 0; base < nblk; base += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int i = base + d;
      if (i < nblk) {                               // (wave-uniform)
        const int newer = min(D - 1, nblk - 1 - i);
        if (newer >= 3) DM_WAIT(12);
        else if (newer == 2) DM_WAIT(8);
        else if (newer == 1) DM_WAIT(4);
        else DM_WAIT(0);
        bf16x8 kf0, kf1;
        bf16x4 vf0, vf1, vf2, vf3;
        asm volatile(
            "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\t"
            "ds_read_b64_tr_b16 %2, %8\n\tds_read_b64_tr_b16 %3, %8 offset:32\n\tds_read_b64_tr_b16 %4, %8 offset:64\n\t"
            "ds_read_b64_tr_b16 %5, %8 offset:96\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(kf0), "=&v"(kf1), "=&v"(vf0), "=&v"(vf1), "=&v"(vf2), "=&v"(vf3)
            : "v"(ka0 + (uint32_t)(d * DM_SLOT)), "v"(ka1 + (uint32_t)(d * DM_SLOT)), "v"(va0 + (uint32_t)(d * DM_SLOT))
            : "memory");
        f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f};
        S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[0], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[1], S, 0, 0, 0);
        if (i + D < nblk) DM_ISSUE(i + D, d);        // (the slot's keys and values are in registers)
        float sv[4];
        if (i * 16 + 16 <= s) {                      // (wave-uniform) every key of the block is a shared one
#pragma unroll
          for (int e = 0; e < 4; ++e) sv[e] = S[e] * p.scale;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int v = i * 16 + 4 * q4 + e;
            bool ok = v < nv;
            if (!ALLSAME) ok = ok && (v < s || (v - s) % G == r16);
            sv[e] = ok ? S[e] * p.scale : -INFINITY;
          }
        }
        float bm = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(mrun, bm);
        const float mu = mn == -INFINITY ? 0.f : mn;
        const float alpha = __expf(mrun - mu);
        mrun = mn;
        float ev[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ev[e] = __expf(sv[e] - mu);
        lrun = lrun * alpha + ((ev[0] + ev[1]) + (ev[2] + ev[3]));
        bf16x4 pf;
#pragma unroll
        for (int e = 0; e < 4; ++e) pf[e] = (bf16)ev[e];
#pragma unroll
        for (int db = 0; db < 4; ++db) acc[db] *= alpha;
        const s16x4 pb = __builtin_bit_cast(s16x4, pf);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vf0), pb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vf1), pb, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vf2), pb, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, vf3), pb, acc[3], 0, 0, 0);
      }
    }
  }
#undef DM_ISSUE
#undef DM_DMA
#undef DM_WAIT
  lrun += __shfl_xor(lrun, 16, 64);
  lrun += __shfl_xor(lrun, 32, 64);
  if (r16 < G) {
    const float inv = 1.f / lrun;
    bf16* op = p.out + (long)(n * G + r16) * p.ldo + h * 64 + 4 * q4;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[db][e] * inv);
      *(bf16x4*)(op + db * 16) = o;
    }
  }
}

// the newest position's key / value rows into the cache, in front of the per-row attention kernel (the grouped kernel copies its own rows)
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
  p.nblocks = (int)waves;                                                       // one wave per workgroup
  const int LD = p.anc ? (p.Lkv + 3) & ~3 : 0;
  const unsigned lds = (unsigned)DM_DEPTH * DM_SLOT + G * (unsigned)LD * 4u;    // ring + ancestry rows: <= 16 KiB
  const dim3 grid((unsigned)((waves + 7) / 8 * 8));
  if (p.anc) hipLaunchKernelGGL((decode_attn_mfma_kernel<G, false>), grid, dim3(64), lds, stream, p, LD);
  else hipLaunchKernelGGL((decode_attn_mfma_kernel<G, true>), grid, dim3(64), lds, stream, p, LD);
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
  static const bool per_beam = getenv("SPMM_DECODE_PER_BEAM") != nullptr;       // (debugging aid: the one-wave-per-row kernel)
  const bool grouped = !per_beam && group >= 1 && group <= 8 && (anc != nullptr || kv_div == group);
  if (grouped) {
    switch (group) {
      case 1: launch_group<1>(p, stream); break;
      case 2: launch_group<2>(p, stream); break;
      case 3: launch_group<3>(p, stream); break;
      case 4: launch_group<4>(p, stream); break;
      case 5: launch_group<5>(p, stream); break;
      case 6: launch_group<6>(p, stream); break;
      case 7: launch_group<7>(p, stream); break;
      default: launch_group<8>(p, stream); break;
    }
  } else {
    if (knew) hipLaunchKernelGGL(cache_write_kernel, dim3((unsigned)(((long)R * nH * 8 + 255) / 256)), dim3(256), 0, stream, p, Lkv - 1);
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
  } else {
    if (!was_done && p.anc) {
      // A molecule that has just finished stays in the batch until the next compaction (or to the end, under a replayed graph) and keeps
      // decoding [PAD]s nobody reads.  Left alone its ancestry rows would say "own row" for every later position -- k unrelated rows per
      // position, the attention kernel's expensive case, growing with every step.  All its beams take beam 0's ancestry instead: one
      // shared row per position.
      const long crow0 = (long)ci * k;
      const int own0 = p.rowmap ? p.rowmap[crow0] : (int)crow0;
      int a0[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pos = lane + 64 * q;
        a0[q] = pos < L ? (pos < t ? p.anc[crow0 * p.anc_ld + pos] : own0) : 0;
      }
      for (int r = 0; r < k; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pos = lane + 64 * q;
          if (pos < L) p.anc[(crow0 + r) * p.anc_ld + pos] = a0[q];
        }
    }
    if (lane < k) {
      p.ids_out[(long)ci * k + lane] = 0;
      if (p.parent_out) p.parent_out[(long)ci * k + lane] = lane;
    }
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
