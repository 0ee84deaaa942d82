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
  const bf16* K; const bf16* V; long seq_stride, tok_stride;
  const int* anc; int anc_ld; int kv_div; int group; int nblocks;
  bf16* out; long ldo;
  int R, nH, Lkv; float scale;
  const int* t_ptr;             // optional device step index: the cache holds positions 0..*t_ptr, i.e. Lkv = *t_ptr + 1 (graph replay)
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
  const long own = (long)(r / max(p.kv_div, 1)) * p.seq_stride + h * 64 + c * 8;
  const int Lkv = p.t_ptr ? min(*p.t_ptr + 1, p.Lkv) : p.Lkv;
  const int niter = (Lkv + 7) >> 3;
  float mx = -INFINITY;
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    float part = 0.f;
    if (j < Lkv) {
      const long off = (anc_row ? (long)anc_row[j] * p.seq_stride + h * 64 + c * 8 : own) + (long)j * p.tok_stride;
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
      const long off = (anc_row ? (long)anc_row[j] * p.seq_stride + h * 64 + c * 8 : own) + (long)j * p.tok_stride;
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
template <int G, bool ALLSAME>   // ALLSAME: no ancestry table, every beam of the molecule reads the same key/value rows (cross-attention)
__global__ __launch_bounds__(256) void decode_attn_group_kernel(DecAttnP p) {
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
#pragma unroll
  for (int b = 0; b < G; ++b) {
    const int r = n * G + b;
    const bf16x8 qv = *(const bf16x8*)(p.q + (long)r * p.ldq + h * 64 + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[b][e] = (float)qv[e] * p.scale;
    if (!ALLSAME || b == 0)
      for (int j = lane; j < niter * 8; j += 64)
        sanc[wave][b][j] = p.anc ? (j < Lkv ? p.anc[(long)r * p.anc_ld + j] : 0) : r / max(p.kv_div, 1);
  }
  __builtin_amdgcn_wave_barrier();
  const long hoff = h * 64 + c * 8;
  float mx[G];
#pragma unroll
  for (int b = 0; b < G; ++b) mx[b] = -INFINITY;
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    const bool valid = j < Lkv;
    const long joff = (long)j * p.tok_stride + hoff;
    const int a0 = sanc[wave][0][j];
    bf16x8 k0;
#pragma unroll
    for (int e = 0; e < 8; ++e) k0[e] = (bf16)0.f;
    if (valid) k0 = *(const bf16x8*)(p.K + (long)a0 * p.seq_stride + joff);
#pragma unroll
    for (int b = 0; b < G; ++b) {
      bf16x8 kb = k0;
      if (!ALLSAME && b > 0) {
        const int ab = sanc[wave][b][j];
        if (valid && ab != a0) kb = *(const bf16x8*)(p.K + (long)ab * p.seq_stride + joff);
      }
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) part += (float)kb[e] * qf[b][e];
      part += dpp_f<0xB1>(part); part += dpp_f<0x4E>(part); part += dpp_f<0x141>(part);     // over the key's 8 lanes
      if (valid) {
        if (c == 0) ssc[wave][b][j] = part;
        mx[b] = fmaxf(mx[b], part);
      }
    }
  }
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
#pragma unroll 2
  for (int i = 0; i < niter; ++i) {
    const int j = i * 8 + g;
    if (j < Lkv) {
      const long joff = (long)j * p.tok_stride + hoff;
      const int a0 = sanc[wave][0][j];
      const bf16x8 v0 = *(const bf16x8*)(p.V + (long)a0 * p.seq_stride + joff);
#pragma unroll
      for (int b = 0; b < G; ++b) {
        bf16x8 vb = v0;
        if (!ALLSAME && b > 0) {
          const int ab = sanc[wave][b][j];
          if (ab != a0) vb = *(const bf16x8*)(p.V + (long)ab * p.seq_stride + joff);
        }
        const float e = __expf(ssc[wave][b][j] - mx[b]);
        sum[b] += e;
        const float pe = (float)(bf16)e;             // the tiled training kernel feeds bf16 probabilities to the PV MFMA
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[b][d] += pe * (float)vb[d];
      }
    }
  }
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

template <int G>
void launch_group(DecAttnP p, hipStream_t stream) {
  const long waves = (long)(p.R / G) * p.nH;
  p.nblocks = (int)((waves + 3) / 4);
  if (p.anc) hipLaunchKernelGGL((decode_attn_group_kernel<G, false>), dim3((unsigned)((p.nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((decode_attn_group_kernel<G, true>), dim3((unsigned)((p.nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
}

}  // namespace

extern "C" int spmm_decode_attn(const void* q, long ldq, const void* K, const void* V, long seq_stride, long tok_stride,
                                const int* anc, int anc_ld, int kv_div, int group, void* out, long ldo, int R, int nH, int Lkv,
                                float scale, const int* t_ptr, hipStream_t stream) {
  SPMM_CHECK_SHAPE(R > 0 && nH > 0 && Lkv > 0 && Lkv <= 256, "spmm_decode_attn: R=%d nH=%d Lkv=%d (Lkv <= 256)", R, nH, Lkv);
  SPMM_CHECK_SHAPE((anc != nullptr && anc_ld >= Lkv) || (anc == nullptr && kv_div > 0), "spmm_decode_attn: anc_ld=%d kv_div=%d", anc_ld, kv_div);
  SPMM_CHECK_SHAPE(seq_stride % 8 == 0 && tok_stride % 8 == 0 && ldq >= (long)nH * 64 && ldo >= (long)nH * 64 && ldq % 8 == 0 && ldo % 8 == 0,
                   "spmm_decode_attn: strides must keep 16-B alignment (seq %ld tok %ld)", seq_stride, tok_stride);
  SPMM_CHECK_SHAPE(group > 0 && R % group == 0, "spmm_decode_attn: R=%d must be a multiple of group=%d", R, group);
  const long waves = (long)R * nH;
  const int nblocks = (int)((waves + 3) / 4);
  DecAttnP p = {(const bf16*)q, ldq, (const bf16*)K, (const bf16*)V, seq_stride, tok_stride, anc, anc_ld, kv_div, group, nblocks,
                (bf16*)out, ldo, R, nH, Lkv, scale, t_ptr};
  // beams of a molecule on one wave whenever the K/V rows of a group are (mostly) shared: cross-attention (kv_div == group) and
  // self-attention through an ancestry table
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
