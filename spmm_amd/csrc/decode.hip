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
  hipLaunchKernelGGL(decode_attn_kernel, dim3((unsigned)((nblocks + 7) / 8 * 8)), dim3(256), 0, stream, p);
  SPMM_LAUNCH_CHECK("spmm_decode_attn");
  return SPMM_OK;
}
