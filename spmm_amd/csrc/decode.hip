// Single-query attention over a key/value cache for autoregressive PV -> SMILES beam decoding.
// Replaces, for the newest position of every beam, xbert.py:305-354 (scores, 1/sqrt(d), softmax, context, head merge) of
// both BertSelfAttention flavours: self-attention reads the per-layer cache of the beam's own ancestry, cross-attention
// reads the PV keys/values computed once per molecule and shared by its k beams.
//
// Shape of the work: R rows (molecules x beams) x nH heads, one query each, Lkv <= 256 keys, d = 64.  It is an HBM-bound
// gather (2 x Lkv x 128 B per row-head), so there is no MFMA here: one wave64 per (row, head); lane j scores key j with a
// 64-term fp32 dot product (16-B loads), wave-shuffle softmax, then lane d accumulates output dim d over the keys
// (128-B coalesced V rows).  Beams are never physically reordered: `anc[r, j]` names the cache row that holds position j of
// row r's hypothesis (updated by the host-side beam bookkeeping with one small gather per step).
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

struct DecAttnP {
  const bf16* q; long ldq;
  const bf16* K; const bf16* V; long seq_stride, tok_stride;
  const int* anc; int anc_ld; int kv_div;
  bf16* out; long ldo;
  int R, nH, Lkv; float scale;
};

__global__ __launch_bounds__(256) void decode_attn_kernel(DecAttnP p) {
  __shared__ float sq[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * 4 + wave;
  if (gw >= (long)p.R * p.nH) return;
  const int r = (int)(gw / p.nH), h = (int)(gw - (long)r * p.nH);
  sq[wave][lane] = (float)p.q[(long)r * p.ldq + h * 64 + lane] ;
  __builtin_amdgcn_wave_barrier();
  // scores: lane owns keys lane, lane+64, lane+128, lane+192
  float s[4]; long long base[4];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = lane + 64 * c;
    s[c] = -INFINITY; base[c] = 0;
    if (c * 64 < p.Lkv && j < p.Lkv) {
      const long seq = p.anc ? (long)p.anc[(long)r * p.anc_ld + j] : (long)(r / p.kv_div);
      base[c] = seq * p.seq_stride + (long)j * p.tok_stride + h * 64;
      const bf16* kp = p.K + base[c];
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < 64; d += 8) {
        const bf16x8 kv = *(const bf16x8*)(kp + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)kv[e] * sq[wave][d + e];
      }
      s[c] = acc * p.scale;
      mx = fmaxf(mx, s[c]);
    }
  }
  mx = wave_max(mx);
  float e[4], sum = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    e[c] = (s[c] == -INFINITY) ? 0.f : __expf(s[c] - mx);
    sum += e[c];
    e[c] = (float)(bf16)e[c];          // the tiled training kernel feeds bf16 probabilities to the PV MFMA; keep the same rounding
  }
  sum = wave_sum(sum);
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = min(64, p.Lkv - 64 * c);
    for (int jj = 0; jj < n; ++jj) {
      const float pj = __shfl(e[c], jj, 64);
      const long long b = __shfl(base[c], jj, 64);
      acc += pj * (float)p.V[b + lane];
    }
  }
  p.out[(long)r * p.ldo + h * 64 + lane] = (bf16)(acc / sum);
}

}  // namespace

extern "C" int spmm_decode_attn(const void* q, long ldq, const void* K, const void* V, long seq_stride, long tok_stride,
                                const int* anc, int anc_ld, int kv_div, void* out, long ldo, int R, int nH, int Lkv, float scale,
                                hipStream_t stream) {
  SPMM_CHECK_SHAPE(R > 0 && nH > 0 && Lkv > 0 && Lkv <= 256, "spmm_decode_attn: R=%d nH=%d Lkv=%d (Lkv <= 256)", R, nH, Lkv);
  SPMM_CHECK_SHAPE((anc != nullptr && anc_ld >= Lkv) || (anc == nullptr && kv_div > 0), "spmm_decode_attn: anc_ld=%d kv_div=%d", anc_ld, kv_div);
  SPMM_CHECK_SHAPE(seq_stride % 8 == 0 && tok_stride % 8 == 0 && ldq >= (long)nH * 64 && ldo >= (long)nH * 64,
                   "spmm_decode_attn: strides must keep 16-B alignment (seq %ld tok %ld)", seq_stride, tok_stride);
  DecAttnP p = {(const bf16*)q, ldq, (const bf16*)K, (const bf16*)V, seq_stride, tok_stride, anc, anc_ld, kv_div, (bf16*)out, ldo, R, nH, Lkv, scale};
  const long waves = (long)R * nH;
  hipLaunchKernelGGL(decode_attn_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p);
  SPMM_LAUNCH_CHECK("spmm_decode_attn");
  return SPMM_OK;
}
