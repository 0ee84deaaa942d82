// Row bookkeeping of one pretraining step as device kernels (spmm_amd/step.py): the packed-row plan of the text passes, the
// index arrays of the fusion batch (S6) and of its CLS-only top layer, row assembly / scatter helpers.  These replace the ~80 small
// tensor-library launches (argsort, cumsum, cat, index_select, index_add_, zeros ...) a step used to spend on them; nothing here is
// arithmetic of the model.  Reference: the gathers they serve are SPMM_models.py:139-150,181-199 (which sequences cross-attend to
// which), :154-178 (hard negatives), :95,105,201 (only position 0 of the ITM / feature passes is read).
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

constexpr long SRC_B = 1l << 40;        // spmm_gather_rows2: index bit that selects the second source

// dst[r] = idx[r] < 0 ? 0 : (idx[r] & SRC_B ? srcB : srcA)[idx[r] & (SRC_B - 1)]     (rows of H bf16, 16-byte vectors)
__global__ void gather_rows2_kernel(bf16* __restrict__ dst, const bf16* __restrict__ srcA, const bf16* __restrict__ srcB,
                                    const long* __restrict__ idx, long rows, int H) {
  const int h8 = H / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * h8; i += (long)gridDim.x * blockDim.x) {
    const long r = i / h8;
    const int c = (int)(i - r * h8) * 8;
    const long j = idx[r];
    bf16x8 v = {};
    if (j >= 0) v = *(const bf16x8*)(((j & SRC_B) ? srcB : srcA) + (j & (SRC_B - 1)) * H + c);
    *(bf16x8*)(dst + r * H + c) = v;
  }
}

// dst[idx[r]] += src[r]  (bf16 rows, every idx at most once: no atomics)
__global__ void add_rows_bf16_kernel(bf16* __restrict__ dst, const long* __restrict__ idx, const bf16* __restrict__ src, long rows, int H) {
  const int h8 = H / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * h8; i += (long)gridDim.x * blockDim.x) {
    const long r = i / h8;
    const int c = (int)(i - r * h8) * 8;
    bf16* d = dst + idx[r] * H + c;
    const bf16x8 a = *(const bf16x8*)d, b = *(const bf16x8*)(src + r * H + c);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)a[e] + (float)b[e]);
    *(bf16x8*)d = o;
  }
}

__global__ void zero_bytes_kernel(uint4* __restrict__ p, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p[i] = uint4{0u, 0u, 0u, 0u};
}

__global__ void zero_rows_kernel(char* __restrict__ p, long rows, long row16, long stride_bytes) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * row16; i += (long)gridDim.x * blockDim.x) {
    const long r = i / row16, c = i - r * row16;
    *(uint4*)(p + r * stride_bytes + c * 16) = uint4{0u, 0u, 0u, 0u};
  }
}

// out = dz * gelu'(pre)  (erf-GELU; the small heads' backward, xbert.py:673 / SPMM_models.py:62-66)
__global__ void gelu_bwd_kernel(const bf16* __restrict__ dz, const bf16* __restrict__ pre, bf16* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const bf16x8 a = *(const bf16x8*)(dz + i * 8), x = *(const bf16x8*)(pre + i * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xf = (float)x[e];
      const float cdf = 0.5f * (1.0f + erff(xf * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * xf * xf);
      o[e] = (bf16)((float)a[e] * (cdf + xf * pdf));
    }
    *(bf16x8*)(out + i * 8) = o;
  }
}

// ---- packed-row plan of the text passes (PretrainStep._pack_plan): ONE workgroup of 1024 threads.
// mask [B, Lt] (0 / non-0); M = the row count the host sized the launches with.  Valid tokens in row-major order get packed rows
// 0, 1, 2, ...; whatever the mask holds, no index written here leaves [0, M) resp. the dense range (a mask that contradicts M raises
// `bad`, the step is then skipped like a non-finite loss).
__global__ __launch_bounds__(1024) void pack_plan_kernel(const int* __restrict__ mask, int B, int Lt, int M, int* __restrict__ lens32,
                                                          int* __restrict__ row0_32, long* __restrict__ row0_64, long* __restrict__ rows,
                                                          long* __restrict__ gidx2, long* __restrict__ gidx4, long* __restrict__ inv,
                                                          long* __restrict__ idx_m, int* __restrict__ bad) {
  extern __shared__ int sh[];             // [B] valid counts, [B] exclusive starts, [2] flags
  int* cnt = sh;
  int* start = sh + B;
  int* flag = sh + 2 * B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  if (tid < 2) flag[tid] = 0;
  __syncthreads();
  // pass 1: per sequence the number of valid tokens and whether they form a non-empty prefix
  for (int s = wave; s < B; s += nw) {
    int n = 0, holes = 0;
    for (int c0 = 0; c0 < Lt; c0 += 64) {
      const int l = c0 + lane;
      const bool v = l < Lt && mask[(long)s * Lt + l] != 0;
      const unsigned long long b = __ballot(v);
      n += __popcll(b);
      // a prefix chunk has its set bits contiguous from bit 0, and a later chunk may only be non-empty if every earlier one was full
      holes |= (b & (b + 1)) != 0ull;
      holes |= (b != 0ull && n - __popcll(b) != c0);
    }
    if (lane == 0) {
      cnt[s] = n;
      if (holes || n < 1) atomicOr(&flag[0], 1);
    }
  }
  __syncthreads();
  for (int s = tid; s < B; s += blockDim.x) {
    int a = 0;
    for (int j = 0; j < s; ++j) a += cnt[j];
    start[s] = a;
  }
  __syncthreads();
  if (tid == 0) {
    const int total = start[B - 1] + cnt[B - 1];
    if (total != M) flag[0] = 1;
    flag[1] = total;
    if (bad && flag[0]) atomicOr(bad, 1);
  }
  __syncthreads();
  const int total = flag[1];
  for (int s = tid; s < B; s += blockDim.x) {       // what the launches index with: clamped to the M rows the host sized
    int r0 = start[s] < M - 1 ? start[s] : M - 1;
    int n = cnt[s] < M - r0 ? cnt[s] : M - r0;
    n = n < 1 ? 1 : n;
    lens32[s] = n; row0_32[s] = r0; row0_64[s] = r0;
    idx_m[s] = r0;                                  // [position 0 of every packed sequence | every row of the second packed copy]: the momentum
  }                                                 // text encoder's last layer keeps these rows of its [packed | packed] batch (step.py)
  for (int i = tid; i < M; i += blockDim.x) idx_m[B + i] = (long)M + i;
  // pass 2: packed row of every valid token, dense row of every packed row
  for (int s = wave; s < B; s += nw) {
    int base = start[s];
    for (int c0 = 0; c0 < Lt; c0 += 64) {
      const int l = c0 + lane;
      const bool v = l < Lt && mask[(long)s * Lt + l] != 0;
      const unsigned long long b = __ballot(v);
      const int p = base + __popcll(b & ((1ull << lane) - 1ull));
      if (l < Lt) {
        const long d = (long)s * Lt + l;
        inv[d] = (v && p < M) ? p : -1;
        if (v && p < M) { rows[p] = d; gidx2[p] = d; gidx4[p] = d; gidx4[M + p] = (long)B * Lt + d; }
      }
      base += __popcll(b);
    }
  }
  for (int p = total + tid; p < M; p += blockDim.x) { rows[p] = 0; gidx2[p] = 0; gidx4[p] = 0; gidx4[M + p] = (long)B * Lt; }   // (bad mask only)
  for (long j = tid; j < (long)B * Lt; j += blockDim.x) { gidx2[M + j] = (long)B * Lt + j; inv[(long)B * Lt + j] = M + j; }
}

// ---- index arrays of the fusion batch (PretrainStep.forward, packed text layout).  Row layout of the batch X6:
//   [ PV queries of the three ITM passes: pe | pe[neg_p] | pe   (3B x Lp)          rows [0, o_tp)
//   | text queries, packed: te | te                          (2M)                 rows [o_tp, o_lm)
//   | the LM pass (hidden10, dense B x Lt)                                          rows [o_lm, o_12)
//   | the causal PV pass (B x Lp)                                                   rows [o_12, o_8)
//   | text negatives as queries, PACKED: sequence s = te[neg_t[s]]                  rows [o_8, o_8 + Mn),  Mn = sum_s len[neg_t[s]] <= B Lt ]
// Mn is known only here, on the device: the batch is allocated for B Lt such rows and every row-wise / GEMM launch over it takes its row
// count R = o_8 + Mn from rows_dev (include/spmm_hip.h, "device-side row counts") -- no host read, no padding rows computed.
// Sources of the assembly gather: A = y1 = [prop_embeds (B Lp) ; prop_embeds_causal (B Lp)], B = y2 = [text_embeds (M) ; hidden10 (B Lt)].
// small32: 35 x B int32, layout in units of B (spmm_amd/ops.py::FUSION_SMALL mirrors it):
//   0 ar | 1 kvidx_pv[3B] = ar,ar,neg_t | 4 kvidx_tp[2B] = ar,neg_p | 6 qrow0_tp[2B] | 8 qlen_tp[2B] | 10 row0_8[B] | 11 len_8[B]
//   12 skv_row0_pv[3B] | 15 skv_len_pv[3B] | 18 skv_row0_tx[2B] | 20 skv_len_tx[2B] | 22 start_t[B+1] | 24 list_t[4B] | 28 start_p[B+1]
//   30 list_p[4B] | 34 rows_dev[2] = {o_8 + Mn, Mn}
__global__ __launch_bounds__(256) void fusion_plan_kernel(const long* __restrict__ neg, const int* __restrict__ lens, const int* __restrict__ row0,
                                                           int B, int Lt, int Lp, int M, long* __restrict__ idx6, long* __restrict__ neg_rows,
                                                           long* __restrict__ idx_top, int* __restrict__ small32) {
  extern __shared__ int r8[];                          // [B + 1] first packed row of every text negative (every workgroup computes it)
  const long BLp = (long)B * Lp, BLt = (long)B * Lt;
  const long o_tp = 3 * BLp, o_lm = o_tp + 2l * M, o_12 = o_lm + BLt, o_8 = o_12 + BLp, Rcap = o_8 + BLt;
  const long* neg_p = neg;
  const long* neg_t = neg + B;
  for (int s = threadIdx.x; s <= B; s += blockDim.x) {
    int a = 0;
    for (int j = 0; j < s; ++j) a += lens[neg_t[j]];
    r8[s] = a;
  }
  __syncthreads();
  const int Mn = r8[B];
  const long gsz = (long)gridDim.x * blockDim.x, g0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long r = g0; r < o_8; r += gsz) {
    long j;
    if (r < o_tp) {
      const long g = r / BLp, w = r - g * BLp, s = w / Lp, l = w - s * Lp;
      j = (g == 1 ? neg_p[s] : s) * Lp + l;
    } else if (r < o_lm) {
      const long w = r - o_tp;
      j = SRC_B | (w >= M ? w - M : w);
    } else if (r < o_12) {
      j = SRC_B | (M + (r - o_lm));
    } else {
      j = BLp + (r - o_12);
    }
    idx6[r] = j;
  }
  for (long w = g0; w < BLt; w += gsz) {               // the packed negatives: token l of sequence s <- token l of te[neg_t[s]]
    const long s = w / Lt, l = w - s * Lt, n = neg_t[s];
    if (l < lens[n]) {
      const long j = r8[s] + l;
      idx6[o_8 + j] = SRC_B | (row0[n] + l);
      neg_rows[j] = row0[n] + l;
    }
    if (w >= Mn) { idx6[o_8 + w] = -1; neg_rows[w] = -1; }
  }
  // top layer: rows of the layer input it keeps -- position 0 of the 6B ITM sequences, then every row of the LM and causal-PV passes
  for (long w = g0; w < 6l * B + BLt + BLp; w += gsz) {
    long j;
    if (w < 3l * B) j = w * Lp;
    else if (w < 4l * B) j = o_tp + row0[w - 3l * B];
    else if (w < 5l * B) j = o_tp + M + row0[w - 4l * B];
    else if (w < 6l * B) j = o_8 + r8[w - 5l * B];
    else j = o_lm + (w - 6l * B);
    idx_top[w] = j;
  }
  (void)Rcap;
  if (blockIdx.x != 0) return;
  int* S = small32;
  if (threadIdx.x == 0) { S[34 * B] = (int)o_8 + Mn; S[34 * B + 1] = Mn; }
  for (int s = threadIdx.x; s < B; s += blockDim.x) {
    const int np = (int)neg_p[s], nt = (int)neg_t[s];
    S[s] = s;
    S[1 * B + s] = s; S[2 * B + s] = s; S[3 * B + s] = nt;
    S[4 * B + s] = s; S[5 * B + s] = np;
    S[6 * B + s] = row0[s]; S[7 * B + s] = row0[s] + M;
    S[8 * B + s] = lens[s]; S[9 * B + s] = lens[s];
    S[10 * B + s] = r8[s]; S[11 * B + s] = lens[nt];
    S[12 * B + s] = s * Lp; S[13 * B + s] = (B + s) * Lp; S[14 * B + s] = (2 * B + s) * Lp;
    S[15 * B + s] = Lp; S[16 * B + s] = Lp; S[17 * B + s] = Lp;
    S[18 * B + s] = (int)o_tp + row0[s]; S[19 * B + s] = (int)o_tp + M + row0[s];
    S[20 * B + s] = lens[s]; S[21 * B + s] = lens[s];
    // inverse (CSR) maps of the shared key/value sources: the consumers of source u in consumer order
    //   text source u: PV sequences u, B+u, {2B+s : neg_t[s] == u}, 3B+u      PV source u: text sequences u, {B+s : neg_p[s] == u}, 2B+u, 3B+u
    int ct = 0, cp = 0, bt = 0, bp = 0;               // consumers drawn as negatives: of u, and of all sources before u
    for (int j = 0; j < B; ++j) {
      const int a = (int)neg_t[j], b = (int)neg_p[j];
      ct += a == s; bt += a < s;
      cp += b == s; bp += b < s;
    }
    const int st = 3 * s + bt, sp = 3 * s + bp;
    S[22 * B + s] = st; S[28 * B + s] = sp;
    if (s == B - 1) { S[22 * B + B] = st + 3 + ct; S[28 * B + B] = sp + 3 + cp; }
    int* Lt_ = S + 24 * B + st;
    int* Lp_ = S + 30 * B + sp;
    Lt_[0] = s; Lt_[1] = B + s;
    Lp_[0] = s;
    int kt = 2, kp = 1;
    for (int j = 0; j < B; ++j) {
      if ((int)neg_t[j] == s) Lt_[kt++] = 2 * B + j;
      if ((int)neg_p[j] == s) Lp_[kp++] = B + j;
    }
    Lt_[kt] = 3 * B + s;
    Lp_[kp] = 2 * B + s; Lp_[kp + 1] = 3 * B + s;
  }
}

int blocks_for(long work, int block) {
  long g = (work + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int spmm_gather_rows2(void* dst, const void* srcA, const void* srcB, const long* idx, long rows, int H, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && H > 0 && H % 8 == 0 && idx && srcA, "spmm_gather_rows2: rows=%ld H=%d", rows, H);
  hipLaunchKernelGGL(gather_rows2_kernel, dim3(blocks_for(rows * (H / 8), 256)), dim3(256), 0, stream, (bf16*)dst, (const bf16*)srcA,
                     (const bf16*)(srcB ? srcB : srcA), idx, rows, H);
  SPMM_LAUNCH_CHECK("spmm_gather_rows2");
  return SPMM_OK;
}

extern "C" int spmm_add_rows_bf16(void* dst, const long* idx, const void* src, long rows, int H, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && H > 0 && H % 8 == 0 && idx, "spmm_add_rows_bf16: rows=%ld H=%d", rows, H);
  hipLaunchKernelGGL(add_rows_bf16_kernel, dim3(blocks_for(rows * (H / 8), 256)), dim3(256), 0, stream, (bf16*)dst, idx, (const bf16*)src, rows, H);
  SPMM_LAUNCH_CHECK("spmm_add_rows_bf16");
  return SPMM_OK;
}

extern "C" int spmm_zero_bytes(void* p, long nbytes, hipStream_t stream) {
  SPMM_CHECK_SHAPE(nbytes > 0 && nbytes % 16 == 0 && ((uintptr_t)p & 15) == 0, "spmm_zero_bytes: %ld bytes at %p must be 16-byte granular", nbytes, p);
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(blocks_for(nbytes / 16, 256)), dim3(256), 0, stream, (uint4*)p, nbytes / 16);
  SPMM_LAUNCH_CHECK("spmm_zero_bytes");
  return SPMM_OK;
}

extern "C" int spmm_zero_rows(void* p, long rows, long row_bytes, long stride_bytes, hipStream_t stream) {
  SPMM_CHECK_SHAPE(rows > 0 && row_bytes > 0 && row_bytes % 16 == 0 && stride_bytes % 16 == 0 && ((uintptr_t)p & 15) == 0,
                   "spmm_zero_rows: %ld rows of %ld bytes at stride %ld must be 16-byte granular", rows, row_bytes, stride_bytes);
  hipLaunchKernelGGL(zero_rows_kernel, dim3(blocks_for(rows * (row_bytes / 16), 256)), dim3(256), 0, stream, (char*)p, rows, row_bytes / 16, stride_bytes);
  SPMM_LAUNCH_CHECK("spmm_zero_rows");
  return SPMM_OK;
}

extern "C" int spmm_gelu_bwd(const void* dz, const void* pre, void* out, long n, hipStream_t stream) {
  SPMM_CHECK_SHAPE(n > 0 && n % 8 == 0, "spmm_gelu_bwd: n=%ld must be a positive multiple of 8", n);
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks_for(n / 8, 256)), dim3(256), 0, stream, (const bf16*)dz, (const bf16*)pre, (bf16*)out, n / 8);
  SPMM_LAUNCH_CHECK("spmm_gelu_bwd");
  return SPMM_OK;
}

extern "C" int spmm_pack_plan(const int* mask, int B, int Lt, int M, int* lens32, int* row0_32, long* row0_64, long* rows, long* gidx2,
                              long* gidx4, long* inv, long* idx_m, int* bad, hipStream_t stream) {
  SPMM_CHECK_SHAPE(B >= 1 && B <= 8191 && Lt >= 1 && M >= 1 && (long)M <= (long)B * Lt, "spmm_pack_plan: B=%d Lt=%d M=%d (B <= 8191: (2B + 2) ints of LDS within 64 KiB)", B, Lt, M);
  hipLaunchKernelGGL(pack_plan_kernel, dim3(1), dim3(1024), (2 * B + 2) * sizeof(int), stream, mask, B, Lt, M, lens32, row0_32, row0_64, rows,
                     gidx2, gidx4, inv, idx_m, bad);
  SPMM_LAUNCH_CHECK("spmm_pack_plan");
  return SPMM_OK;
}

extern "C" int spmm_fusion_plan(const long* neg, const int* lens32, const int* row0_32, int B, int Lt, int Lp, int M,
                                long* idx6, long* neg_rows, long* idx_top, int* small32, hipStream_t stream) {
  SPMM_CHECK_SHAPE(B >= 1 && B <= 8191 && Lt >= 1 && Lp >= 1 && M >= 1 && (long)M <= (long)B * Lt, "spmm_fusion_plan: B=%d Lt=%d Lp=%d M=%d", B, Lt, Lp, M);
  const long Rcap = 4l * B * Lp + 2l * M + 2l * B * Lt;
  SPMM_CHECK_SHAPE(Rcap < (1l << 31), "spmm_fusion_plan: %ld rows do not fit the int32 row tables", Rcap);
  hipLaunchKernelGGL(fusion_plan_kernel, dim3(blocks_for(Rcap, 256) > 64 ? 64 : blocks_for(Rcap, 256)), dim3(256), (B + 1) * sizeof(int), stream, neg, lens32,
                     row0_32, B, Lt, Lp, M, idx6, neg_rows, idx_top, small32);
  SPMM_LAUNCH_CHECK("spmm_fusion_plan");
  return SPMM_OK;
}
