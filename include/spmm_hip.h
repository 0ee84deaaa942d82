/* spmm_hip.h -- C ABI of libspmm_hip.so: the MI355X (gfx950) kernels behind the SPMM pretraining step.
 *
 * The reference (jinhojsk515/spmm) is pure Python on stock torch ops and has no FFI layer of its own; this ABI is
 * the boundary introduced beneath its Python class API.  Each entry point names the
 * reference code it replaces (file:line under /root/reference).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to call these from SPMM_models.py / xbert.py.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (PyTorch allocates; the
 *    library never allocates, frees or retains device memory).
 *  - "bf16" tensors are raw 16-bit bfloat16, passed as void*.  Token-major activations: row = seq * L + pos.
 *  - every call is asynchronous on `stream` (a hipStream_t); no call synchronises or reads device data on the host.
 *  - return value: 0 ok, 1 bad shape/argument, 2 HIP launch failure, 3 unsupported; message via spmm_last_error()
 *    (thread local).  No C++ exception crosses the boundary.
 *  - scalars that change from step to step (alpha, temp, lr, dropout seed, queue pointer, loss-gradient scales)
 *    are read from device memory so that a whole training step can be captured once into a hipGraph and replayed.
 */
#ifndef SPMM_HIP_H
#define SPMM_HIP_H

#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
typedef hipStream_t spmm_stream_t;
#else
typedef void* spmm_stream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

int spmm_version(void);
const char* spmm_last_error(void);

/* epilogues of spmm_gemm_nt */
#define SPMM_EPI_BF16 0        /* C(bf16) = alpha*acc/(*div) + bias + R                                  */
#define SPMM_EPI_GELU 1        /* C2(bf16) = pre = acc + bias ; C(bf16) = gelu_erf(pre)                  */
#define SPMM_EPI_F32 2         /* C(f32)  = alpha*acc/(*div) + bias                                      */
#define SPMM_EPI_F32_ATOMIC 3  /* C(f32) += alpha*acc  with atomicAdd (split-K allowed)                  */
#define SPMM_EPI_GELU_GRAD 4   /* C(bf16) = acc * gelu_erf'(G)                                           */
#define SPMM_EPI_F32_ACC 5     /* C(f32) += alpha*acc + bias   (plain read-modify-write, single owner)   */
#define SPMM_EPI_GELU_DERIV 6  /* pre = acc + bias ; C(bf16) = gelu_erf(pre) ; C2(bf16) = gelu_erf'(pre): the FFN forward keeps  */
                               /* the derivative (one shared exp) so that the backward epilogue is SPMM_EPI_MUL                  */
#define SPMM_EPI_MUL 7         /* C(bf16) = alpha*acc * G      (G bf16, e.g. the stored gelu'; colsum allowed)                   */

/* C[M,N] = A[M,K] . W[N,K]^T on MFMA (bf16 in, fp32 accumulate).  Replaces every nn.Linear on the path
 * (xbert.py:280-300 query/key/value, :370 attention output.dense, :435 intermediate.dense + erf GELU :436,
 * :448 output.dense, :673 transform.dense, :695 tied decoder; SPMM_models.py:31-42 heads) and, with transposed
 * operands, their dgrad/wgrad GEMMs; also the similarity GEMMs SPMM_models.py:108-111,121-124 (split-bf16 K=3E).
 * colsum (optional, bf16 / GELU-grad epilogues): colsum[n] += sum_m C[m][n] -- the bias gradient when C is a dY.
 * kernel (per call, no process state): 0 = chosen from the shape; SPMM_GEMM_K128 = 128x128 tile (every epilogue, split-K);
 * SPMM_GEMM_K128PC = the 128x128 tile with four loader waves beside the four compute waves (LDS-DMA issue and a four-stage ring off the
 * computing waves' instruction stream; holds a CU alone: chosen where the tiling gives at most one workgroup per CU -- the decoder's
 * 5 000-row launches);
 * SPMM_GEMM_K256x128 = 256x128 three-stage ring (no atomic epilogue); SPMM_GEMM_K256 = 256x256 tile, one barrier per k-step;
 * SPMM_GEMM_K256P8 = 256x256 tile on the 8-phase schedule (bf16-output epilogues, K % 128 == 0) -- the default for the
 * training step's large GEMMs: one persistent workgroup per CU walking its XCD's tile range.  SPMM_GEMM_K256P8_TILES = the same
 * kernel with one workgroup per tile, SPMM_GEMM_AUTO_TILES = the automatic choice with that variant wherever the 8-phase kernel
 * is chosen: for launches that share the GPU with a long-running kernel on another stream (a collective): the hardware dispatcher
 * then places tiles on whatever CUs are free instead of 1/256th of the work waiting for an occupied CU (tools/gemm_bench contend). */
#define SPMM_GEMM_AUTO 0
#define SPMM_GEMM_K128 1
#define SPMM_GEMM_K256x128 2
#define SPMM_GEMM_K256 3
#define SPMM_GEMM_K128PC 5
#define SPMM_GEMM_K256P8 8
#define SPMM_GEMM_K256P8_TILES 9
#define SPMM_GEMM_AUTO_TILES 16
int spmm_gemm_nt(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int splits, const float* bias,
                 const float* div_ptr, float alpha, const void* R, long ldr, const void* G, long ldg, void* C, long ldc,
                 void* C2, long ldc2, int epi, float* colsum, int kernel, const int* M_dev, spmm_stream_t stream);
/* Device-side row counts (`M_dev` / `R_dev` / `rows_dev`, optional, null = none): spmm_gemm_nt (8-phase kernel),
 * spmm_gemm_tn (8-phase kernel, split launch), spmm_colsum_bf16, spmm_ln_fwd and spmm_ln_bwd take a pointer to an int in device memory
 * holding the number of rows to process, <= the host-side row count the launch and the buffers are sized for.  Rows past it are neither
 * read nor written.  For batches whose tail length only the device knows -- here the text hard negatives drawn on the device
 * (SPMM_models.py:166-178) that re-enter the fusion layers as packed query rows: no host read sizes the step. */
/* Weight-gradient GEMM C[N,K] += alpha * A[M,N]^T . B[M,K] straight from the token-major activations (LDS transpose reads,
 * no transposed copies); `splits` > 1 reduces partial slabs from `workspace` (spmm_gemm_tn_workspace_bytes) without atomics.
 * Replaces autograd's weight-gradient matmuls of every nn.Linear on the path.  spmm_colsum_bf16: bias gradients. */
long spmm_gemm_tn_workspace_bytes(int M, int N, int K, int splits);
/* kernel (per call): 0 = chosen from the shape; 1 = 128x128 tiles; 8 = 256x256 tiles on the 8-phase schedule (N, K % 8 == 0).
 * spmm_gemm_tn_splits returns the split count that fills the chip for that choice (pass the same `kernel` to both). */
int spmm_gemm_tn_splits(int M, int N, int K, int kernel);
int spmm_gemm_tn(const void* A, long lda, const void* B, long ldb, int M, int N, int K, int splits, float alpha, float* C,
                 long ldc, float* workspace, int kernel, const int* M_dev, spmm_stream_t stream);
/* C[n*ldc + k] += sum over ns slabs of N*K floats in `ws`: the split reduction spmm_gemm_tn runs itself, as an entry of its own. */
int spmm_gemm_tn_reduce(const float* ws, int ns, int N, int K, float* C, long ldc, spmm_stream_t stream);
int spmm_colsum_bf16(const void* x, long ld, int R, int C, float* out, const int* R_dev, spmm_stream_t stream);

/* Attention core softmax(QK^T/8 + mask) -> dropout -> .V for head_dim 64, Lq,Lkv <= 256 (one workgroup holds the K/V
 * panel of a head in LDS; the forward runs one workgroup per 128-query chunk, the backward one launch per 128-query chunk).
 * Replaces BertSelfAttention.forward xbert.py:305-354 incl. the additive masks of :889-948 (self: 0/-10000, causal
 * for sequences >= causal_from) and invert_attention_mask :1038-1043 (cross: 0/finfo.min, is_cross=1).
 * kv_seq (optional, [nseq]): query sequence s reads the keys/values of sequence kv_seq[s] -- the passes of SPMM.forward
 * that cross-attend to the same encoder_hidden_states (SPMM_models.py:139-150,181-199,224-231,245-250) share one K/V
 * projection instead of four.  kmask stays per query sequence; dK/dV are written per query sequence.
 * q_row0/q_len ([nseq]) and kv_row0/kv_len ([number of key/value sources]), optional pairs: packed variable-length
 * layouts -- sequence s owns q_len[s] <= Lq rows from row q_row0[s] of Q/O/dO/dQ, source u owns kv_len[u] <= Lkv rows from
 * row kv_row0[u] of K/V (and of dK/dV unless kv_seq is given).  Rows of padding tokens whose outputs never reach a loss
 * (SPMM_models.py:139-206 read only position 0 of those passes) are then simply not computed.
 * Sequences longer than 256 run as <= 128-long query / key chunks over several launches (spmm_amd/ops.py::attn_fwd_long):
 * q_off / kv_off give the chunk's position for the causal mask; backward d_mode 1 writes only D[q] = sum_kv P dP of this key
 * chunk to Dbuf [nseq, nH, Lq], d_mode 2 reads the D summed over all key chunks from Dbuf (0: computed in-kernel; d_mode != 0
 * takes Lq, Lkv <= 128). */
int spmm_attn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const int* kmask,
                  const int* kv_seq, const int* q_row0, const int* q_len, const int* kv_row0, const int* kv_len, void* O,
                  long ldo, float* LSE, int nseq, int nH, int Lq, int Lkv, int causal_from, int is_cross, float dropout_p,
                  const uint64_t* seed_ptr, uint64_t seed_salt, int q_off, int kv_off, spmm_stream_t stream);
int spmm_attn_bwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const int* kmask,
                  const int* kv_seq, const int* q_row0, const int* q_len, const int* kv_row0, const int* kv_len,
                  const void* O, long ldo, const float* LSE, const void* dO, long lddo, void* dQ,
                  long lddq, void* dK, long lddk, void* dV, long lddv, int nseq, int nH, int Lq, int Lkv, int causal_from,
                  int is_cross, float dropout_p, const uint64_t* seed_ptr, uint64_t seed_salt, int q_off, int kv_off, int d_mode,
                  float* Dbuf, spmm_stream_t stream);

/* Fused cross-attention block, forward (csrc/xattn.hip) -- the unit BASELINE.json's metric names, one launch:
 *   Y = LayerNorm(dropout_h(dropout_a(softmax(Q K^T / 8 + mask)) V . Wo^T + bo) + R)
 * Replaces BertSelfAttention.forward xbert.py:305-354 (cross instantiation :285-290, encoder mask 0/finfo.min from
 * invert_attention_mask :1038-1043) followed by BertSelfOutput.forward xbert.py:369-373.  Q (already projected, xbert.py:280)
 * and K / V (:286-287, projected once per unique source) come from GEMM launches.  Layout arguments as spmm_attn_fwd
 * (kv_seq, packed q_row0/q_len, kv_row0/kv_len); R / Y / Z / mean / rstd / CTX are indexed by query row like Q.
 * WoF = spmm_xattn_pack_wo(attention.output.dense.weight).  Optional outputs (null = not kept): Z (pre-LayerNorm sum, what
 * spmm_ln_bwd reads), mean/rstd, CTX (attention context: weight gradient of Wo, spmm_attn_bwd), LSE.  The two dropouts draw the
 * masks spmm_attn_fwd (salt_a) and spmm_ln_fwd (salt_h, row counter row_base + row) would draw, so the existing backward kernels
 * regenerate them.  H = nH*64 in {128, 256, 768}, Lkv <= 128 (spmm_xattn_supported). */
int spmm_xattn_supported(int H, int nH, int Lq, int Lkv);
int spmm_xattn_pack_wo(const void* W, long ldw, void* out, int H, spmm_stream_t stream);
int spmm_xattn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const int* kmask,
                   const int* kv_seq, const int* q_row0, const int* q_len, const int* kv_row0, const int* kv_len,
                   const void* WoF, const float* bo, const void* R, long ldr, const float* gamma, const float* beta, float eps,
                   void* Y, long ldy, void* Z, long ldz, float* mean, float* rstd, void* CTX, long ldc, float* LSE,
                   int nseq, int nH, int Lq, int Lkv, float attn_dropout_p, uint64_t salt_a, float hidden_dropout_p,
                   uint64_t salt_h, const uint64_t* seed_ptr, long row_base, spmm_stream_t stream);
/* out[u] = sum over k in [start[u], start[u+1]) of src[list[k]]  (rows of W bf16 elements, fp32 accumulation, fixed order).
 * Folds the per-query-sequence dK/dV of a cross-attention whose sequences share key/value sources (kv_seq) back onto the
 * unique sources before the K/V weight- and data-gradient GEMMs. */
int spmm_segment_sum_bf16(const void* src, const int* start, const int* list, void* out, int U, long W, spmm_stream_t stream);

/* y = LayerNorm(dropout(x) + res): BertSelfOutput xbert.py:369-373, BertOutput :447-451, transform LN :675,
 * property_mtr_head LN SPMM_models.py:41.  zout (may alias x) keeps the pre-norm sum for backward. */
int spmm_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* zout, float* mean,
                float* rstd, long rows, int H, float eps, float dropout_p, const uint64_t* seed_ptr, uint64_t salt,
                const int* rows_dev, spmm_stream_t stream);
/* The same with the fp32 residual stream (EngineOptions.resid_fp32): res32 is read in fp32, the normalised row is
 * written as bf16 (y: the MFMA operand of the next GEMM) AND fp32 (y32: the next residual / the loss heads' input; may be null). */
int spmm_ln_fwd_r32(const void* x, const float* res32, const float* gamma, const float* beta, void* y, float* y32, void* zout,
                    float* mean, float* rstd, long rows, int H, float eps, float dropout_p, const uint64_t* seed_ptr,
                    uint64_t salt, spmm_stream_t stream);
/* dz = dLN(dy + dy2); dx = dropout-mask(dz) when drop_on_dy == 0; drop_on_dy == 1 masks dy instead (embeddings);
 * dgamma/dbeta accumulate with atomics (may be null for frozen parameters); dxsum (optional) += column sums of dx,
 * i.e. the bias gradient of the dense layer whose output was normalised.
 * beta_from_y (optional, [H]): `z` then holds the LayerNorm's OUTPUT y (what spmm_ln_fwd wrote, no dropout behind it) and the normalised
 * values are recovered as (y - beta) / gamma (0 where gamma == 0); `mean` may be null and spmm_ln_fwd need not keep its pre-norm sum. */
int spmm_ln_bwd(const void* dy, const void* dy2, const void* z, const float* mean, const float* rstd, const float* gamma,
                void* dz, void* dx, float* dgamma, float* dbeta, long rows, int H, float dropout_p,
                const uint64_t* seed_ptr, uint64_t salt, int drop_on_dy, float* dxsum, const int* rows_dev,
                const float* beta_from_y, spmm_stream_t stream);

/* One decode step of BertEmbeddings.forward (xbert.py:193-220) at inference: y[r] = LN(word[ids[r]] + pos[pos_index] +
 * type[0]) for `rows` single-token rows (every beam is at the same position).  pos_ptr (optional) overrides pos_index
 * with a device-resident step counter so that a captured hipGraph of the decode step can be replayed. */
int spmm_embed_step_ln_fwd(const int* ids, int pos_index, const int* pos_ptr, const float* word, const float* pos, const float* type0,
                           const float* gamma, const float* beta, void* y, long rows, int H, float eps, spmm_stream_t stream);

/* Single-query attention over a key/value cache (xbert.py:305-354 for the newest position only; the cache slots the
 * reference sketches at xbert.py:291-295,480,1344-1348).  Row r, head h: softmax_j(q_r,h . K[s(r,j), j, h] * scale) V[...],
 * j < Lkv <= 256, head_dim 64.  Key/value (s, j) of head h lives at element offset s*seq_stride + j*tok_stride + h*head_stride from K / V (token-major rows:
 * tok_stride = row width, head_stride = 64; head-major caches [S, nH, L, 64]: tok_stride = 64, head_stride = L*64).
 * s(r,j) = anc[r*anc_ld + j] (self-attention: beam ancestry table, cache rows are never moved) or r / kv_div when anc is
 * null (cross-attention: the k beams of a molecule share its PV keys/values).  `group` (R % group == 0): rows
 * n*group .. n*group+group-1 are the beams of one molecule -- served by one wave per head (group 2..8), which loads a key row they share once.
 * No mask: beams carry no padding.  t_ptr (optional, device int): the number of valid keys is *t_ptr + 1 (<= Lkv) instead of
 * Lkv -- the step counter of a replayed hipGraph.  knew / vnew (optional, with anc; row stride ldn): key and value of the newest
 * position (the last valid one) of every row, straight from the projection output: every row attends ITS OWN newest key / value
 * there (the table's entry for that position must name the row's own cache row: s(r, last) = r, or rowmap[r]), and the launch copies
 * them into that cache row for the positions to come, so the caller needs no cache-update copies of its own.  rowmap (optional, [R]):
 * the cache row row r's newest position goes to (s = rowmap[r]) -- after the caller dropped finished molecules from its batch, the
 * rows it still decodes keep writing to the cache rows their ancestry tables name. */
int spmm_decode_attn(const void* q, long ldq, const void* K, const void* V, long seq_stride, long tok_stride, long head_stride, const int* anc,
                     int anc_ld, int kv_div, int group, void* out, long ldo, int R, int nH, int Lkv, float scale,
                     const int* t_ptr, const void* knew, const void* vnew, long ldn, const int* rowmap, spmm_stream_t stream);

/* One position of the k-beam PV -> SMILES search for N molecules (d_pv2smiles_batched.py:36-50; the top-k branch of
 * `generate`, d_pv2smiles_single.py:41-44) in one launch.  logits [N*k, V] fp32 (row stride ldl): next-token logits of every beam.
 * Per molecule: log softmax of the k most probable successors of every beam on top of the beam's score cur_p -> k*k candidates;
 * candidates whose token is [SEP] (id 3) are appended to the molecule's finals in row-major order (fin_p / fin_len / fin_tok,
 * F + 1 slots per molecule, slot F a dump; fin_n counts) and struck out with -1e5; a molecule holding >= k finals is done (done[n] = 1,
 * *n_done counts them) and keeps its state; otherwise the k best candidates become the new beams: tokens [N, k, Lmax] (int32
 * histories; position t receives the new token), cur_p, ids_out [N*k] (the token to feed next; 0 for finished molecules),
 * parent_out (optional) and the K/V ancestry table anc [N*k, anc_ld] of spmm_decode_attn (optional: positions < t inherited from
 * the parent row, positions >= t the row itself; the beams of a molecule that finishes at this position all take beam 0's row of the table --
 * nobody reads their outputs any more, and one shared row per position is the cheap case of spmm_decode_attn).  t = tokens held by every live beam = *t_ptr + t_off when t_ptr is given (a
 * replayed hipGraph), else the argument.  mol (optional, [N]): the batch is a compacted subset -- molecule i of logits / ids_out / anc is
 * molecule mol[i] of the state arrays; rowmap (optional, [N*k]): the K/V cache row of beam row i*k + b (what "the row itself" means in
 * anc).  k <= 8, k <= V <= 512, Lmax <= 256. */
int spmm_beam_step(const float* logits, long ldl, int N, int k, int V, int Lmax, int F, int t, const int* t_ptr, int t_off,
                   int* tokens, float* cur_p, float* fin_p, int* fin_len, int* fin_tok, int* fin_n, unsigned char* done,
                   int* anc, int anc_ld, int* ids_out, int* parent_out, int* n_done, const int* mol, const int* rowmap,
                   spmm_stream_t stream);

/* mode 0: BertEmbeddings.forward xbert.py:193-220 from token ids.  mode 1: the PV path -- property_embed Linear(1,H),
 * bernoulli mask blend with property_mask, property_cls prepend (SPMM_models.py:82-88) fused with BertEmbeddings
 * (inputs_embeds branch).  Sequence s reads PV source row s % src_mod.  mode 2: generic inputs_embeds branch of
 * BertEmbeddings (pv_x = fp32 inputs_embeds [nseq*L, H]). */
int spmm_embed_ln_fwd(int mode, const int* ids, const float* word, const float* pos, const float* type0, const float* pv_x,
                      const float* pv_mask, const float* pv_w, const float* pv_b, const float* pv_cls,
                      const float* pv_masktok, int src_mod, const float* gamma, const float* beta, void* y, void* zout,
                      float* mean, float* rstd, long nseq, int L, int H, float eps, float dropout_p,
                      const uint64_t* seed_ptr, uint64_t salt, spmm_stream_t stream);
int spmm_embed_bwd(int mode, const void* dz, const int* ids, const float* pv_x, const float* pv_mask, int src_mod,
                   float* dword, float* dpos, float* dtype0, float* d_w, float* d_b, float* d_cls, float* d_masktok,
                   long nseq, int L, int H, spmm_stream_t stream);

/* layout helpers for the NT-only GEMM: activations^T for wgrad (+ fused bias-gradient column sums), fp32 master
 * weight -> bf16 shadow and transposed bf16 shadow for dgrad, casts, row gather / scatter-add
 * (SPMM_models.py:164-183 negatives; their backward). */
int spmm_transpose_bf16(const void* in, long ldi, void* out, long ldo, int R, int C, int Rpad, float* colsum,
                        spmm_stream_t stream);
int spmm_cast_transpose(const float* in, void* out, void* outT, int R, int C, spmm_stream_t stream);
/* all transposed bf16 weight shadows in one launch; descs_dev: device array of {const float* src; bf16* dstT; int R, C, tile0, ntc} */
long spmm_cast_transpose_desc_bytes(void);
int spmm_cast_transpose_multi(const void* descs_dev, int ndesc, int total_tiles, spmm_stream_t stream);
int spmm_cast_f32_bf16(const float* in, void* out, long n, spmm_stream_t stream);
int spmm_cast_bf16_f32(const void* in, float* out, long n, spmm_stream_t stream);
int spmm_acc_rows(float* dst, long ldd, const void* src, long lds, const long* idx, long rows, int H, int atomic,
                  spmm_stream_t stream);
int spmm_gather_rows(void* dst, const void* src, const long* idx, long rows, int H, spmm_stream_t stream);

/* Row bookkeeping of a step as device kernels (csrc/plan.hip) -- what SPMM.forward does with torch.cat / indexing on whole
 * tensors (SPMM_models.py:139-150,166-199: the ITM passes' query / key-value pairings and hard negatives; :95,105,201: only
 * position 0 of those passes is read), here on packed rows and without a tensor-library launch per index array.
 *  spmm_gather_rows2: dst[r] = idx[r] < 0 ? 0 : (idx[r] & 2^40 ? srcB : srcA)[idx[r] & (2^40 - 1)]  (rows of H bf16, H % 8 == 0).
 *  spmm_add_rows_bf16: dst[idx[r]] += src[r], every idx at most once.   spmm_zero_bytes / spmm_zero_rows: 16-byte granular fill with zeros
 *   (contiguous / `rows` pieces of row_bytes at a stride).
 *  spmm_gelu_bwd: out = dz * gelu'(pre), erf-GELU (the transform of BertLMPredictionHead xbert.py:673 and the GELU of
 *   property_mtr_head SPMM_models.py:39, backward).
 *  spmm_pack_plan: from the attention mask [B, Lt] and the host's count M of valid tokens: per sequence length / first packed row
 *   (lens32, row0_32, row0_64), rows[M] = dense row of every packed row, gidx2[M + B Lt] / gidx4[2M] = gather indices that build the
 *   [packed | dense] and [packed | packed] batches of the text encoders from their dense [2B Lt] embeddings, inv[2 B Lt] = packed row of
 *   every dense row (-1: padding) for the way back, idx_m[B + M] = rows the momentum text encoder's last layer keeps of its [packed | packed]
 *   batch (position 0 of the first copy, every row of the second); *bad |= 1 when the mask is not B non-empty prefixes with M tokens in all.
 *  spmm_fusion_plan: index arrays of the fusion batch (layout in csrc/plan.hip) from the sampled negatives neg[2B] (prop | text):
 *   idx6 (assembly gather over A = [prop_embeds ; prop_embeds_causal], B = [text_embeds ; hidden10]; the text negatives re-enter as PACKED
 *   query rows at the end of the batch, Mn = sum of their lengths known only on the device), neg_rows [B Lt] (row of text_embeds behind
 *   every packed negative row, -1 past Mn), idx_top (rows the top fusion layer keeps: position 0 of the 6B ITM sequences, every row of the
 *   LM and causal-PV passes), small32 [35 B + 2] (sequence -> key/value source maps, packed row tables, CSR inverse maps of the two shared
 *   key/value sources, and rows_dev = {rows of the batch, Mn}: the device-side row counts of the launches over it). */
int spmm_gather_rows2(void* dst, const void* srcA, const void* srcB, const long* idx, long rows, int H, spmm_stream_t stream);
int spmm_add_rows_bf16(void* dst, const long* idx, const void* src, long rows, int H, spmm_stream_t stream);
int spmm_zero_bytes(void* p, long nbytes, spmm_stream_t stream);
int spmm_zero_rows(void* p, long rows, long row_bytes, long stride_bytes, spmm_stream_t stream);
int spmm_gelu_bwd(const void* dz, const void* pre, void* out, long n, spmm_stream_t stream);
int spmm_pack_plan(const int* mask, int B, int Lt, int M, int* lens32, int* row0_32, long* row0_64, long* rows, long* gidx2,
                   long* gidx4, long* inv, long* idx_m, int* bad, spmm_stream_t stream);
int spmm_fusion_plan(const long* neg, const int* lens32, const int* row0_32, int B, int Lt, int Lp, int M,
                     long* idx6, long* neg_rows, long* idx_top, int* small32, spmm_stream_t stream);

/* F.normalize(proj(cls), dim=-1) SPMM_models.py:92,95,101,105; also emits split-bf16 GEMM operands. */
int spmm_l2norm_fwd(const float* x, long ldx, float* y, float* nrm, void* a3, void* w3, void* yT, long ldt, int rows, int E,
                    spmm_stream_t stream);
int spmm_l2norm_bwd(const float* dy, const float* y, const float* nrm, const float* gscale, void* dx, int rows, int E,
                    spmm_stream_t stream);
/* soft-target contrastive loss rows + gradient (SPMM_models.py:113-131). */
int spmm_ita_rows(const float* S, const float* SM, long ldj, int nrows, int B, int J, const float* alpha_ptr,
                  const float* temp_ptr, void* dS, long ldd, int Jpad, float* losses, int loss_slot, float* dtemp,
                  int* nan_flag, spmm_stream_t stream);
/* hard negatives SPMM_models.py:154-178: one multinomial draw per row, on device. */
int spmm_sample_neg(const float* S, long ldj, int B, const long* forced, const uint64_t* seed_ptr, uint64_t salt, long* out,
                    long out_offset, spmm_stream_t stream);
/* next-token CE + distillation SPMM_models.py:233-238 and its gradient w.r.t. the student logits. */
int spmm_lm_loss(const float* logits, const float* logits_m, long ldl, const int* ids, long nseq, int L, int V,
                 const float* alpha_ptr, int* n_nonpad_ws, const float* gscale, void* dlogits, long ldd, int Vpad,
                 float* losses, int loss_slot, spmm_stream_t stream);
/* itm_head + cross entropy SPMM_models.py:201-206 (forward and backward in one pass). */
int spmm_itm_head(const void* xa, long stride_a, const void* xb, long stride_b, int H, const float* W, const float* bias,
                  int n, int B, const float* gscale, float* losses, int loss_slot, float* logits_out, void* dxa, void* dxb,
                  float* dW, float* db, int do_bwd, int x_is_f32, spmm_stream_t stream);
/* property_mtr_head final Linear(H,1) + masked MSE * 5 SPMM_models.py:251-256. */
int spmm_mpm_head(const void* h, int Lp, int H, const float* w, const float* bias, const float* target, const float* mask,
                  int B, int* n_keep_ws, const float* gscale, float* losses, int loss_slot, float* pred_out, void* dh,
                  float* dw, float* db, int do_bwd, int x_is_f32, spmm_stream_t stream);
/* Small dense heads of the inference tier: out[r,n] = act(bias[n] + sum_k x[r,k] W[n,k]) in fp32 (x fp32 or bf16; act 0 = none,
 * 1 = erf-GELU) -- property_embed / property_proj / text_proj / itm_head / property_mtr_head called as modules on a few rows
 * (SPMM_models.py:36-43; d_smiles2pv.py:15-25, d_pv2smiles_batched.py:25-27). */
int spmm_rows_linear(const void* x, int x_is_bf16, long ldx, const float* W, const float* bias, float* out, long ldo, long rows,
                     int N, int K, int act, spmm_stream_t stream);
/* _dequeue_and_enqueue SPMM_models.py:272-286 (+ the bf16 GEMM shadows of the queue).  skip_flag (optional, device int): when
 * non-zero neither the queue nor the pointer is touched -- the reference's NaN guard returns before the enqueue
 * (SPMM_models.py:132-134 vs :208), so a non-finite momentum feature never enters the queue. */
int spmm_enqueue(const float* feats, int n, int E, float* queue, int Q, void* w3, void* qT, long ldt, int Bloc, long* ptr,
                 int advance, const int* skip_flag, spmm_stream_t stream);
int spmm_queue_shadow(const float* queue, int E, int Q, void* w3, void* qT, long ldt, int Bloc, spmm_stream_t stream);
/* temp.clamp_(0.01, 0.5) SPMM_models.py:80-81 */
int spmm_clamp_scalar(float* p, float lo, float hi, spmm_stream_t stream);

/* clip_grad_norm_(5.) + AdamW SPMM_models.py:340,361-362 and the EMA of the momentum encoders :266-269. */
long spmm_adam_scalars_bytes(void);
long spmm_grad_sqnorm_workspace_bytes(void);
/* deterministic (fixed-order) sum of squares: replicas with identical gradients get identical clip coefficients */
int spmm_grad_sqnorm(const float* g, long n, float* out_zeroed, float* workspace, spmm_stream_t stream);
int spmm_adamw_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, const float* lr_ptr,
                    float beta1, float beta2, float eps, float weight_decay, const float* normsq, float max_norm, int* step,
                    const int* nan_flag, void* scalars_ws, spmm_stream_t stream);
int spmm_ema_update(float* pm, const float* p, void* bf16_shadow, long n, float momentum, spmm_stream_t stream);
int spmm_axpy_scalar(float* dst, const float* src, const float* scale_ptr, float scale, int n, spmm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPMM_HIP_H */
