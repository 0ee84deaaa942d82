// bf16 "NT" GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (both operands K-contiguous, the
// nn.Linear layout), fp32 accumulate on v_mfma_f32_32x32x16_bf16, fused epilogues.
//
// Replaces on the reference path: every nn.Linear of xbert.py (query/key/value :280-300, attention
// output.dense :370, intermediate.dense :435, output.dense :448, LM transform/decoder :673,:695) and their
// dgrad / wgrad GEMMs in backward (dgrad uses the [K,N] transposed bf16 weight shadow, wgrad the
// transposed activations, so all three are NT problems), and the similarity GEMMs SPMM_models.py:108-124.
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2 MFMA 32x32 tiles,
// 64 fp32 accumulators per lane).  A/W tiles are staged HBM->LDS with global_load_lds_dwordx4 (no VGPR
// round trip), double buffered (2 x 32 KiB).  LDS rows are 128 B, so the 16-B slot index is XOR-swizzled
// with (row>>1)&7 -- applied on the global SOURCE address (LDS-DMA writes lane-linear) and again on the
// ds_read_b128 address -- which makes every 16-lane ds_read_b128 group hit 16 distinct slots.
// The MFMA is issued "swapped" (A-operand = W rows, B-operand = A rows) so each lane ends up with 4
// consecutive output columns per accumulator quad -> 8-byte bf16 / 16-byte fp32 stores.
// Workgroup ids are remapped so that consecutive tiles (same A row panel) run on the same XCD / L2.
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + W

enum Epi { EPI_BF16 = 0, EPI_GELU = 1, EPI_F32 = 2, EPI_F32_ATOMIC = 3, EPI_GELU_GRAD = 4, EPI_F32_ACC = 5 };

struct GemmP {
  const bf16* A; long lda;
  const bf16* W; long ldw;
  int M, N, K;               // this launch reduces K elements per split-slice; K % 64 == 0
  int ksplit;                // elements of K per grid.z slice (multiple of 64)
  const float* bias;         // [N] or null
  const float* div_ptr;      // optional device scalar: acc /= *div_ptr
  float alpha;               // acc *= alpha
  const bf16* R; long ldr;   // optional bf16 addend (residual-gradient accumulate)
  const bf16* G; long ldg;   // EPI_GELU_GRAD: pre-activation
  void* C; long ldc;
  bf16* C2; long ldc2;       // EPI_GELU: pre-activation output
  float* colsum;             // optional: colsum[n] += sum_m C[m][n] of the (bf16-rounded) output -- bias gradient of the producing layer
  int order;                 // tile order inside an XCD's range: 0 n-fastest, 1 m-fastest, 2 blocked (8 m-panels x GN n-tiles)
  int krot;                  // v3: tiles start their k loop at different slices (de-synchronises the CUs' walks over shared W lines)
  int stagger;               // v3: every other first-wave workgroup sleeps stagger x ~3.9 us so the chip's epilogue store bursts interleave
};

// LDS-DMA staging: 128 rows x 8 slots(16 B) = 1024 chunks, 4 per thread; chunk id -> (row = id>>3,
// physical slot = id&7).  The wave's 64 chunks are contiguous in LDS (wave-uniform base + lane*16).
__device__ __forceinline__ void stage_tile_dma(const bf16* __restrict__ src, long ld, int row0, int nrows, int k0,
                                               char* lds_tile, int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = c * 256 + tid;
    const int row = id >> 3, ps = id & 7;
    const int ls = ps ^ ((row >> 1) & 7);                 // logical k-slot held at this physical slot
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;               // clamp: out-of-range rows are never stored
    const bf16* g = src + (long)grow * ld + k0 + ls * 8;
    const int wave_base = __builtin_amdgcn_readfirstlane((c * 256 + (tid & ~63)) * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lds_tile + wave_base), 16, 0, 0);
  }
}
// Register-staged fallback (same LDS image): load early, write after the MFMA block.
__device__ __forceinline__ void stage_tile_load(const bf16* __restrict__ src, long ld, int row0, int nrows, int k0,
                                                int tid, bf16x8 (&r)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = c * 256 + tid;
    const int row = id >> 3, ps = id & 7;
    const int ls = ps ^ ((row >> 1) & 7);
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;
    r[c] = *(const bf16x8*)(src + (long)grow * ld + k0 + ls * 8);
  }
}
__device__ __forceinline__ void stage_tile_write(char* lds_tile, int tid, const bf16x8 (&r)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) *(bf16x8*)(lds_tile + (c * 256 + tid) * 16) = r[c];
}


// Coalesced bf16 epilogue: a wave's 64x64 output tile goes through LDS (8 KiB per wave per output) so that global
// stores are 16 B per lane along rows (8 lanes = one 128-B row segment) instead of 8-B fragments scattered over 32 rows.
// acc[ni][mi][r] = D[n][m], m = lane&31, n = (r&3) + 8*(r>>2) + 4*(lane>>5).   Tile rows are 128 B; the 16-B chunk index is
// XOR-ed with (row & 7) so the row-parallel writes and the row-major reads both spread over the banks.
// Split in two halves so that a kernel can let DIFFERENT waves do the second half (wave-specialised stores, v6):
//   epi_convert: accumulators -> scale / bias / GELU -> bf16 tile in LDS       (GMODE 1: pre-activation, 2: activation only)
//   epi_store  : LDS tile -> (+R, * gelu'(G)) -> row-coalesced global stores (+ fused column sums)
template <int EPI, int GMODE = 0>
__device__ __forceinline__ void epi_convert(const GemmP& p, f32x16 (&acc)[2][2], char* t0, char* t1, int n_base, int lane,
                                            const float* bias_lds = nullptr) {
  float scale = p.alpha;
  if (p.div_ptr) scale /= *p.div_ptr;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int row = mi * 32 + (lane & 31);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = ni * 32 + 8 * g + 4 * (lane >> 5);       // 4 consecutive n
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][g * 4 + j] * scale;
        if (p.bias) {
          const int n = n_base + col;
          if (n < p.N) {
            const f32x4 b = bias_lds ? *(const f32x4*)(bias_lds + n) : *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += b[j];
          }
        }
        const int off = row * 128 + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 4) << 1));
        if constexpr (EPI == EPI_GELU && GMODE == 1) {
          *(bf16x4*)(t0 + off) = to_bf16x4(v[0], v[1], v[2], v[3]);
        } else if constexpr (EPI == EPI_GELU && GMODE == 2) {
          *(bf16x4*)(t0 + off) = to_bf16x4(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
        } else if constexpr (EPI == EPI_GELU) {
          if (p.C2) *(bf16x4*)(t1 + off) = to_bf16x4(v[0], v[1], v[2], v[3]);
          *(bf16x4*)(t0 + off) = to_bf16x4(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
        } else {
          *(bf16x4*)(t0 + off) = to_bf16x4(v[0], v[1], v[2], v[3]);
        }
      }
    }
}

// cs_carry (optional): column-sum partials kept in registers across consecutive calls on the SAME columns (the two m-halves of
// a wave tile); they are flushed with atomics only when `flush` is set -- half the same-address atomics per tile.
template <int EPI, bool NOSTORE = false, int GMODE = 0>
__device__ __forceinline__ void epi_store(const GemmP& p, const char* t0, const char* t1, int m_base, int n_base, int lane,
                                          float* cs_carry = nullptr, bool flush = true) {
  // all R / G operand loads first: a load placed behind a store to a possibly aliasing pointer would be serialised behind it
  bf16x8 rr[8], gg[8];
  if (p.R || EPI == EPI_GELU_GRAD) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), c16 = lane & 7;
      const int m = m_base + row, n = n_base + c16 * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { rr[it][e] = (bf16)0.f; gg[it][e] = (bf16)0.f; }
      if (m >= p.M || n >= p.N) continue;
      const bool full = n + 8 <= p.N;
      if (p.R) {
        if (full) rr[it] = *(const bf16x8*)(p.R + (long)m * p.ldr + n);
        else { const bf16x4 h4 = *(const bf16x4*)(p.R + (long)m * p.ldr + n); rr[it][0] = h4[0]; rr[it][1] = h4[1]; rr[it][2] = h4[2]; rr[it][3] = h4[3]; }
      }
      if constexpr (EPI == EPI_GELU_GRAD) {
        if (full) gg[it] = *(const bf16x8*)(p.G + (long)m * p.ldg + n);
        else { const bf16x4 h4 = *(const bf16x4*)(p.G + (long)m * p.ldg + n); gg[it][0] = h4[0]; gg[it][1] = h4[1]; gg[it][2] = h4[2]; gg[it][3] = h4[3]; }
      }
    }
  }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cs_carry) {
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cs_carry[e];
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), c16 = lane & 7;
    const int m = m_base + row, n = n_base + c16 * 8;
    if (m >= p.M || n >= p.N) continue;
    const int off = row * 128 + ((c16 ^ (row & 7)) << 4);
    bf16x8 o = *(const bf16x8*)(t0 + off);
    const bool full = n + 8 <= p.N;
    if (p.R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)o[e] + (float)rr[it][e]);
    }
    if constexpr (EPI == EPI_GELU_GRAD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)o[e] * gelu_erf_grad((float)gg[it][e]));
    }
    if (p.colsum) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += (float)o[e];
    }
    if constexpr (NOSTORE) { asm volatile("" :: "v"(o)); continue; }
    bf16* dst = (EPI == EPI_GELU && GMODE == 1) ? p.C2 + (long)m * p.ldc2 + n : (bf16*)p.C + (long)m * p.ldc + n;
    if (full) {
      *(bf16x8*)dst = o;
      if constexpr (EPI == EPI_GELU && GMODE == 0) { if (p.C2) *(bf16x8*)(p.C2 + (long)m * p.ldc2 + n) = *(const bf16x8*)(t1 + off); }
    } else {   // ragged last chunk (N % 8 == 4)
      bf16x4 lo4; lo4[0] = o[0]; lo4[1] = o[1]; lo4[2] = o[2]; lo4[3] = o[3];
      *(bf16x4*)dst = lo4;
      if constexpr (EPI == EPI_GELU && GMODE == 0) { if (p.C2) *(bf16x4*)(p.C2 + (long)m * p.ldc2 + n) = *(const bf16x4*)(t1 + off); }
    }
  }
  if (p.colsum && cs_carry && !flush) {
#pragma unroll
    for (int e = 0; e < 8; ++e) cs_carry[e] = cs[e];
    return;
  }
  if (p.colsum) {   // lanes l, l^8, l^16, l^32 hold the same 8 columns for different rows
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      cs[e] = v;
    }
    if (lane < 8) {
      const int n = n_base + lane * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n + e < p.N) atomicAdd(p.colsum + n + e, cs[e]);
    }
  }
}

// both halves by the same wave on its private region (same-wave LDS round trip: the compiler's lgkmcnt waits suffice)
template <int EPI, bool NOSTORE = false, int GMODE = 0>
__device__ __forceinline__ void epilogue_bf16(const GemmP& p, f32x16 (&acc)[2][2], char* wtile, int m_base, int n_base, int lane,
                                              float* cs_carry = nullptr, bool flush = true) {
  epi_convert<EPI, GMODE>(p, acc, wtile, wtile + 8192, n_base, lane);
  epi_store<EPI, NOSTORE, GMODE>(p, wtile, wtile + 8192, m_base, n_base, lane, cs_carry, flush);
}

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile id: workgroup b runs on XCD b%8; give each XCD a contiguous range of tiles.
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tile_m = t / ntn, tile_n = t % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = blockIdx.z * p.ksplit;
  const int kend = min(p.K, kbeg + p.ksplit);
  const int nk = (kend - kbeg) / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 ra[4], rw[4];
  if (nk > 0) {
    if constexpr (GLDS) {
      stage_tile_dma(p.A, p.lda, m0, p.M, kbeg, smem, tid);
      stage_tile_dma(p.W, p.ldw, n0, p.N, kbeg, smem + TILE_BYTES, tid);
    } else {
      stage_tile_load(p.A, p.lda, m0, p.M, kbeg, tid, ra);
      stage_tile_load(p.W, p.ldw, n0, p.N, kbeg, tid, rw);
      stage_tile_write(smem, tid, ra);
      stage_tile_write(smem + TILE_BYTES, tid, rw);
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();   // tile kt landed (vmcnt(0) is part of the barrier's release while LDS-DMA is pending)
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    if (kt + 1 < nk) {
      if constexpr (GLDS) {
        stage_tile_dma(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, nxt, tid);
        stage_tile_dma(p.W, p.ldw, n0, p.N, kbeg + (kt + 1) * BK, nxt + TILE_BYTES, tid);
      } else {
        stage_tile_load(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, tid, ra);
        stage_tile_load(p.W, p.ldw, n0, p.N, kbeg + (kt + 1) * BK, tid, rw);
      }
    }
    const char* As = cur;
    const char* Ws = cur + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int s = kk * 2 + (lane >> 5);
      bf16x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ar = wm * 64 + i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(As + ar * 128 + ((s ^ ((ar >> 1) & 7)) << 4));
        const int wr = wn * 64 + i * 32 + (lane & 31);
        wf[i] = *(const bf16x8*)(Ws + wr * 128 + ((s ^ ((wr >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    if constexpr (!GLDS) {
      if (kt + 1 < nk) {   // the other buffer was last read in iteration kt-1, before this iteration's barrier
        stage_tile_write(nxt, tid, ra);
        stage_tile_write(nxt + TILE_BYTES, tid, rw);
      }
    }
  }

  if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_GELU_GRAD) {
    __syncthreads();                                  // every wave is done reading the staging buffers
    epilogue_bf16<EPI>(p, acc, smem + wave * 16384, m0 + wm * 64, n0 + wn * 64, lane);
    return;
  }
  // ---- epilogue: acc[ni][mi][r] = D[n][m], m = lane&31, n = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float scale = p.alpha;
  if (p.div_ptr) scale /= *p.div_ptr;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * (lane >> 5);
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][g * 4 + j] * scale;
        if (p.bias && (EPI != EPI_F32_ATOMIC || blockIdx.z == 0)) {
          const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += b[j];
        }
        if (p.R) {
          const bf16x4 r = *(const bf16x4*)(p.R + (long)m * p.ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
        }
        if constexpr (EPI == EPI_BF16) {
          *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) = to_bf16x4(v[0], v[1], v[2], v[3]);
        } else if constexpr (EPI == EPI_GELU) {
          *(bf16x4*)(p.C2 + (long)m * p.ldc2 + n) = to_bf16x4(v[0], v[1], v[2], v[3]);
          *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) =
              to_bf16x4(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
        } else if constexpr (EPI == EPI_GELU_GRAD) {
          const bf16x4 x = *(const bf16x4*)(p.G + (long)m * p.ldg + n);
          *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) =
              to_bf16x4(v[0] * gelu_erf_grad((float)x[0]), v[1] * gelu_erf_grad((float)x[1]),
                        v[2] * gelu_erf_grad((float)x[2]), v[3] * gelu_erf_grad((float)x[3]));
        } else if constexpr (EPI == EPI_F32) {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = o;
        } else if constexpr (EPI == EPI_F32_ACC) {
          f32x4* dst = (f32x4*)((float*)p.C + (long)m * p.ldc + n);
          f32x4 o = *dst;
          o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
          *dst = o;
        } else {  // EPI_F32_ATOMIC
          float* dst = (float*)p.C + (long)m * p.ldc + n;
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicAdd(dst + j, v[j]);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// v2: 256x128x64 tile, 512 threads (8 waves as 4x2, 64x64 per wave), THREE LDS stages (3 x 48 KiB) filled by LDS-DMA.
// PMC on v1 (93184x3072x768): MFMA pipe 27 % busy, waves parked 55 % of their cycles at the vmcnt(0)+barrier of the
// 2-stage loop.  Here two tiles are always in flight: the wait before the barrier is a COUNTED s_waitcnt vmcnt(6)
// (= the 6 DMA instructions of the newest tile may still be pending) and the barrier is the raw s_barrier, so the
// compiler does not drain the DMA queue (cdna_hip_programming.md: pipelining across barriers).
constexpr int BM2 = 256;
constexpr int A2_BYTES = BM2 * BK * 2;              // 32 KiB
constexpr int STAGE2_BYTES = A2_BYTES + TILE_BYTES; // 48 KiB
constexpr int LDS2_BYTES = 3 * STAGE2_BYTES;        // 144 KiB

__device__ __forceinline__ void stage2_dma(const bf16* __restrict__ src, long ld, int row0, int nrows, int k0, char* lds_tile, int tid,
                                           int nchunks_per_thread) {
  for (int c = 0; c < nchunks_per_thread; ++c) {
    const int id = c * 512 + tid;
    const int row = id >> 3, ps = id & 7;
    const int ls = ps ^ ((row >> 1) & 7);
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;
    const bf16* g = src + (long)grow * ld + k0 + ls * 8;
    const int wave_base = __builtin_amdgcn_readfirstlane((c * 512 + (tid & ~63)) * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lds_tile + wave_base), 16, 0, 0);
  }
}

template <int EPI, int VAR>
__global__ __launch_bounds__(512) void gemm_nt_v2_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem2[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntm = (p.M + BM2 - 1) / BM2, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tile_m = t / ntn, tile_n = t % ntn;
  const int m0 = tile_m * BM2, n0 = tile_n * BN;
  const int nk = p.K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define STAGE2(kt_)                                                                           \
  do {                                                                                        \
    char* b_ = smem2 + ((kt_) % 3) * STAGE2_BYTES;                                            \
    stage2_dma(p.A, p.lda, m0, p.M, (kt_) * BK, b_, tid, 4);                                  \
    stage2_dma(p.W, p.ldw, n0, p.N, (kt_) * BK, b_ + A2_BYTES, tid, 2);                       \
  } while (0)

  STAGE2(0);
  if (nk > 1) STAGE2(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr bool PRIO = (VAR & 1) != 0, LATE_DMA = (VAR & 2) != 0, PREFETCH = (VAR & 4) != 0;
    if (!LATE_DMA && kt + 2 < nk) STAGE2(kt + 2);
    const char* As = smem2 + (kt % 3) * STAGE2_BYTES;
    const char* Ws = As + A2_BYTES;
    const int ar0 = wm * 64 + (lane & 31), wr0 = wn * 64 + (lane & 31);
    auto ldfrag = [&](int kk, bf16x8 (&af)[2], bf16x8 (&wf)[2]) {
      const int s = kk * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ar = ar0 + i * 32, wr = wr0 + i * 32;
        af[i] = *(const bf16x8*)(As + ar * 128 + ((s ^ ((ar >> 1) & 7)) << 4));
        wf[i] = *(const bf16x8*)(Ws + wr * 128 + ((s ^ ((wr >> 1) & 7)) << 4));
      }
    };
    if constexpr (PREFETCH) {
      bf16x8 afA[2], wfA[2], afB[2], wfB[2];
      ldfrag(0, afA, wfA);
#pragma unroll
      for (int kk = 0; kk < 4; kk += 2) {
        ldfrag(kk + 1, afB, wfB);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfA[ni], afA[mi], acc[ni][mi], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (LATE_DMA && kk == 0 && kt + 2 < nk) STAGE2(kt + 2);
        if (kk + 2 < 4) ldfrag(kk + 2, afA, wfA);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfB[ni], afB[mi], acc[ni][mi], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 af[2], wf[2];
        ldfrag(kk, af, wf);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (LATE_DMA && kk == 0 && kt + 2 < nk) STAGE2(kt + 2);
      }
    }
  }
#undef STAGE2

  if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_GELU_GRAD) {
    __builtin_amdgcn_s_barrier();                     // no DMA pending (vmcnt(0) above); all waves done with the ring
    epilogue_bf16<EPI>(p, acc, smem2 + wave * 16384, m0 + wm * 64, n0 + wn * 64, lane);
    return;
  }
  float scale = p.alpha;
  if (p.div_ptr) scale /= *p.div_ptr;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + (lane & 31);
      if (m >= p.M) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * (lane >> 5);
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][g * 4 + j] * scale;
        if (p.bias) {
          const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += b[j];
        }
        if (p.R) {
          const bf16x4 r = *(const bf16x4*)(p.R + (long)m * p.ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
        }
        if constexpr (EPI == EPI_BF16) {
          *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) = to_bf16x4(v[0], v[1], v[2], v[3]);
        } else if constexpr (EPI == EPI_GELU) {
          *(bf16x4*)(p.C2 + (long)m * p.ldc2 + n) = to_bf16x4(v[0], v[1], v[2], v[3]);
          *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) =
              to_bf16x4(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
        } else if constexpr (EPI == EPI_GELU_GRAD) {
          const bf16x4 x = *(const bf16x4*)(p.G + (long)m * p.ldg + n);
          *(bf16x4*)((bf16*)p.C + (long)m * p.ldc + n) =
              to_bf16x4(v[0] * gelu_erf_grad((float)x[0]), v[1] * gelu_erf_grad((float)x[1]),
                        v[2] * gelu_erf_grad((float)x[2]), v[3] * gelu_erf_grad((float)x[3]));
        } else if constexpr (EPI == EPI_F32) {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = o;
        } else {  // EPI_F32_ACC
          f32x4* dst = (f32x4*)((float*)p.C + (long)m * p.ldc + n);
          f32x4 o = *dst;
          o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
          *dst = o;
        }
      }
    }
}

// (tile_m, tile_n) of sequence index t.  Workgroup b runs on XCD b%8 and the remap gives every XCD a contiguous range of t,
// dispatched in increasing order, so the ~32 tiles an XCD runs concurrently are 32 consecutive t.  order 2 makes those a
// block of 8 m-panels x GN n-tiles (each A k-slice first-touched by GN CUs instead of by all n-tiles at once, and reused
// from L2 by the following rounds of the same 8 m-panels).
__device__ __forceinline__ void tile_of(int t, int ntm, int ntn, int order, int& tm, int& tn) {
  if (order == 1) { tn = t / ntm; tm = t - tn * ntm; return; }
  if (order == 2) {
    const int gn = (ntn % 4 == 0) ? 4 : (ntn % 3 == 0 ? 3 : (ntn % 2 == 0 ? 2 : 1));
    const int sr = t / (8 * ntn);
    const int rows = min(8, ntm - sr * 8);
    const int r = t - sr * 8 * ntn;
    const int nb = r / (rows * gn), rr = r - nb * rows * gn;
    tm = sr * 8 + rr % rows;
    tn = nb * gn + rr / rows;
    return;
  }
  if (order >= 3 && ntn >= 8) {
    // ng column groups, each walked row-major: an XCD's chunk of consecutive t stays inside one group, so the W panel it
    // keeps re-reading is <= ~2.4 MB (L2-resident) instead of the whole W (4.7 MB at N=3072, K=768 > the 4-MiB L2).
    const int ng = order == 3 ? 2 : 4, base = ntn / ng, rem = ntn - base * ng;
    int c0 = 0, u = t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g >= ng) break;
      const int cg = base + (g < rem ? 1 : 0), sz = ntm * cg;
      if (u < sz || g == ng - 1) { tm = u / cg; tn = c0 + (u - tm * cg); return; }
      u -= sz; c0 += cg;
    }
  }
  tm = t / ntn; tn = t - tm * ntn;
}

// ------------------------------------------------------------------------------------------------------------
// v3: 256x256x64 tile, 512 threads (8 waves as 2(m) x 4(n), 128x64 per wave = 4x2 MFMA tiles, 128 fp32 accumulators),
// two 64-KiB LDS stages.  Rationale: a k-step of the 256x128 tile pulls 48 KiB through the ~64 B/clk/CU L2->LDS path for
// 1024 MFMA cycles (75 % of the MFMA time at the PEAK L2 rate); 256x256 pulls 64 KiB for 2048 MFMA cycles (37 %), so
// the DMA stream stops being the bottleneck.  One barrier per k-step; the next tile's 8 DMA instructions per thread are
// issued right after it and have a full 2048-cycle compute phase to land.
constexpr int BM3 = 256, BN3 = 256;
constexpr int T3_BYTES = 256 * BK * 2;              // 32 KiB per operand tile
constexpr int STAGE3_BYTES = 2 * T3_BYTES;          // 64 KiB
constexpr int LDS3_BYTES = 2 * STAGE3_BYTES;        // 128 KiB

// one 16-B chunk per lane (one LDS-DMA wave-instruction) of a 256-row operand tile: chunk group c in 0..3
__device__ __forceinline__ void dma_one(const bf16* __restrict__ src, long ld, int row0, int nrows, int k0, char* lds_tile, int tid, int c) {
  const int id = c * 512 + tid;
  const int row = id >> 3, ps = id & 7;
  const int ls = ps ^ ((row >> 1) & 7);
  int grow = row0 + row;
  grow = grow < nrows ? grow : nrows - 1;
  const bf16* g = src + (long)grow * ld + k0 + ls * 8;
  const int wave_base = __builtin_amdgcn_readfirstlane((c * 512 + (tid & ~63)) * 16);
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lds_tile + wave_base), 16, 0, 0);
}

// ABL (timing experiments only, results are garbage): 1 = no DMA, 2 = no LDS fragment reads, 3 = MFMA only
template <int EPI, bool INTERLEAVE, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_nt_v3_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves, wave tile 128 (m) x 64 (n)
  const int ntm = (p.M + BM3 - 1) / BM3, ntn = (p.N + BN3 - 1) / BN3, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  int tile_m, tile_n;
  tile_of(t, ntm, ntn, p.order, tile_m, tile_n);
  const int m0 = tile_m * BM3, n0 = tile_n * BN3;
  const int nk = p.K / BK;
  if (p.stagger && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  f32x16 acc[2][2][2];      // [m half][ni][mi]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

  const int rot = p.krot ? (tile_m * 5 + tile_n * 3) % nk : 0;
#define KOF3(kt_) ((((kt_) + rot) >= nk ? (kt_) + rot - nk : (kt_) + rot) * BK)
#define STAGE3(kt_)                                                                           \
  do {                                                                                        \
    char* b_ = smem3 + ((kt_) & 1) * STAGE3_BYTES;                                            \
    const int k0_ = KOF3(kt_);                                                                \
    stage2_dma(p.A, p.lda, m0, p.M, k0_, b_, tid, 4);                                         \
    stage2_dma(p.W, p.ldw, n0, p.N, k0_, b_ + T3_BYTES, tid, 4);                              \
  } while (0)

  if (ABL != 1 && ABL != 3) STAGE3(0);
  bf16x8 cst;
#pragma unroll
  for (int e = 0; e < 8; ++e) cst[e] = (bf16)(0.001f * (lane + e));
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                   // tile kt landed; everyone finished reading the other stage
    const bool more = kt + 1 < nk;
    if (!INTERLEAVE && more && ABL != 1 && ABL != 3) STAGE3(kt + 1);
    const char* As = smem3 + (kt & 1) * STAGE3_BYTES;
    const char* Ws = As + T3_BYTES;
    char* nxt = smem3 + ((kt + 1) & 1) * STAGE3_BYTES;
    const int knext = KOF3(kt + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int s = kk * 2 + (lane >> 5);
      bf16x8 af[4], wf[2];
      if constexpr (ABL >= 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[i] = cst; asm volatile("" : "+v"(af[i])); }
#pragma unroll
        for (int i = 0; i < 2; ++i) { wf[i] = cst; asm volatile("" : "+v"(wf[i])); }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ar = wm * 128 + i * 32 + (lane & 31);
          af[i] = *(const bf16x8*)(As + ar * 128 + ((s ^ ((ar >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int wr = wn * 64 + i * 32 + (lane & 31);
          wf[i] = *(const bf16x8*)(Ws + wr * 128 + ((s ^ ((wr >> 1) & 7)) << 4));
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[h][ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[h * 2 + mi], acc[h][ni][mi], 0, 0, 0);
        if constexpr (INTERLEAVE) {
          // one LDS-DMA instruction of the next tile behind every group of 4 MFMAs: the 64 DMA wave-instructions of a
          // k-step trickle into the address path while the matrix pipe works instead of blocking all waves up front
          __builtin_amdgcn_sched_barrier(0);
          if (more) {
            if (h == 0) dma_one(p.A, p.lda, m0, p.M, knext, nxt, tid, kk);
            else dma_one(p.W, p.ldw, n0, p.N, knext, nxt + T3_BYTES, tid, kk);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
#undef STAGE3
  __syncthreads();                                     // all waves done with the staging buffers
  char* wt = smem3 + wave * 16384;                     // 16 KiB private epilogue region per wave
  if constexpr (ABL == 5) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[h][i][j]));
    return;
  }
  float cs_carry[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // column sums of both m-halves go out in one set of atomics
#pragma unroll
  for (int h = 0; h < 2; ++h)
    epilogue_bf16<EPI, ABL == 4>(p, acc[h], wt, m0 + wm * 128 + h * 64, n0 + wn * 64, lane, cs_carry, h == 1);
}

// ------------------------------------------------------------------------------------------------------------
// v7: the 256x256x64 tile with FOUR waves (2 x 2), one per SIMD, each owning a 128x128 wave tile = 4 x 4 MFMA tiles
// (256 fp32 accumulators; the 512-register budget of a lone wave per SIMD holds them next to two sets of fragments).
// Why: in v3 every k-step pulls (128 + 64) x 128 B of fragments per wave x 8 waves = 192 KiB out of LDS while the LDS-DMA
// writes 64 KiB into it -- 2048 clocks of the 128 B/clk LDS port, the same as the 2062 MFMA clocks of the step, so the
// LDS port is a co-bottleneck.  128x128 wave tiles read (128 + 128) x 128 B x 4 = 128 KiB (1024 + 512 clocks).  With a
// single wave per SIMD there is no other wave to hide LDS latency, so the fragments of k-group kk+1 are fetched before the
// 16 MFMAs of kk are issued (explicit double buffer).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_v7_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem7[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves, wave tile 128 (m) x 128 (n)
  const int ntm = (p.M + BM3 - 1) / BM3, ntn = (p.N + BN3 - 1) / BN3, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  int tile_m, tile_n;
  tile_of(t, ntm, ntn, p.order, tile_m, tile_n);
  const int m0 = tile_m * BM3, n0 = tile_n * BN3;
  const int nk = p.K / BK;

  f32x16 acc[2][2][2][2];      // [n half][m half][ni][mi]: acc[hn][hm] is the 64x64 block the shared epilogue takes
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][i][j][r] = 0.f;

  // Per-thread source pointers of its 16 chunks of a stage (8 of A, 8 of W), computed once: a k-step only adds BK elements.
  // Chunk c of an operand: tile row (c*256 + tid) >> 3, physical 16-B slot (tid & 7), logical slot swizzled on the source.
  const bf16* gsrc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int op = c >> 3, cc = c & 7;
    const int id = cc * 256 + tid;
    const int row = id >> 3, ps = id & 7;
    const int ls = ps ^ ((row >> 1) & 7);
    const int nrows = op == 0 ? p.M : p.N;
    int grow = (op == 0 ? m0 : n0) + row;
    grow = grow < nrows ? grow : nrows - 1;
    gsrc[c] = (op == 0 ? p.A + (long)grow * p.lda : p.W + (long)grow * p.ldw) + ls * 8;
  }
  auto dma = [&](int c, int kt) {                       // chunk c of stage kt
    char* tile = smem7 + (kt & 1) * STAGE3_BYTES + (c >> 3) * T3_BYTES;
    const int wave_base = __builtin_amdgcn_readfirstlane(((c & 7) * 256 + (tid & ~63)) * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gsrc[c] + kt * BK), (LDS_AS void*)(tile + wave_base), 16, 0, 0);
  };
  auto frags = [&](const char* As, const char* Ws, int kk, bf16x8 (&af)[4], bf16x8 (&wf)[4]) {
    const int s = kk * 2 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ar = wm * 128 + i * 32 + (lane & 31);
      af[i] = *(const bf16x8*)(As + ar * 128 + ((s ^ ((ar >> 1) & 7)) << 4));
      const int wr = wn * 128 + i * 32 + (lane & 31);
      wf[i] = *(const bf16x8*)(Ws + wr * 128 + ((s ^ ((wr >> 1) & 7)) << 4));
    }
  };

#pragma unroll
  for (int c = 0; c < 16; ++c) dma(c, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                   // tile kt landed; everyone finished reading the other stage
    const bool more = kt + 1 < nk;
    const char* As = smem7 + (kt & 1) * STAGE3_BYTES;
    const char* Ws = As + T3_BYTES;
    bf16x8 af[2][4], wf[2][4];
    frags(As, Ws, 0, af[0], wf[0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk + 1 < 4) frags(As, Ws, kk + 1, af[(kk + 1) & 1], wf[(kk + 1) & 1]);
#pragma unroll
      for (int n4 = 0; n4 < 4; ++n4) {
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4)
          acc[n4 >> 1][m4 >> 1][n4 & 1][m4 & 1] =
              __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][n4], af[kk & 1][m4], acc[n4 >> 1][m4 >> 1][n4 & 1][m4 & 1], 0, 0, 0);
        // the lone wave of this SIMD also has to issue the next stage's DMA: two instructions behind each of the first
        // eight MFMA groups of the k-step, so the matrix pipe keeps running while the address path takes them
        if (kk < 2) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) {
            dma((kk * 4 + n4) * 2, kt + 1);
            dma((kk * 4 + n4) * 2 + 1, kt + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  __syncthreads();                                     // all waves done with the staging buffers
  char* wt = smem7 + wave * 16384;                     // 16 KiB private epilogue region per wave
#pragma unroll
  for (int hn = 0; hn < 2; ++hn)
#pragma unroll
    for (int hm = 0; hm < 2; ++hm)
      epilogue_bf16<EPI>(p, acc[hn][hm], wt, m0 + wm * 128 + hm * 64, n0 + wn * 128 + hn * 64, lane);
}

// ------------------------------------------------------------------------------------------------------------
// v5: PERSISTENT v3.  One 512-thread workgroup per CU walks its share of the 256x256 tiles; the k-steps of consecutive
// tiles form one continuous double-buffered stream, so the next tile's first stage is already in flight while the current
// tile's epilogue runs, no workgroup dispatch sits between tiles, and the epilogue's (fire-and-forget) global stores drain
// under the next tile's MFMAs.  Motivation (ablation on 93184x3072x768): 790 TF as launched per tile, 1035 with the stores
// skipped, 1268 with no epilogue.  Workgroup b keeps XCD b%8 and takes tiles start_x + (b>>3) + j*(blocks per XCD) of that
// XCD's contiguous range, so the tiles an XCD runs concurrently stay neighbours (same A panels in its L2).
// The epilogue uses only the 64-KiB stage that was just consumed (8 KiB per wave); GELU's two outputs go out in two passes.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_v5_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem5[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntm = (p.M + BM3 - 1) / BM3, ntn = (p.N + BN3 - 1) / BN3, nt = ntm * ntn;
  const int nk = p.K / BK;
  // this workgroup's tiles
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int q = nt >> 3, r = nt & 7;
  const int xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int xn = xcd < r ? q + 1 : q;
  const int my_tiles = lb < xn ? (xn - lb + bpx - 1) / bpx : 0;
  if (my_tiles == 0) return;
  const long S = (long)my_tiles * nk;

  f32x16 acc[2][2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.f;

  // staging cursor (one step ahead of the compute cursor)
  int sj = 0, skt = 0, sm0, sn0;
  {
    int tm, tn;
    tile_of(xs + lb, ntm, ntn, p.order, tm, tn);
    sm0 = tm * BM3; sn0 = tn * BN3;
  }
  int cm0 = sm0, cn0 = sn0, ckt = 0;
  auto stage = [&](long g) {
    char* b_ = smem5 + (g & 1) * STAGE3_BYTES;
    stage2_dma(p.A, p.lda, sm0, p.M, skt * BK, b_, tid, 4);
    stage2_dma(p.W, p.ldw, sn0, p.N, skt * BK, b_ + T3_BYTES, tid, 4);
    if (++skt == nk) {
      skt = 0;
      ++sj;
      if (sj < my_tiles) {
        int tm, tn;
        tile_of(xs + lb + sj * bpx, ntm, ntn, p.order, tm, tn);
        sm0 = tm * BM3; sn0 = tn * BN3;
      }
    }
  };
  stage(0);
  int cj = 0;
  for (long g = 0; g < S; ++g) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // stage g landed (issued a whole k-step ago); older epilogue stores done
    __builtin_amdgcn_s_barrier();
    if (g + 1 < S) stage(g + 1);
    const char* As = smem5 + (g & 1) * STAGE3_BYTES;
    const char* Ws = As + T3_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int s = kk * 2 + (lane >> 5);
      bf16x8 af[4], wf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ar = wm * 128 + i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(As + ar * 128 + ((s ^ ((ar >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int wr = wn * 64 + i * 32 + (lane & 31);
        wf[i] = *(const bf16x8*)(Ws + wr * 128 + ((s ^ ((wr >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[h][ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[h * 2 + mi], acc[h][ni][mi], 0, 0, 0);
    }
    if (++ckt == nk) {      // tile finished: epilogue out of the stage that was just consumed; stage g+1 keeps landing meanwhile
      ckt = 0;
      __builtin_amdgcn_s_barrier();                      // every wave is done reading stage g
      char* wt = smem5 + (g & 1) * STAGE3_BYTES + wave * 8192;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if constexpr (EPI == EPI_GELU) {
          if (p.C2) epilogue_bf16<EPI_GELU, false, 1>(p, acc[h], wt, cm0 + wm * 128 + h * 64, cn0 + wn * 64, lane);
          epilogue_bf16<EPI_GELU, false, 2>(p, acc[h], wt, cm0 + wm * 128 + h * 64, cn0 + wn * 64, lane);
        } else {
          epilogue_bf16<EPI>(p, acc[h], wt, cm0 + wm * 128 + h * 64, cn0 + wn * 64, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.f;
      }
      ++cj;
      if (cj < my_tiles) {
        int tm, tn;
        tile_of(xs + lb + cj * bpx, ntm, ntn, p.order, tm, tn);
        cm0 = tm * BM3; cn0 = tn * BN3;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// v6: v5 + WAVE SPECIALISATION.  vmcnt is an in-order per-wave counter that also counts stores, so in v5 every wave had to
// see its own epilogue stores (a chip-wide 32-MiB burst ~ 8 us at HBM write speed) complete before it could consume the next
// LDS-DMA stage: the stores never overlapped the next tile.  Here waves 4-7 ("loaders") issue ALL LDS-DMA and the R / G
// fix-up loads and never store; every wave converts its accumulators into its 8-KiB LDS region; waves 0-3 ("storers") issue
// ALL global stores (two regions each) and contain no VMEM load at all (the bias vector is staged into LDS once per kernel),
// so nothing in a storer ever waits on vmcnt and the stores drain under the next tile's MFMAs.
constexpr int BIAS6_BYTES = 16384;                  // bias staged in LDS: N <= 4096
constexpr int LDS6_BYTES = LDS3_BYTES + BIAS6_BYTES;

// loader waves: tile[row][*] (+)= R, *= gelu'(G), in place in LDS (16 B per lane, row-coalesced global loads)
template <int EPI>
__device__ __forceinline__ void epi_fixup(const GemmP& p, char* t0, int m_base, int n_base, int lane) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), c16 = lane & 7;
    const int m = m_base + row, n = n_base + c16 * 8;
    if (m >= p.M || n >= p.N) continue;
    const int off = row * 128 + ((c16 ^ (row & 7)) << 4);
    const bool full = n + 8 <= p.N;
    bf16x8 o = *(const bf16x8*)(t0 + off);
    if (p.R) {
      bf16x8 r = {};
      if (full) r = *(const bf16x8*)(p.R + (long)m * p.ldr + n);
      else { const bf16x4 h4 = *(const bf16x4*)(p.R + (long)m * p.ldr + n); r[0] = h4[0]; r[1] = h4[1]; r[2] = h4[2]; r[3] = h4[3]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)o[e] + (float)r[e]);
    }
    if constexpr (EPI == EPI_GELU_GRAD) {
      bf16x8 x = {};
      if (full) x = *(const bf16x8*)(p.G + (long)m * p.ldg + n);
      else { const bf16x4 h4 = *(const bf16x4*)(p.G + (long)m * p.ldg + n); x[0] = h4[0]; x[1] = h4[1]; x[2] = h4[2]; x[3] = h4[3]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)o[e] * gelu_erf_grad((float)x[e]));
    }
    *(bf16x8*)(t0 + off) = o;
  }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_v6_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem6[];
  float* bias_lds = (float*)(smem6 + LDS3_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool loader = wave >= 4;
  const int ntm = (p.M + BM3 - 1) / BM3, ntn = (p.N + BN3 - 1) / BN3, nt = ntm * ntn;
  const int nk = p.K / BK;
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int q = nt >> 3, r = nt & 7;
  const int xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int xn = xcd < r ? q + 1 : q;
  const int my_tiles = lb < xn ? (xn - lb + bpx - 1) / bpx : 0;
  if (my_tiles == 0) return;
  const long S = (long)my_tiles * nk;
  if (p.bias) {
    for (int i = tid; i < p.N; i += 512) bias_lds[i] = p.bias[i];
  }
  GemmP pe = p;                                   // epilogue view: R / G are applied by the loaders' fix-up, not by epi_store
  pe.R = nullptr;

  f32x16 acc[2][2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.f;

  int sj = 0, skt = 0, sm0, sn0;
  {
    int tm, tn;
    tile_of(xs + lb, ntm, ntn, p.order, tm, tn);
    sm0 = tm * BM3; sn0 = tn * BN3;
  }
  int cm0 = sm0, cn0 = sn0, ckt = 0, cj = 0;
  // loaders only: 2048 chunks per operand tile / 256 loader threads = 8 LDS-DMA instructions per operand per stage
  auto stage = [&](long g) {
    char* b_ = smem6 + (g & 1) * STAGE3_BYTES;
    const int lt = tid - 256;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bf16* src = op ? p.W : p.A;
      const long ld = op ? p.ldw : p.lda;
      const int row0 = op ? sn0 : sm0, nrows = op ? p.N : p.M;
      char* tile = b_ + op * T3_BYTES;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int id = c * 256 + lt;
        const int row = id >> 3, ps = id & 7;
        const int ls = ps ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < nrows ? grow : nrows - 1;
        const bf16* gp = src + (long)grow * ld + skt * BK + ls * 8;
        const int wave_base = __builtin_amdgcn_readfirstlane((c * 256 + (lt & ~63)) * 16);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gp, (LDS_AS void*)(tile + wave_base), 16, 0, 0);
      }
    }
  };
  auto advance_stage = [&]() {
    if (++skt == nk) {
      skt = 0;
      ++sj;
      if (sj < my_tiles) {
        int tm, tn;
        tile_of(xs + lb + sj * bpx, ntm, ntn, p.order, tm, tn);
        sm0 = tm * BM3; sn0 = tn * BN3;
      }
    }
  };
  if (loader) stage(0);
  advance_stage();
  for (long g = 0; g < S; ++g) {
    if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the loaders' DMA of stage g (and fix-up loads) landed
    __builtin_amdgcn_s_barrier();
    if (g + 1 < S) {
      if (loader) stage(g + 1);
      advance_stage();
    }
    const char* As = smem6 + (g & 1) * STAGE3_BYTES;
    const char* Ws = As + T3_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int s = kk * 2 + (lane >> 5);
      bf16x8 af[4], wf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ar = wm * 128 + i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(As + ar * 128 + ((s ^ ((ar >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int wr = wn * 64 + i * 32 + (lane & 31);
        wf[i] = *(const bf16x8*)(Ws + wr * 128 + ((s ^ ((wr >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[h][ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[h * 2 + mi], acc[h][ni][mi], 0, 0, 0);
    }
    if (++ckt == nk) {
      ckt = 0;
      char* base = smem6 + (g & 1) * STAGE3_BYTES;         // the stage just consumed: 8 regions of 8 KiB
      char* mine = base + wave * 8192;
      constexpr bool FIX = (EPI == EPI_GELU_GRAD);
      const bool fix = FIX || p.R != nullptr;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        constexpr int NPASS = (EPI == EPI_GELU) ? 2 : 1;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          if (EPI == EPI_GELU && pass == 0 && !p.C2) continue;
          __builtin_amdgcn_s_barrier();                    // regions free: main-loop reads / previous stores' LDS reads done
          if constexpr (EPI == EPI_GELU) {
            if (pass == 0) epi_convert<EPI_GELU, 1>(p, acc[h], mine, mine, cn0 + wn * 64, lane, bias_lds);
            else epi_convert<EPI_GELU, 2>(p, acc[h], mine, mine, cn0 + wn * 64, lane, bias_lds);
          } else {
            epi_convert<EPI, 0>(p, acc[h], mine, mine, cn0 + wn * 64, lane, bias_lds);
          }
          __builtin_amdgcn_s_barrier();
          if (fix) {                                        // loaders patch R / gelu'(G) into two regions each
            if (loader) {
#pragma unroll
              for (int rr = 0; rr < 2; ++rr) {
                const int w2 = (wave - 4) + rr * 4;
                epi_fixup<EPI>(p, base + w2 * 8192, cm0 + (w2 >> 2) * 128 + h * 64, cn0 + (w2 & 3) * 64, lane);
              }
            }
            __builtin_amdgcn_s_barrier();
          }
          if (!loader) {                                    // storers: two regions each, no VMEM loads in this path
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int w2 = wave + rr * 4;
              const char* reg = base + w2 * 8192;
              const int mb = cm0 + (w2 >> 2) * 128 + h * 64, nb = cn0 + (w2 & 3) * 64;
              if constexpr (EPI == EPI_GELU) {
                if (pass == 0) epi_store<EPI_GELU, false, 1>(pe, reg, reg, mb, nb, lane);
                else epi_store<EPI_GELU, false, 2>(pe, reg, reg, mb, nb, lane);
              } else if constexpr (EPI == EPI_GELU_GRAD) {
                epi_store<EPI_BF16, false, 0>(pe, reg, reg, mb, nb, lane);      // gelu' already applied by the fix-up
              } else {
                epi_store<EPI, false, 0>(pe, reg, reg, mb, nb, lane);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.f;
      }
      ++cj;
      if (cj < my_tiles) {
        int tm, tn;
        tile_of(xs + lb + cj * bpx, ntm, ntn, p.order, tm, tn);
        cm0 = tm * BM3; cn0 = tn * BN3;
      }
    }
  }
}

template <int EPI>
int launch_v6_one(const GemmP& p, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_v6_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS6_BYTES);
    if (e != hipSuccess) {
      spmm_set_error("spmm_gemm_nt: cannot raise dynamic LDS to %d: %s", LDS6_BYTES, hipGetErrorString(e));
      return SPMM_ERR_LAUNCH;
    }
    attr = true;
  }
  const long nt = (long)((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3);
  int nb = nt >= 256 ? 256 : (int)((nt + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_nt_v6_kernel<EPI>), dim3(nb), dim3(512), LDS6_BYTES, st, p);
  return SPMM_OK;
}

template <int EPI>
int launch_v5_one(const GemmP& p, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_v5_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_BYTES);
    if (e != hipSuccess) {
      spmm_set_error("spmm_gemm_nt: cannot raise dynamic LDS to %d: %s", LDS3_BYTES, hipGetErrorString(e));
      return SPMM_ERR_LAUNCH;
    }
    attr = true;
  }
  const long nt = (long)((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3);
  int nb = nt >= 256 ? 256 : (int)((nt + 7) / 8) * 8;     // one workgroup per CU, a multiple of the 8 XCDs
  hipLaunchKernelGGL((gemm_nt_v5_kernel<EPI>), dim3(nb), dim3(512), LDS3_BYTES, st, p);
  return SPMM_OK;
}

// ------------------------------------------------------------------------------------------------------------
// v4: the 256x256 tile with a FOUR-slot ring of BK=32 stages (4 x 32 KiB = 128 KiB), three stages always in flight.
// Model behind it (PMC + aliasing experiments): the loop is DMA-latency bound, throughput ~ bytes in flight / latency;
// v3 keeps 64 KiB in flight per CU (=> ~25 GB/s/CU at the ~2.5 us it takes a stage to land, 39 % MFMA busy), this ring
// keeps 96 KiB in flight.  LDS rows are 64 B here, so the 16-B slot index is XOR-ed with (row>>2)&3.
constexpr int BK4 = 32;
constexpr int T4_BYTES = 256 * BK4 * 2;             // 16 KiB per operand tile
constexpr int STAGE4_BYTES = 2 * T4_BYTES;          // 32 KiB
constexpr int LDS4_BYTES = 4 * STAGE4_BYTES;        // 128 KiB

__device__ __forceinline__ void dma4(const bf16* __restrict__ src, long ld, int row0, int nrows, int k0, char* lds_tile, int tid) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = c * 512 + tid;
    const int row = id >> 2, ps = id & 3;
    const int ls = ps ^ ((row >> 2) & 3);
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;
    const bf16* g = src + (long)grow * ld + k0 + ls * 8;
    const int wave_base = __builtin_amdgcn_readfirstlane((c * 512 + (tid & ~63)) * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lds_tile + wave_base), 16, 0, 0);
  }
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_v4_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntm = (p.M + BM3 - 1) / BM3, ntn = (p.N + BN3 - 1) / BN3, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  int tile_m, tile_n;
  tile_of(t, ntm, ntn, p.order, tile_m, tile_n);
  const int m0 = tile_m * BM3, n0 = tile_n * BN3;
  const int nk = p.K / BK4;

  f32x16 acc[2][2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

#define STAGE4(kt_)                                                                       \
  do {                                                                                    \
    char* b_ = smem4 + ((kt_) & 3) * STAGE4_BYTES;                                        \
    dma4(p.A, p.lda, m0, p.M, (kt_) * BK4, b_, tid);                                      \
    dma4(p.W, p.ldw, n0, p.N, (kt_) * BK4, b_ + T4_BYTES, tid);                           \
  } while (0)

  STAGE4(0);
  if (nk > 1) STAGE4(1);
  if (nk > 2) STAGE4(2);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; the 4 DMA instructions of each younger stage may still be in flight
    const int younger = min(nk - 1 - kt, 2);
    if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // also: everyone is done reading slot (kt-1)&3 == (kt+3)&3
    if (kt + 3 < nk) STAGE4(kt + 3);
    const char* As = smem4 + (kt & 3) * STAGE4_BYTES;
    const char* Ws = As + T4_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int s = kk * 2 + (lane >> 5);
      bf16x8 af[4], wf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ar = wm * 128 + i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(As + ar * 64 + ((s ^ ((ar >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int wr = wn * 64 + i * 32 + (lane & 31);
        wf[i] = *(const bf16x8*)(Ws + wr * 64 + ((s ^ ((wr >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[h][ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[h * 2 + mi], acc[h][ni][mi], 0, 0, 0);
    }
  }
#undef STAGE4
  __builtin_amdgcn_s_barrier();
  char* wt = smem4 + wave * 16384;
#pragma unroll
  for (int h = 0; h < 2; ++h)
    epilogue_bf16<EPI>(p, acc[h], wt, m0 + wm * 128 + h * 64, n0 + wn * 64, lane);
}

template <int EPI>
int launch_v4_one(const GemmP& p, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_v4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4_BYTES);
    if (e != hipSuccess) {
      spmm_set_error("spmm_gemm_nt: cannot raise dynamic LDS to %d: %s", LDS4_BYTES, hipGetErrorString(e));
      return SPMM_ERR_LAUNCH;
    }
    attr = true;
  }
  dim3 grid(((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3));
  hipLaunchKernelGGL((gemm_nt_v4_kernel<EPI>), grid, dim3(512), LDS4_BYTES, st, p);
  return SPMM_OK;
}

template <int EPI, bool IL>
int launch_v3_il(const GemmP& p, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_v3_kernel<EPI, IL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_BYTES);
    if (e != hipSuccess) {
      spmm_set_error("spmm_gemm_nt: cannot raise dynamic LDS to %d: %s", LDS3_BYTES, hipGetErrorString(e));
      return SPMM_ERR_LAUNCH;
    }
    attr = true;
  }
  dim3 grid(((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3));
  hipLaunchKernelGGL((gemm_nt_v3_kernel<EPI, IL>), grid, dim3(512), LDS3_BYTES, st, p);
  return SPMM_OK;
}
static int g_krot = 0;
static int g_stagger = 0;
static int g_force_tile = 0;          // 0: heuristic, 1/2/3: force the 128x128 / 256x128 / 256x256 kernel (tuning sweeps)
static int g_tile_order = 3;          // row-major, split into 2 column groups when N >= 2048 and K <= 1024: lowest measured L2-miss traffic (profiles/r01_pmc_nt_gemm.txt); launch time is order-insensitive
static int g_v3_abl = 0;           // timing ablations of v3 (0 = none)
static int g_v3_interleave = 0;   // interleaving the DMA issue with the MFMA groups measured equal / slightly worse
template <int ABL>
int launch_v3_abl(const GemmP& p, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)gemm_nt_v3_kernel<EPI_BF16, false, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_BYTES);
  dim3 grid(((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3));
  hipLaunchKernelGGL((gemm_nt_v3_kernel<EPI_BF16, false, ABL>), grid, dim3(512), LDS3_BYTES, st, p);
  return SPMM_OK;
}
template <int EPI>
int launch_v3_one(const GemmP& p, hipStream_t st) {
  if (EPI == EPI_BF16 && g_v3_abl == 1) return launch_v3_abl<1>(p, st);
  if (EPI == EPI_BF16 && g_v3_abl == 2) return launch_v3_abl<2>(p, st);
  if (EPI == EPI_BF16 && g_v3_abl == 3) return launch_v3_abl<3>(p, st);
  if (EPI == EPI_BF16 && g_v3_abl == 4) return launch_v3_abl<4>(p, st);
  if (EPI == EPI_BF16 && g_v3_abl == 5) return launch_v3_abl<5>(p, st);
  return g_v3_interleave ? launch_v3_il<EPI, true>(p, st) : launch_v3_il<EPI, false>(p, st);
}
static int g_use_v7 = 0;           // 1: four-wave 128x128-wave-tile variant of the 256x256 kernel
template <int EPI>
int launch_v7_one(const GemmP& p, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_v7_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS3_BYTES);
    if (e != hipSuccess) {
      spmm_set_error("spmm_gemm_nt: cannot raise dynamic LDS to %d: %s", LDS3_BYTES, hipGetErrorString(e));
      return SPMM_ERR_LAUNCH;
    }
    attr = true;
  }
  dim3 grid(((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3));
  hipLaunchKernelGGL((gemm_nt_v7_kernel<EPI>), grid, dim3(256), LDS3_BYTES, st, p);
  return SPMM_OK;
}
static int g_use_v4 = 0;
static int g_use_v5 = 0;           // 0 (default): per-tile v3; 1: persistent v5; 2: persistent + wave-specialised v6 -- all three measure the same
int launch_v3(int epi, const GemmP& p, hipStream_t st) {
  if (g_use_v7 && g_v3_abl == 0) {
    switch (epi) {
      case EPI_BF16: return launch_v7_one<EPI_BF16>(p, st);
      case EPI_GELU: return launch_v7_one<EPI_GELU>(p, st);
      case EPI_GELU_GRAD: return launch_v7_one<EPI_GELU_GRAD>(p, st);
      default: return -1;
    }
  }
  if (g_use_v5 == 2 && !g_use_v4 && g_v3_abl == 0 && p.N <= 4096) {
    switch (epi) {
      case EPI_BF16: return launch_v6_one<EPI_BF16>(p, st);
      case EPI_GELU: return launch_v6_one<EPI_GELU>(p, st);
      case EPI_GELU_GRAD: return launch_v6_one<EPI_GELU_GRAD>(p, st);
      default: return -1;
    }
  }
  if (g_use_v5 && !g_use_v4 && g_v3_abl == 0) {
    switch (epi) {
      case EPI_BF16: return launch_v5_one<EPI_BF16>(p, st);
      case EPI_GELU: return launch_v5_one<EPI_GELU>(p, st);
      case EPI_GELU_GRAD: return launch_v5_one<EPI_GELU_GRAD>(p, st);
      default: return -1;
    }
  }
  if (g_use_v4) {
    switch (epi) {
      case EPI_BF16: return launch_v4_one<EPI_BF16>(p, st);
      case EPI_GELU: return launch_v4_one<EPI_GELU>(p, st);
      case EPI_GELU_GRAD: return launch_v4_one<EPI_GELU_GRAD>(p, st);
      default: return -1;
    }
  }
  switch (epi) {
    case EPI_BF16: return launch_v3_one<EPI_BF16>(p, st);
    case EPI_GELU: return launch_v3_one<EPI_GELU>(p, st);
    case EPI_GELU_GRAD: return launch_v3_one<EPI_GELU_GRAD>(p, st);
    default: return -1;
  }
}

static int g_v2_variant = 0;
template <int EPI, int VAR>
int launch_v2_var(const GemmP& p, dim3 grid, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_v2_kernel<EPI, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    if (e != hipSuccess) {
      spmm_set_error("spmm_gemm_nt: cannot raise dynamic LDS to %d: %s", LDS2_BYTES, hipGetErrorString(e));
      return SPMM_ERR_LAUNCH;
    }
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt_v2_kernel<EPI, VAR>), grid, dim3(512), LDS2_BYTES, st, p);
  return SPMM_OK;
}
template <int EPI>
int launch_v2_one(const GemmP& p, dim3 grid, hipStream_t st) {
  switch (g_v2_variant) {
    case 1: return launch_v2_var<EPI, 1>(p, grid, st);
    case 2: return launch_v2_var<EPI, 2>(p, grid, st);
    case 3: return launch_v2_var<EPI, 3>(p, grid, st);
    case 4: return launch_v2_var<EPI, 4>(p, grid, st);
    case 5: return launch_v2_var<EPI, 5>(p, grid, st);
    case 7: return launch_v2_var<EPI, 7>(p, grid, st);
    default: return launch_v2_var<EPI, 0>(p, grid, st);
  }
}
int launch_v2(int epi, const GemmP& p, hipStream_t st) {
  dim3 grid(((p.M + BM2 - 1) / BM2) * ((p.N + BN - 1) / BN));
  switch (epi) {
    case EPI_BF16: return launch_v2_one<EPI_BF16>(p, grid, st);
    case EPI_GELU: return launch_v2_one<EPI_GELU>(p, grid, st);
    case EPI_F32: return launch_v2_one<EPI_F32>(p, grid, st);
    case EPI_GELU_GRAD: return launch_v2_one<EPI_GELU_GRAD>(p, grid, st);
    case EPI_F32_ACC: return launch_v2_one<EPI_F32_ACC>(p, grid, st);
    default: return -1;
  }
}

template <bool GLDS>
int launch(int epi, const GemmP& p, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: hipLaunchKernelGGL((gemm_nt_kernel<EPI_BF16, GLDS>), grid, dim3(256), 0, st, p); break;
    case EPI_GELU: hipLaunchKernelGGL((gemm_nt_kernel<EPI_GELU, GLDS>), grid, dim3(256), 0, st, p); break;
    case EPI_F32: hipLaunchKernelGGL((gemm_nt_kernel<EPI_F32, GLDS>), grid, dim3(256), 0, st, p); break;
    case EPI_F32_ATOMIC: hipLaunchKernelGGL((gemm_nt_kernel<EPI_F32_ATOMIC, GLDS>), grid, dim3(256), 0, st, p); break;
    case EPI_GELU_GRAD: hipLaunchKernelGGL((gemm_nt_kernel<EPI_GELU_GRAD, GLDS>), grid, dim3(256), 0, st, p); break;
    case EPI_F32_ACC: hipLaunchKernelGGL((gemm_nt_kernel<EPI_F32_ACC, GLDS>), grid, dim3(256), 0, st, p); break;
    default: spmm_set_error("spmm_gemm_nt: unknown epilogue %d", epi); return SPMM_ERR_UNSUPPORTED;
  }
  return SPMM_OK;
}

}  // namespace

static int g_gemm_use_glds = 1;    // 1: LDS-DMA staging, 0: register staging (v1 only), 2: force the v1 kernel with LDS-DMA
extern "C" void spmm_gemm_set_staging(int use_lds_dma) { g_gemm_use_glds = use_lds_dma; }
extern "C" void spmm_gemm_set_variant(int v) {
  if (v >= 300 && v <= 304) { g_tile_order = v - 300; return; }
  if (v == 400 || v == 401) { g_use_v4 = v - 400; return; }
  if (v >= 500 && v <= 505) { g_v3_abl = v - 500; return; }
  if (v == 700 || v == 701) { g_krot = v - 700; return; }
  if (v >= 800 && v <= 803) { g_force_tile = v - 800; return; }
  if (v >= 900 && v <= 915) { g_stagger = v - 900; return; }
  if (v == 1000 || v == 1001) { g_use_v7 = v - 1000; return; }
  if (v >= 600 && v <= 602) { g_use_v5 = v - 600; return; }   // 600: v3 per-tile launch, 601: persistent v5, 602: wave-specialised v6
  if (v == 200) g_v3_interleave = 0;
  else if (v == 201) g_v3_interleave = 1;
  else g_v2_variant = v;
}

// Which tile kernel for an M x N output?  Per-tile efficiency (bytes through the per-CU load path per FLOP) favours the big
// tile, wave quantisation on 256 CUs favours the small one: score = efficiency x fill of the last wave.  The 256x256 and
// 256x128 kernels hold one workgroup per CU (128 / 144 KiB LDS); the 128x128 kernel (32 KiB) runs several per CU, so its fill
// is smooth.  Constants from tools/bench_gemm_tiles.py.  Used for the decoder's shapes (M = beams x molecules < 6000), e.g.
// 5000x768x3072: 640 TF with 128x128 vs 472 with 256x128 vs 292 with 256x256.
static int pick_tile(int M, int N, int epi) {
  const bool bf = epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_GELU_GRAD;
  if (M >= 6000) {
    // Training-step shapes keep the simple rule (256x256 whenever it gives >= 96 tiles): inside the step the small-M GEMMs run
    // next to another stream's kernels, which fill the CUs a coarse tiling leaves idle, and the per-tile efficiency of the big
    // tile wins -- the score below made the whole step 4 % slower (1 497 vs 1 560 molecules/s) although it wins in isolation.
    const long t3 = (long)((M + 255) / 256) * ((N + 255) / 256);
    if (bf && t3 >= 96) return 3;
    return (epi != EPI_F32_ATOMIC) ? 2 : 1;
  }
  auto tiles = [&](int bm, int bn) { return (double)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  auto fill = [](double t) { const double w = (double)(long)((t + 255) / 256); return t / (256.0 * w); };
  const double t3 = tiles(256, 256), t2 = tiles(256, 128), t1 = tiles(128, 128);
  const double s3 = bf ? 1.00 * fill(t3) : 0.0;
  const double s2 = (M >= 512 && epi != EPI_F32_ATOMIC) ? 0.85 * fill(t2) : 0.0;
  const double s1 = 0.80 * (t1 < 256 ? t1 / 256.0 : 1.0);
  if (s3 >= s2 && s3 >= s1) return 3;
  return s2 >= s1 ? 2 : 1;
}

extern "C" int spmm_gemm_nt(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int splits,
                            const float* bias, const float* div_ptr, float alpha, const void* R, long ldr,
                            const void* G, long ldg, void* C, long ldc, void* C2, long ldc2, int epi, float* colsum,
                            hipStream_t stream) {
  SPMM_CHECK_SHAPE(M > 0 && N > 0 && K > 0, "spmm_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  SPMM_CHECK_SHAPE(K % 64 == 0, "spmm_gemm_nt: K=%d must be a multiple of 64", K);
  SPMM_CHECK_SHAPE(N % 4 == 0, "spmm_gemm_nt: N=%d must be a multiple of 4", N);
  SPMM_CHECK_SHAPE(lda % 8 == 0 && ldw % 8 == 0, "spmm_gemm_nt: lda/ldw must be multiples of 8 elements");
  SPMM_CHECK_SHAPE(ldc % 4 == 0, "spmm_gemm_nt: ldc must be a multiple of 4");
  SPMM_CHECK_SHAPE(epi == EPI_F32 || epi == EPI_F32_ATOMIC || epi == EPI_F32_ACC || (ldc % 8 == 0 && (!R || ldr % 8 == 0) && (!G || ldg % 8 == 0) && (!C2 || ldc2 % 8 == 0) && (uintptr_t)C % 16 == 0),
                   "spmm_gemm_nt: bf16 outputs need 16-B aligned rows (ldc/ldr/ldg/ldc2 multiples of 8)");
  SPMM_CHECK_SHAPE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), "spmm_gemm_nt: A/W must be 16-B aligned");
  if (splits < 1) splits = 1;
  SPMM_CHECK_SHAPE(splits == 1 || epi == EPI_F32_ATOMIC, "spmm_gemm_nt: split-K needs the atomic epilogue");
  SPMM_CHECK_SHAPE(epi != EPI_GELU_GRAD || G != nullptr, "spmm_gemm_nt: GELU-grad epilogue needs G");
  int ksplit = ((K / 64 + splits - 1) / splits) * 64;
  splits = (K + ksplit - 1) / ksplit;
  GemmP p;
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw;
  p.M = M; p.N = N; p.K = K; p.ksplit = ksplit; p.bias = bias; p.div_ptr = div_ptr; p.alpha = alpha;
  p.R = (const bf16*)R; p.ldr = ldr; p.G = (const bf16*)G; p.ldg = ldg; p.C = C; p.ldc = ldc;
  p.C2 = (bf16*)C2; p.ldc2 = ldc2; p.order = (g_tile_order >= 3 && K > 1024) ? 0 : g_tile_order; p.colsum = colsum; p.krot = g_krot; p.stagger = g_stagger;
  SPMM_CHECK_SHAPE(colsum == nullptr || epi == EPI_BF16 || epi == EPI_GELU_GRAD, "spmm_gemm_nt: colsum is only fused into the bf16 / GELU-grad epilogues");
  if (g_gemm_use_glds == 1 && splits == 1 && g_v2_variant != 100) {   // v3: 256x256 tile when it still fills the chip
    const bool want3 = g_force_tile ? g_force_tile == 3 : (pick_tile(M, N, epi) == 3 || g_v2_variant == 101);
    if (want3 && (epi == EPI_BF16 || epi == EPI_GELU || epi == EPI_GELU_GRAD)) {
      int rc3 = launch_v3(epi, p, stream);
      if (rc3 > 0) return rc3;
      if (rc3 == 0) {
        SPMM_LAUNCH_CHECK("spmm_gemm_nt(v3)");
        return SPMM_OK;
      }
    }
  }
  if (g_gemm_use_glds == 1 && splits == 1 && epi != EPI_F32_ATOMIC && M >= 512 && (g_force_tile ? g_force_tile != 1 : pick_tile(M, N, epi) != 1)) {   // v2: 256x128 tile, 3-stage ring
    int rc2 = launch_v2(epi, p, stream);
    if (rc2 > 0) return rc2;
    if (rc2 == 0) {
      SPMM_LAUNCH_CHECK("spmm_gemm_nt(v2)");
      return SPMM_OK;
    }
  }
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  dim3 grid(ntm * ntn, 1, splits);
  int rc = g_gemm_use_glds ? launch<true>(epi, p, grid, stream) : launch<false>(epi, p, grid, stream);
  if (rc != SPMM_OK) return rc;
  SPMM_LAUNCH_CHECK("spmm_gemm_nt");
  return SPMM_OK;
}
