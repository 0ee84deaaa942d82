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

enum Epi { EPI_BF16 = 0, EPI_GELU = 1, EPI_F32 = 2, EPI_F32_ATOMIC = 3, EPI_GELU_GRAD = 4, EPI_F32_ACC = 5, EPI_GELU_DERIV = 6, EPI_MUL = 7 };
// bf16-output epilogues (LDS-transposed, row-coalesced stores): BF16, GELU (C2 = pre-activation), GELU_DERIV (C2 = gelu'(pre)),
// GELU_GRAD (C = acc * gelu'(G), G = pre-activation), MUL (C = acc * G, G = a stored derivative)
constexpr bool epi_is_bf16(int e) { return e == EPI_BF16 || e == EPI_GELU || e == EPI_GELU_GRAD || e == EPI_GELU_DERIV || e == EPI_MUL; }

struct GemmP {
  const bf16* A; long lda;
  const bf16* W; long ldw;
  int M, N, K;               // this launch reduces K elements per split-slice; K % 64 == 0
  int ksplit;                // elements of K per grid.z slice (multiple of 64)
  const float* bias;         // [N] or null
  const float* div_ptr;      // optional device scalar: acc /= *div_ptr
  float alpha;               // acc *= alpha
  const bf16* R; long ldr;   // optional bf16 addend (residual-gradient accumulate)
  const bf16* G; long ldg;   // EPI_GELU_GRAD: pre-activation; EPI_MUL: the factor
  void* C; long ldc;
  bf16* C2; long ldc2;       // EPI_GELU: pre-activation output; EPI_GELU_DERIV: gelu'(pre-activation)
  float* colsum;             // optional: colsum[n] += sum_m C[m][n] of the (bf16-rounded) output -- bias gradient of the producing layer
  int order;                 // tile order inside an XCD's range (tile_of): 0 row-major, 3 row-major inside 2 column groups
  // optional device-side row count (8-phase kernel only): the launch is SIZED for M rows, the kernel computes *M_ptr <= M of them -- the
  // rows of a batch whose tail length only the device knows (step.py: the hard negatives drawn as text queries)
  const int* M_ptr;
};

// device-side row count (GemmP::M_ptr): the kernel argument copy of M is lowered once, before anything derives from it
#define GEMM_DYN_M(P)                                          \
  do {                                                         \
    if ((P).M_ptr) {                                           \
      const int m_ = *(P).M_ptr;                               \
      (P).M = m_ < (P).M ? (m_ > 0 ? m_ : 1) : (P).M;          \
    }                                                          \
  } while (0)

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

// Coalesced bf16 epilogue: a wave's 64x64 output tile goes through LDS (8 KiB per wave per output) so that global
// stores are 16 B per lane along rows (8 lanes = one 128-B row segment) instead of 8-B fragments scattered over 32 rows.
// acc[ni][mi][r] = D[n][m], m = lane&31, n = (r&3) + 8*(r>>2) + 4*(lane>>5).   Tile rows are 128 B; the 16-B chunk index is
// XOR-ed with (row & 7) so the row-parallel writes and the row-major reads both spread over the banks.
//   epi_convert: accumulators -> scale / bias / GELU -> bf16 tile in LDS
//   epi_store  : LDS tile -> (+R, * gelu'(G)) -> row-coalesced global stores (+ fused column sums)
template <int EPI>
__device__ __forceinline__ void epi_convert(const GemmP& p, f32x16 (&acc)[2][2], char* t0, char* t1, int n_base, int lane) {
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
            const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += b[j];
          }
        }
        const int off = row * 128 + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 4) << 1));
        if constexpr (EPI == EPI_GELU) {
          if (p.C2) *(bf16x4*)(t1 + off) = to_bf16x4(v[0], v[1], v[2], v[3]);
          *(bf16x4*)(t0 + off) = to_bf16x4(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
        } else if constexpr (EPI == EPI_GELU_DERIV) {
          float gv[4], dv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) gelu_erf_both(v[j], gv[j], dv[j]);
          if (p.C2) *(bf16x4*)(t1 + off) = to_bf16x4(dv[0], dv[1], dv[2], dv[3]);
          *(bf16x4*)(t0 + off) = to_bf16x4(gv[0], gv[1], gv[2], gv[3]);
        } else {
          *(bf16x4*)(t0 + off) = to_bf16x4(v[0], v[1], v[2], v[3]);
        }
      }
    }
}

// cs_carry (optional): column-sum partials kept in registers across consecutive calls on the SAME columns (the two m-halves of
// a wave tile); they are flushed with atomics only when `flush` is set -- half the same-address atomics per tile.
template <int EPI>
__device__ __forceinline__ void epi_store(const GemmP& p, const char* t0, const char* t1, int m_base, int n_base, int lane,
                                          float* cs_carry = nullptr, bool flush = true) {
  // all R / G operand loads first: a load placed behind a store to a possibly aliasing pointer would be serialised behind it
  bf16x8 rr[8], gg[8];
  if (p.R || EPI == EPI_GELU_GRAD || EPI == EPI_MUL) {
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
      if constexpr (EPI == EPI_GELU_GRAD || EPI == EPI_MUL) {
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
    if constexpr (EPI == EPI_MUL) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)o[e] * (float)gg[it][e]);
    }
    if (p.colsum) {
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += (float)o[e];
    }
    bf16* dst = (bf16*)p.C + (long)m * p.ldc + n;
    if (full) {
      *(bf16x8*)dst = o;
      if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_DERIV) { if (p.C2) *(bf16x8*)(p.C2 + (long)m * p.ldc2 + n) = *(const bf16x8*)(t1 + off); }
    } else {   // ragged last chunk (N % 8 == 4)
      bf16x4 lo4; lo4[0] = o[0]; lo4[1] = o[1]; lo4[2] = o[2]; lo4[3] = o[3];
      *(bf16x4*)dst = lo4;
      if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_DERIV) { if (p.C2) *(bf16x4*)(p.C2 + (long)m * p.ldc2 + n) = *(const bf16x4*)(t1 + off); }
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
template <int EPI>
__device__ __forceinline__ void epilogue_bf16(const GemmP& p, f32x16 (&acc)[2][2], char* wtile, int m_base, int n_base, int lane,
                                              float* cs_carry = nullptr, bool flush = true) {
  epi_convert<EPI>(p, acc, wtile, wtile + 8192, n_base, lane);
  epi_store<EPI>(p, wtile, wtile + 8192, m_base, n_base, lane, cs_carry, flush);
}

// The output side of the 128x128 kernels (4 waves as 2 x 2, 64 x 64 per wave): bf16 epilogues through the wave's 16 KiB of LDS, the
// fp32 ones straight from the accumulators.
template <int EPI>
__device__ __forceinline__ void tile128_out(const GemmP& p, f32x16 (&acc)[2][2], char* smem, int wave, int m0, int n0, int wm, int wn, int lane) {
  if constexpr (epi_is_bf16(EPI)) {                   // (the caller's barrier: every wave is done reading the staging buffers)
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

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile id: workgroup b runs on XCD b%8; give each XCD a contiguous range of tiles.
  GEMM_DYN_M(p);
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    if (i >= (xcd < r ? q + 1 : q)) return;          // (a device-side row count left this workgroup without a tile)
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

  if (nk > 0) {
    stage_tile_dma(p.A, p.lda, m0, p.M, kbeg, smem, tid);
    stage_tile_dma(p.W, p.ldw, n0, p.N, kbeg, smem + TILE_BYTES, tid);
  }
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();   // tile kt landed (vmcnt(0) is part of the barrier's release while LDS-DMA is pending)
    char* cur = smem + (kt & 1) * STAGE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
    if (kt + 1 < nk) {
      stage_tile_dma(p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, nxt, tid);
      stage_tile_dma(p.W, p.ldw, n0, p.N, kbeg + (kt + 1) * BK, nxt + TILE_BYTES, tid);
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
  }

  if constexpr (epi_is_bf16(EPI)) __syncthreads();   // every wave is done reading the staging buffers
  tile128_out<EPI>(p, acc, smem, wave, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------------
// pc: the same 128x128x64 tile with the work of a k-step split between two kinds of waves -- 4 COMPUTE waves (the 128x128 kernel's
// wave tiles, MFMAs and epilogues; fragment reads one k-slice ahead of the MFMAs) and 4 LOADER waves that issue every LDS-DMA of a
// four-stage ring and count their own vmcnt.  One wave of each kind per SIMD: a loader's DMA issue (8 instructions per k-step, 60-185
// cycles each) runs beside its partner's 16 MFMAs instead of in front of them, and three K-tiles are in flight instead of one.  (With the
// loads issued by the computing waves themselves the deeper ring alone measured nothing, EXPERIMENTS.md 1.11: the k-step was the wave's own
// instruction stream.)  One raw s_barrier per k-step: barrier(kt) = "K-tile kt has landed (the loaders waited for their pieces) and every
// compute wave is done reading K-tile kt-1 (its fragment reads are consumed by MFMAs issued before the barrier)"; after it the loaders
// restage the stage of K-tile kt-1 with K-tile kt+3.
// Measured (tools/gemm_small_m.py, M = 5 000): 768x768 13.9 -> 12.2 us, 768x3072 36.2 -> 27.3-28.8 us; ~980 cycles per k-step against
// ~1 370 -- and neither eight loader waves, nor a fifth stage, nor double-buffered fragments move it further: four waves of 64x64 tiles read
// 64 KiB of fragments per k-step and the DMA writes another 32 KiB, 768 cycles of the CU's 128-B/clk LDS against 512 of the matrix pipe
// (the 256x256 tile's 128x64 wave tiles read half as much per FLOP).  It holds the CU alone (128 KiB), so it is the automatic choice only
// where the 128x128 tiling gives at most one workgroup per CU; with more tiles several 32-KiB workgroups of the plain kernel share a CU
// and win (5000x2304x768: 33.8 against 38 us).
#ifndef PC_AUTO
#define PC_AUTO 1                  // 0: never chosen automatically
#endif
constexpr int PC_STAGES = 4;                          // (5 stages -- all of the 160 KiB -- measured the same: the k-step is not waiting for its operands)
constexpr int PC_LDS = PC_STAGES * STAGE_BYTES;     // 128 KiB
#ifndef PC_LOADERS
#define PC_LOADERS 4              // loader waves (4 or 8; 8 -- two per SIMD, four DMA instructions per wave and k-step -- measured the same)
#endif
constexpr int PC_LT = PC_LOADERS * 64;               // loader threads
constexpr int PC_NC = 1024 / PC_LT;                  // 16-byte chunks of an operand tile per loader thread
constexpr int PC_THREADS = 256 + PC_LT;

template <int EPI>
__global__ __launch_bounds__(PC_THREADS) void gemm_nt_pc_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_pc[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  GEMM_DYN_M(p);
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    if (i >= (xcd < r ? q + 1 : q)) return;          // (a device-side row count left this workgroup without a tile)
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int tile_m = t / ntn, tile_n = t % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = blockIdx.z * p.ksplit;
  const int kend = min(p.K, kbeg + p.ksplit);
  const int nk = (kend - kbeg) / BK;

  if (wave >= 4) {
    // ---------------- loader waves: chunk id = c * PC_LT + ltid -> tile row id >> 3, physical 16-B slot id & 7 (stage_tile_dma's layout)
    const int ltid = tid - 256;
    uint32_t offA[PC_NC], offW[PC_NC], dst[PC_NC];
#pragma unroll
    for (int c = 0; c < PC_NC; ++c) {
      const int id = c * PC_LT + ltid;
      const int row = id >> 3, ps = id & 7;
      const int ls = ps ^ ((row >> 1) & 7);
      int ga = m0 + row, gw = n0 + row;
      ga = ga < p.M ? ga : p.M - 1;                        // clamped rows are computed and never stored
      gw = gw < p.N ? gw : p.N - 1;
      offA[c] = (uint32_t)ga * (uint32_t)(p.lda * 2) + (uint32_t)(ls * 16);
      offW[c] = (uint32_t)gw * (uint32_t)(p.ldw * 2) + (uint32_t)(ls * 16);
      dst[c] = (uint32_t)((c * PC_LT + (ltid & ~63)) * 16);  // wave-uniform: the wave's 64 chunks are contiguous in LDS
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem_pc;
    const char* gA = (const char*)p.A + (size_t)kbeg * 2;
    const char* gW = (const char*)p.W + (size_t)kbeg * 2;
#define PC_DMA(VOFF, SBASE, LDSDST)                                                                                          \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSDST) : "memory", "m0")
#define PC_STAGE(KT)                                                                                                         \
    do {                                                                                                                     \
      const uint32_t st_ = lds0 + (uint32_t)(((KT) % PC_STAGES) * STAGE_BYTES);                                              \
      const char* a_ = gA + (size_t)(KT) * (BK * 2);                                                                         \
      const char* w_ = gW + (size_t)(KT) * (BK * 2);                                                                         \
      _Pragma("unroll") for (int c = 0; c < PC_NC; ++c) {                                                                    \
        const uint32_t d_ = __builtin_amdgcn_readfirstlane(st_ + dst[c]);                                                    \
        PC_DMA(offA[c], a_, d_);                                                                                             \
      }                                                                                                                      \
      _Pragma("unroll") for (int c = 0; c < PC_NC; ++c) {                                                                    \
        const uint32_t d_ = __builtin_amdgcn_readfirstlane(st_ + (uint32_t)TILE_BYTES + dst[c]);                             \
        PC_DMA(offW[c], w_, d_);                                                                                             \
      }                                                                                                                      \
    } while (0)
    // K-tiles after `kt` still allowed in flight when K-tile kt must have landed: 2 * PC_NC DMA instructions each, issued in order
#define PC_WAIT(NEWER)                                                                                                       \
    do {                                                                                                                     \
      if ((NEWER) >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PC_NC) : "memory");                                     \
      else if ((NEWER) == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PC_NC) : "memory");                                \
      else if ((NEWER) == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PC_NC) : "memory");                                \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                  \
    } while (0)
    constexpr int AHEAD = PC_STAGES - 1;                   // K-tiles issued beyond the one being multiplied
    const int pre = nk < AHEAD ? nk : AHEAD;
    for (int k = 0; k < pre; ++k) PC_STAGE(k);
    PC_WAIT(pre - 1);                                      // K-tile 0 landed
    __builtin_amdgcn_s_barrier();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + AHEAD < nk) PC_STAGE(kt + AHEAD);           // into the stage of K-tile kt-1, read before the barrier just passed
      const int last = kt + AHEAD < nk ? kt + AHEAD : nk - 1;      // newest K-tile issued
      PC_WAIT(last - (kt + 1));                            // K-tile kt+1 landed (nothing to wait for after the last one)
      __builtin_amdgcn_s_barrier();
    }
#undef PC_DMA
#undef PC_STAGE
#undef PC_WAIT
    return;
  }

  // ---------------- compute waves
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragment addresses inside a stage: row (lane & 31) of the wave's two 32-row blocks, k-slot kk * 2 + (lane >> 5) behind the row swizzle
  int aoff[2], woff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    aoff[i] = (wm * 64 + i * 32 + (lane & 31)) * 128;
    woff[i] = TILE_BYTES + (wn * 64 + i * 32 + (lane & 31)) * 128;
  }
  const int sw0 = ((wm * 64 + (lane & 31)) >> 1) & 7, sw1 = ((wn * 64 + (lane & 31)) >> 1) & 7;   // (+32 rows leave (row >> 1) & 7 unchanged)
  const int hi = lane >> 5;
#define PC_FRAGS(KK, BUF)                                                                                          \
  do {                                                                                                             \
    const int sa_ = (((KK) * 2 + hi) ^ sw0) << 4, sw_ = (((KK) * 2 + hi) ^ sw1) << 4;                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
      af[BUF][i] = *(const bf16x8*)(st + aoff[i] + sa_);                                                           \
      wf[BUF][i] = *(const bf16x8*)(st + woff[i] + sw_);                                                           \
    }                                                                                                              \
  } while (0)
  __builtin_amdgcn_s_barrier();                            // K-tile 0 landed
  for (int kt = 0; kt < nk; ++kt) {
    const char* st = smem_pc + (kt % PC_STAGES) * STAGE_BYTES;
    bf16x8 af[2][2], wf[2][2];                             // [buffer][32-row block]: the fragments of k-slice kk + 1 are read while the MFMAs of kk run
    PC_FRAGS(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) PC_FRAGS(kk + 1, (kk + 1) & 1);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][ni], af[kk & 1][mi], acc[ni][mi], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();                          // (also the "ring is free" barrier in front of the epilogue after the last K-tile)
  }
#undef PC_FRAGS
  tile128_out<EPI>(p, acc, smem_pc, wave, m0, n0, wm, wn, lane);
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

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_v2_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem2[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  GEMM_DYN_M(p);
  const int ntm = (p.M + BM2 - 1) / BM2, ntn = (p.N + BN - 1) / BN, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    if (i >= (xcd < r ? q + 1 : q)) return;          // (a device-side row count left this workgroup without a tile)
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
    if (kt + 2 < nk) STAGE2(kt + 2);
    const char* As = smem2 + (kt % 3) * STAGE2_BYTES;
    const char* Ws = As + A2_BYTES;
    const int ar0 = wm * 64 + (lane & 31), wr0 = wn * 64 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int s = kk * 2 + (lane >> 5);
      bf16x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ar = ar0 + i * 32, wr = wr0 + i * 32;
        af[i] = *(const bf16x8*)(As + ar * 128 + ((s ^ ((ar >> 1) & 7)) << 4));
        wf[i] = *(const bf16x8*)(Ws + wr * 128 + ((s ^ ((wr >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
  }
#undef STAGE2

  if constexpr (epi_is_bf16(EPI)) {
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

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_v3_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves, wave tile 128 (m) x 64 (n)
  GEMM_DYN_M(p);
  const int ntm = (p.M + BM3 - 1) / BM3, ntn = (p.N + BN3 - 1) / BN3, nt = ntm * ntn;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    if (i >= (xcd < r ? q + 1 : q)) return;          // (a device-side row count left this workgroup without a tile)
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  int tile_m, tile_n;
  tile_of(t, ntm, ntn, p.order, tile_m, tile_n);
  const int m0 = tile_m * BM3, n0 = tile_n * BN3;
  const int nk = p.K / BK;

  f32x16 acc[2][2][2];      // [m half][ni][mi]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

#define STAGE3(kt_)                                                                           \
  do {                                                                                        \
    char* b_ = smem3 + ((kt_) & 1) * STAGE3_BYTES;                                            \
    stage2_dma(p.A, p.lda, m0, p.M, (kt_) * BK, b_, tid, 4);                                  \
    stage2_dma(p.W, p.ldw, n0, p.N, (kt_) * BK, b_ + T3_BYTES, tid, 4);                       \
  } while (0)

  STAGE3(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                   // tile kt landed; everyone finished reading the other stage
    if (kt + 1 < nk) STAGE3(kt + 1);
    const char* As = smem3 + (kt & 1) * STAGE3_BYTES;
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
  }
#undef STAGE3
  __syncthreads();                                     // all waves done with the staging buffers
  char* wt = smem3 + wave * 16384;                     // 16 KiB private epilogue region per wave
  float cs_carry[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // column sums of both m-halves go out in one set of atomics
#pragma unroll
  for (int h = 0; h < 2; ++h)
    epilogue_bf16<EPI>(p, acc[h], wt, m0 + wm * 128 + h * 64, n0 + wn * 64, lane, cs_carry, h == 1);
}

// ------------------------------------------------------------------------------------------------------------
// p8: the 256x256x64 tile on an 8-phase schedule (cdna_hip_programming.md, "The 256^2 8-phase template"): the default
// kernel for the bf16-output GEMMs of the training step.  8 waves as 2(m) x 4(n), wave tile 128x64 = four 64x32
// quadrants; v_mfma_f32_16x16x32_bf16, 16 per quadrant per K-tile.  What differs from v3:
//  * A K-tile is staged as FOUR 16-KiB half-tiles (A-h0, A-h1, W-h0, W-h1; half h of an operand = the rows every wave
//    needs for its quadrants (h, *) resp. (*, h)), one half-tile = 2 LDS-DMA instructions per thread per PHASE, so the DMA
//    stream trickles through the whole K-tile instead of arriving as one burst behind a barrier.
//  * A phase = {LDS fragment reads of one quadrant's new operand half (12 / 4 / 8 / 0 ds_read_b128) || 2 LDS-DMA issues}
//    -> s_barrier -> 16 MFMA -> s_barrier.  The two wave rows (wr = 0 / 1; one wave of each per SIMD) run ONE barrier apart,
//    so on every SIMD one wave feeds the matrix pipe while its partner reads LDS / issues DMA.
//  * The DMA queue is never drained inside the loop: one counted s_waitcnt vmcnt(6) per K-tile (phase 4 / 8) leaves the
//    three newest half-tiles in flight across the raw s_barriers; three half-tiles (2-3 phases = 1-1.5k cycles) of latency
//    tolerance per load.  Fragment reads are inline-asm ds_read_b128, invisible to the compiler's waitcnt pass (which
//    would otherwise wait for every pending LDS-DMA before any LDS read); their completion is the explicit lgkmcnt(0)
//    in front of each MFMA cluster.
// Hazards (all by construction, never by timing; "phase" = two barriers):
//  RAW  a half-tile is read only in a phase AFTER the phase whose first barrier follows the vmcnt that retired it:
//       phase 4's wait retires the whole next K-tile, read in phases 5-7; phase 8's the one read in phases 1-3.
//  WAR  a slot is restaged >= 2 phases after the phase that read it (A-h0: read P1, restaged P3; W-h1: P2 -> P4; A-h1: P3 ->
//       next P1), or 1 phase after when an lgkmcnt BEFORE the reading phase's first barrier retired the reads (W-h0: the
//       four W reads are issued first in P1 and retired by lgkmcnt(8) there; restaged in P2).
// LDS image of a half-tile: 128 rows x 128 B, 16-B slot index XOR (row>>1)&7 (on the DMA source address and on the read
// address): every ds_read_b128 lane group of the 16x16x32 operand pattern hits 16 distinct slots.
constexpr int P8_HT = 128 * BK * 2;        // half-tile: 128 rows x 64 k = 16 KiB
constexpr int P8_BUF = 4 * P8_HT;          // K-tile: A-h0 | A-h1 | W-h0 | W-h1 = 64 KiB
constexpr int P8_LDS = 2 * P8_BUF;         // 128 KiB ring
constexpr int P8_XLDS = 8 * 4096;          // + 4 KiB per wave for the epilogue's transposition (outside the ring)

template <int OFF>
__device__ __forceinline__ void p8_dsr(bf16x8& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

// Epilogue of the 8-phase kernel.  The wave's 128x64 block leaves as FULL 128-B lines, 8 rows per store instruction
// (tools/store_bench.cpp: a CU sustains ~37-45 B/clk with full-line stores from 8 consecutive lanes, 26 B/clk when a row's 8
// lanes are scattered over the wave, 12 B/clk with 64-B row segments, 6 B/clk with the raw 32-B accumulator pieces), so each
// 16-row block is transposed through LDS -- through a 4-KiB per-wave slice of the 32 KiB the operand ring does NOT use, so the
// ring is free for the next tile's DMA while the epilogue runs.  The LDS accesses are inline asm: C++ accesses would make the
// compiler wait for every pending LDS-DMA (the next tile's pieces issued by `hook`) before each of them.
//   accumulator layout (16x16 block mi, ni): lane (m = lane & 15, g = lane >> 4) holds row mi*16 + m, columns ni*16 + 4g .. +3
//   image of a block: 16 rows x 128 B, 16-B chunk index XOR (row & 7); written 8 B per lane, read 16 B per lane row-major.
// INTERIOR: the whole 256x256 tile is inside the matrix (no predicates).  `hook(blk)` runs after the stores of block blk.
struct P8Out { uint32_t d[4]; };
__device__ __forceinline__ uint32_t p8_pack2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v; v[0] = (bf16)a; v[1] = (bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float p8_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float p8_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void p8_dsw64(uint32_t addr, uint32_t a, uint32_t b) {
  u32x2 v = {a, b};
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void p8_dsr128u(u32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

// HAS_EX: an R (EPI_BF16) / G (EPI_GELU_GRAD) operand is read.  Its loads and their uses are unconditional (edge tiles clamp the
// address): a load whose use sits behind a different branch would stay "pending" in the compiler's waitcnt model at the top of the
// K loop and cost a vmcnt(0) -- a drain of the DMA pipeline -- on every iteration.
template <int EPI, bool INTERIOR, bool HAS_EX, typename Hook>
__device__ __forceinline__ void p8_epilogue_blocks(const GemmP& p, f32x4 (&acc)[2][4][4], uint32_t xb /* LDS byte address of the wave's 4 KiB */,
                                                   int m_base, int n_base, int lane, Hook hook) {
  float scale = p.alpha;
  if (p.div_ptr) scale /= *p.div_ptr;
  const int m = lane & 15, g = lane >> 4, r8 = lane >> 3, c16 = lane & 7;
  f32x4 b[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) b[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      if (INTERIOR || n_base + ni * 16 + 4 * g < p.N) b[ni] = *(const f32x4*)(p.bias + n_base + ni * 16 + 4 * g);
  }
  uint32_t wad[4];                                              // write address of the (ni) piece: row m, columns ni*16 + 4g
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) wad[ni] = xb + (uint32_t)(m * 128 + ((((ni * 2 + (g >> 1)) ^ (m & 7)) << 4) | ((g & 1) << 3)));
  const uint32_t rad = xb + (uint32_t)(r8 * 128 + ((c16 ^ (r8 & 7)) << 4));   // rows r8 and r8 + 8 (+1024 B), chunk c16
  const long n = n_base + c16 * 8;
  const bool n_ok = INTERIOR || n < p.N;                        // N % 8 == 0: a chunk is all in or all out
  const long row0 = m_base + r8;
  bf16* cp = (bf16*)p.C + row0 * p.ldc + n;
  constexpr bool GELU2 = EPI == EPI_GELU || EPI == EPI_GELU_DERIV;    // two outputs
  constexpr bool DERIV = EPI == EPI_GELU_DERIV;
  constexpr bool MULG = EPI == EPI_MUL;
  bf16* c2p = GELU2 && p.C2 ? p.C2 + row0 * p.ldc2 + n : nullptr;
  u32x4 ex[8];                                                  // R / G pieces of one 64-row half in the store layout, all in flight at once
  const bf16* esrc = (EPI == EPI_GELU_GRAD || MULG) ? p.G : p.R;
  const long eld = (EPI == EPI_GELU_GRAD || MULG) ? p.ldg : p.ldr;
  const long nn = n_ok ? n : 0;
#define P8_EX_LOAD1(I_, R_)                                                                                            \
  do {                                                                                                                 \
    ex[I_] = *(const u32x4*)(esrc + (R_) * eld + nn);                                                                  \
  } while (0)
#define P8_LOAD_EX(H)                                                          \
  if constexpr (HAS_EX) {                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                         \
      long r_ = row0 + (H) * 64 + i_ * 8;                                      \
      if (!INTERIOR) r_ = r_ < p.M ? r_ : p.M - 1;                             \
      P8_EX_LOAD1(i_, r_);                                                     \
    }                                                                          \
  }
  P8_LOAD_EX(0);
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int blk = h * 4 + mi;
      const uint32_t bo = (blk & 1) * 2048;                     // two images per wave: block b+1 is written while b's reads return
      const bool ok1 = INTERIOR || (n_ok && row0 + blk * 16 < p.M), ok2 = INTERIOR || (n_ok && row0 + blk * 16 + 8 < p.M);
      float v[4][4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[ni][j] = acc[h][mi][ni][j] * scale + b[ni][j];
      // the accumulators of this block are zeroed for the next tile HERE, in the shadow of the block's LDS round trip and store
      // issue (a P8_ZERO after the epilogue is 128 VALU slots per wave with the MFMA pipe idle)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[h][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 o1, o2;
      if constexpr (GELU2) {
        float w[4][4];                                          // second output: pre-activation (GELU) or gelu'(pre) (GELU_DERIV)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const f32x2 x = {v[ni][j], v[ni][j + 1]};
            f32x2 gg, dd = x;                                     // EPI_GELU keeps the pre-activation as its second output
            gelu_erf_pair<DERIV>(x, gg, dd);
            v[ni][j] = gg.x; v[ni][j + 1] = gg.y;
            w[ni][j] = dd.x; w[ni][j + 1] = dd.y;
          }
        if (c2p) {
          // both outputs of the block go through the wave's two LDS images at once (second output in this block's image, the
          // activation in the other one): one LDS round trip per block instead of two
          u32x4 q1, q2;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) p8_dsw64<0>(wad[ni] + bo, p8_pack2(w[ni][0], w[ni][1]), p8_pack2(w[ni][2], w[ni][3]));
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) p8_dsw64<0>(wad[ni] + (bo ^ 2048u), p8_pack2(v[ni][0], v[ni][1]), p8_pack2(v[ni][2], v[ni][3]));
          p8_dsr128u<0>(q1, rad + bo);
          p8_dsr128u<1024>(q2, rad + bo);
          p8_dsr128u<0>(o1, rad + (bo ^ 2048u));
          p8_dsr128u<1024>(o2, rad + (bo ^ 2048u));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          if (ok1) *(u32x4*)(c2p + (long)blk * 16 * p.ldc2) = q1;
          if (ok2) *(u32x4*)(c2p + (long)blk * 16 * p.ldc2 + 8 * p.ldc2) = q2;
        }
      }
      // (the same wave's LDS operations execute in order: the image writes below cannot pass the reads above)
      if (!(GELU2 && c2p)) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) p8_dsw64<0>(wad[ni] + bo, p8_pack2(v[ni][0], v[ni][1]), p8_pack2(v[ni][2], v[ni][3]));
        p8_dsr128u<0>(o1, rad + bo);
        p8_dsr128u<1024>(o2, rad + bo);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (EPI == EPI_BF16 && HAS_EX) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          o1[d] = p8_pack2(p8_lo(o1[d]) + p8_lo(ex[2 * mi][d]), p8_hi(o1[d]) + p8_hi(ex[2 * mi][d]));
          o2[d] = p8_pack2(p8_lo(o2[d]) + p8_lo(ex[2 * mi + 1][d]), p8_hi(o2[d]) + p8_hi(ex[2 * mi + 1][d]));
        }
      }
      if constexpr (EPI == EPI_GELU_GRAD) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          o1[d] = p8_pack2(p8_lo(o1[d]) * gelu_erf_grad(p8_lo(ex[2 * mi][d])), p8_hi(o1[d]) * gelu_erf_grad(p8_hi(ex[2 * mi][d])));
          o2[d] = p8_pack2(p8_lo(o2[d]) * gelu_erf_grad(p8_lo(ex[2 * mi + 1][d])), p8_hi(o2[d]) * gelu_erf_grad(p8_hi(ex[2 * mi + 1][d])));
        }
      }
      if constexpr (EPI == EPI_MUL) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          o1[d] = p8_pack2(p8_lo(o1[d]) * p8_lo(ex[2 * mi][d]), p8_hi(o1[d]) * p8_hi(ex[2 * mi][d]));
          o2[d] = p8_pack2(p8_lo(o2[d]) * p8_lo(ex[2 * mi + 1][d]), p8_hi(o2[d]) * p8_hi(ex[2 * mi + 1][d]));
        }
      }
      if constexpr (!GELU2) {
        if (p.colsum) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            if (ok1) { cs[2 * d] += p8_lo(o1[d]); cs[2 * d + 1] += p8_hi(o1[d]); }
            if (ok2) { cs[2 * d] += p8_lo(o2[d]); cs[2 * d + 1] += p8_hi(o2[d]); }
          }
        }
      }
#ifdef P8_NOSTORE          // experiment (tools/Makefile gemm_bench_nostore): what the stores of the epilogue cost the NEXT tile's DMA waits
      if (p.M < 0) {
#endif
#ifdef P8_STORE_POLICY     // experiment: cache policy of the C stores ("nt", "sc1", "sc0 sc1")
      if (ok1) asm volatile("global_store_dwordx4 %0, %1, off " P8_STORE_POLICY :: "v"(cp + (long)blk * 16 * p.ldc), "v"(o1) : "memory");
      if (ok2) asm volatile("global_store_dwordx4 %0, %1, off " P8_STORE_POLICY :: "v"(cp + (long)blk * 16 * p.ldc + 8 * p.ldc), "v"(o2) : "memory");
#else
      if (ok1) *(u32x4*)(cp + (long)blk * 16 * p.ldc) = o1;
      if (ok2) *(u32x4*)(cp + (long)blk * 16 * p.ldc + 8 * p.ldc) = o2;
#endif
#ifdef P8_NOSTORE
      }
#endif
      hook(blk);
      // the second half's R / G pieces are requested as soon as the first half's pieces of the same block row have been used
      // (three blocks of lead instead of one: the loads come from HBM)
      if constexpr (HAS_EX && EPI == EPI_GELU_GRAD) {   // (this variant has no registers left for the staggered form: it spilled)
        if (blk == 3) { P8_LOAD_EX(1); }
      } else if constexpr (HAS_EX) {
        if (h == 0) {
#pragma unroll
          for (int i_ = 2 * mi; i_ < 2 * mi + 2; ++i_) {
            long r_ = row0 + 64 + i_ * 8;
            if (!INTERIOR) r_ = r_ < p.M ? r_ : p.M - 1;
            P8_EX_LOAD1(i_, r_);
          }
        }
      }
    }
#undef P8_LOAD_EX
#undef P8_EX_LOAD1
  if constexpr (!GELU2) {
    if (p.colsum) {   // lanes l, l^8, l^16, l^32 hold the same 8 columns for different rows
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = cs[e];
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        cs[e] = t;
      }
      if (lane < 8 && n_ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(p.colsum + n + e, cs[e]);
      }
    }
  }
}

// Persistent: gridDim.x workgroups (<= one per CU) walk the tile list with stride gridDim.x, so a CU pays the workgroup
// launch, the kernel-argument fetch and the address set-up once, and the tile boundary is {epilogue from registers -> DMA
// prologue of the next tile} instead of {drain, exit, dispatch, prologue}.  (PMC at 84000x3072x768, one workgroup per tile:
// the MFMA pipe was busy 49 % of the CU-busy cycles against 82 % at 4096^3 -- ~10 us of boundary per 13.6-us main loop.)
template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt_p8_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem8[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;          // 2 x 4 waves, wave tile 128 (m) x 64 (n)
  GEMM_DYN_M(p);                                     // device-side row count: fewer row panels than the grid was sized for
  const int ntm = (p.M + BM3 - 1) / BM3, ntn = (p.N + BN3 - 1) / BN3, nt = ntm * ntn;
  if ((int)blockIdx.x >= nt) return;                 // (whole workgroups: no barrier has been executed yet)
  const int nk = p.K / BK;                           // even, >= 2 (launcher)

  f32x4 acc[2][4][4];                                // [m half][mi][ni], 128 accumulators
#define P8_ZERO()                                                              \
  _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                             \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                             \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) acc[h_][i_][j_] = f32x4{0.f, 0.f, 0.f, 0.f}
  P8_ZERO();

  // ---- LDS-DMA sources.  DMA instruction i (0/1) of a half-tile: chunk id = i*512 + tid -> half-tile row lr = id >> 3,
  // physical slot id & 7 (LDS destination is lane-linear), logical k-slot = slot ^ ((lr >> 1) & 7).
  // A-h: lr -> tile row (lr >> 6) * 128 + h * 64 + (lr & 63);   W-h: lr -> tile column (lr >> 5) * 64 + h * 32 + (lr & 31).
  uint32_t offA[2][2], offW[2][2];                   // byte offsets from p.A / p.W (+ k), [half][instruction]
  const uint32_t ls16 = (uint32_t)(((tid & 7) ^ ((tid >> 4) & 7)) * 16);
  int m0, n0;
#define P8_TILE(VB)                                                                                           \
  do {                                                                                                        \
    const int b_ = (VB), q_ = nt >> 3, r_ = nt & 7, xcd_ = b_ & 7, i_ = b_ >> 3;                              \
    const int t_ = (xcd_ < r_ ? xcd_ * (q_ + 1) : r_ * (q_ + 1) + (xcd_ - r_) * q_) + i_;                     \
    int tm_, tn_;                                                                                             \
    tile_of(t_, ntm, ntn, p.order, tm_, tn_);                                                                 \
    m0 = tm_ * BM3; n0 = tn_ * BN3;                                                                           \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                        \
      int row_ = m0 + j_ * 128 + h_ * 64 + (tid >> 3);                                                        \
      row_ = row_ < p.M ? row_ : p.M - 1;              /* clamped rows are computed and never stored */       \
      offA[h_][j_] = (uint32_t)row_ * (uint32_t)(p.lda * 2) + ls16;                                                      \
      int col_ = n0 + (j_ * 2 + (tid >> 8)) * 64 + h_ * 32 + ((tid >> 3) & 31);                               \
      col_ = col_ < p.N ? col_ : p.N - 1;                                                                     \
      offW[h_][j_] = (uint32_t)col_ * (uint32_t)(p.ldw * 2) + ls16;                                                      \
    }                                                                                                         \
  } while (0)
  const char* gA = (const char*)p.A;
  const char* gW = (const char*)p.W;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem8;
  // LDS-DMA in its SGPR-base + 32-bit-VGPR-offset form (inline asm: the builtin materialises a 64-bit VGPR address per load --
  // 16 VGPRs of loop-invariant pointers plus three VALU per load -- and this schedule counts its DMA by hand anyway).
  // M0 = wave-uniform LDS destination; one wait state between the M0 write and its use.
#define P8_STG(GB, O, SLOT)                                                                                               \
  do {                                                                                                                    \
    const uint32_t d_ = lds0 + (SLOT) + wave * 1024;                                                                      \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((O)[0]), "s"(GB), "s"(d_) : "memory", "m0");          \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((O)[1]), "s"(GB), "s"(d_ + 8192) : "memory", "m0");   \
  } while (0)
#define P8_STG_A(H, BUF, KT) P8_STG((const char*)(gA + (size_t)(KT) * (BK * 2)), offA[H], (BUF) * P8_BUF + (H) * P8_HT)
#define P8_STG_W(H, BUF, KT) P8_STG((const char*)(gW + (size_t)(KT) * (BK * 2)), offW[H], (BUF) * P8_BUF + (2 + (H)) * P8_HT)
  // K-tile 0 complete + the first three half-tiles of K-tile 1: what the main loop expects to find in flight
#define P8_PROLOGUE()                                                                        \
  do {                                                                                       \
    P8_STG_A(0, 0, 0); P8_STG_W(0, 0, 0); P8_STG_W(1, 0, 0); P8_STG_A(1, 0, 0);              \
    P8_STG_W(0, 1, 1); P8_STG_A(0, 1, 1); P8_STG_W(1, 1, 1);                                 \
  } while (0)

  // ---- fragment read addresses (16x16x32 operand: lane -> row lane & 15, k-slot kk*4 + (lane >> 4))
  uint32_t aad[2][2], wad[2][2];                     // [buffer][kk]
  {
    const uint32_t sw = (uint32_t)((lane >> 1) & 7), r16 = (uint32_t)(lane & 15), q = (uint32_t)(lane >> 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint32_t slot = ((((uint32_t)kk * 4 + q) ^ sw) << 4);
      aad[0][kk] = lds0 + ((uint32_t)wr * 64 + r16) * 128 + slot;
      wad[0][kk] = lds0 + 2 * P8_HT + ((uint32_t)wc * 32 + r16) * 128 + slot;
      aad[1][kk] = aad[0][kk] + P8_BUF;
      wad[1][kk] = wad[0][kk] + P8_BUF;
    }
  }
  bf16x8 fa[4][2], fw0[2][2], fw1[2][2];             // A half [mi][kk]; W halves [ni][kk]
#define P8_RD_A(BUF, H)                                                                       \
  do {                                                                                        \
    p8_dsr<(H) * P8_HT + 0 * 2048>(fa[0][0], aad[BUF][0]);                                    \
    p8_dsr<(H) * P8_HT + 1 * 2048>(fa[1][0], aad[BUF][0]);                                    \
    p8_dsr<(H) * P8_HT + 2 * 2048>(fa[2][0], aad[BUF][0]);                                    \
    p8_dsr<(H) * P8_HT + 3 * 2048>(fa[3][0], aad[BUF][0]);                                    \
    p8_dsr<(H) * P8_HT + 0 * 2048>(fa[0][1], aad[BUF][1]);                                    \
    p8_dsr<(H) * P8_HT + 1 * 2048>(fa[1][1], aad[BUF][1]);                                    \
    p8_dsr<(H) * P8_HT + 2 * 2048>(fa[2][1], aad[BUF][1]);                                    \
    p8_dsr<(H) * P8_HT + 3 * 2048>(fa[3][1], aad[BUF][1]);                                    \
  } while (0)
#define P8_RD_W(BUF, H, FW)                                                                   \
  do {                                                                                        \
    p8_dsr<(H) * P8_HT + 0 * 2048>(FW[0][0], wad[BUF][0]);                                    \
    p8_dsr<(H) * P8_HT + 1 * 2048>(FW[1][0], wad[BUF][0]);                                    \
    p8_dsr<(H) * P8_HT + 0 * 2048>(FW[0][1], wad[BUF][1]);                                    \
    p8_dsr<(H) * P8_HT + 1 * 2048>(FW[1][1], wad[BUF][1]);                                    \
  } while (0)
  // one quadrant x K=64: 16 MFMA, 8 independent accumulators between dependent pairs
#define P8_MM(H, NH, FW)                                                                                        \
  do {                                                                                                          \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                                         \
    _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_)                                                         \
    _Pragma("unroll") for (int ni_ = 0; ni_ < 2; ++ni_)                                                         \
      acc[H][mi_][(NH) * 2 + ni_] =                                                                             \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(FW[ni_][kk_], fa[mi_][kk_], acc[H][mi_][(NH) * 2 + ni_], 0, 0, 0); \
  } while (0)
#define P8_BAR() __builtin_amdgcn_s_barrier()
#define P8_SB() __builtin_amdgcn_sched_barrier(0)
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  // barrier -> operands landed -> MFMA cluster at raised priority -> barrier
#define P8_COMPUTE(H, NH, FW)                    \
  do {                                           \
    P8_BAR();                                    \
    P8_LGKM0();                                  \
    P8_SB();                                     \
    __builtin_amdgcn_s_setprio(1);               \
    P8_MM(H, NH, FW);                            \
    __builtin_amdgcn_s_setprio(0);               \
    P8_SB();                                     \
    P8_BAR();                                    \
    P8_SB();                                     \
  } while (0)

#ifdef P8_PROFILE
  unsigned long long* prof = (unsigned long long*)p.colsum + (size_t)blockIdx.x * 16 * 8;
  p.colsum = nullptr;
  int prof_i = 0;
#define P8_STAMP(K) do { if (prof && tid == 0 && prof_i < 16) prof[prof_i * 8 + (K)] = __builtin_readcyclecounter(); } while (0)
#else
#define P8_STAMP(K) do { } while (0)
#endif
  int vb = blockIdx.x;
  bool pre = false;
  P8_TILE(vb);
  P8_PROLOGUE();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K-tile 0 landed, three half-tiles of K-tile 1 in flight
  P8_BAR();
  P8_SB();
  if (wr == 1) P8_BAR();                             // the second wave row runs one barrier behind the first, for the whole launch
  for (;;) {
    P8_STAMP(0);
    // The K-tile stream does not stop at a tile boundary: the LAST K-tile pair of a tile stages the first two K-tiles of the
    // workgroup's next tile (same slots, same phases, same counted waits), so the next main loop finds exactly what the
    // prologue would have left -- K-tile 0 landed, three half-tiles of K-tile 1 in flight -- and the boundary is the epilogue alone.
    const int cm0 = m0, cn0 = n0;
    const int vbn = vb + (int)gridDim.x;
    const bool has_next = vbn < nt;                  // wave-uniform
    // an interior tile's epilogue issues AT LEAST 16 vector-memory operations per lane (the unpredicated stores of C; more with
    // a second output, R / G loads or column-sum atomics): the next tile's first pair may leave 16 of them outstanding
    const bool pre_next = has_next && cm0 + BM3 <= p.M && cn0 + BN3 <= p.N;
    for (int kt = 0; kt < nk; kt += 2) {
      const bool last = kt + 2 >= nk;
      const bool more = !last || has_next;
      int kn = kt + 2;                               // K-tile (of the tile whose offsets are loaded) staged from phase 2 on
      // ---------------- K-tile kt (buffer 0)
      P8_RD_W(0, 0, fw0); P8_SB(); P8_RD_A(0, 0);                                 // phase 1
      if (!(pre && kt == 0)) P8_STG_A(1, 1, kt + 1);                              // (pre: staged before the previous tile's epilogue)
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                          // the four W-h0 reads: slot free for phase 2
      P8_COMPUTE(0, 0, fw0);
      if (last && has_next) { P8_TILE(vbn); kn = 0; }                             // from here on the DMA sources are the next tile's
      P8_RD_W(0, 1, fw1);                                                         // phase 2
      if (more) P8_STG_W(0, 0, kn);
      P8_COMPUTE(0, 1, fw1);
      P8_RD_A(0, 1);                                                              // phase 3
      if (more) P8_STG_A(0, 0, kn);
      P8_COMPUTE(1, 1, fw1);
      if (more) {                                                                 // phase 4
        P8_STG_W(1, 0, kn);
        // K-tile kt+1 complete (this wave's pieces).  First pair after a boundary: its pieces were all issued BEFORE the previous
        // tile's stores, the six loads above after them -- vmcnt counts loads and stores in issue order, so the stores may stay
        // outstanding (waiting for them here costs the first K-tile of every tile)
        if (pre && kt == 0) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      P8_COMPUTE(1, 0, fw0);
      // ---------------- K-tile kt+1 (buffer 1)
      P8_RD_W(1, 0, fw0); P8_SB(); P8_RD_A(1, 0);                                 // phase 5
      if (more) P8_STG_A(1, 0, kn);
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      P8_COMPUTE(0, 0, fw0);
      P8_RD_W(1, 1, fw1);                                                         // phase 6
      if (more) P8_STG_W(0, 1, kn + 1);
      P8_COMPUTE(0, 1, fw1);
      P8_RD_A(1, 1);                                                              // phase 7
      if (more) P8_STG_A(0, 1, kn + 1);
      P8_COMPUTE(1, 1, fw1);
      if (more) {                                                                 // phase 8
        P8_STG_W(1, 1, kn + 1);
        if (last && pre_next) {                                                   // the fourth half-tile of the next tile's K-tile 1 too:
          P8_STG_A(1, 1, kn + 1);                                                 // nothing of that K-tile is then issued after the stores
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                        // the next K-tile for buffer 0 complete
        }
      }
      P8_COMPUTE(1, 0, fw0);
    }
    P8_STAMP(2);

    // ---- tile boundary: every wave writes its own 128x64 block (no barrier; the wave rows stay one barrier apart)
    {
      // everything the epilogue derives from the lane id is computed HERE, per tile: hoisted out of the tile loop it would live
      // across the main loop and spill
      // ... and so is everything it derives from the kernel arguments (row strides x block offsets, pointers): the copies below
      // are opaque to loop-invariant code motion.  (Hoisted, they cost 50 spilled VGPRs and the compiler's vmcnt(0) on the
      // first use of a reloaded value INSIDE the K loop -- which drains the DMA pipeline on every iteration.)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      GemmP q = p;
      asm volatile("" : "+s"(q.C), "+s"(q.ldc), "+s"(q.C2), "+s"(q.ldc2));
      asm volatile("" : "+s"(q.R), "+s"(q.ldr), "+s"(q.G), "+s"(q.ldg));
      asm volatile("" : "+s"(q.bias), "+s"(q.colsum), "+s"(q.div_ptr), "+s"(q.alpha), "+s"(q.M), "+s"(q.N));
      auto hook = [](int) {};
      const uint32_t xb = lds0 + P8_LDS + wave * 4096;
      const int mb = cm0 + wr * 128, nb = cn0 + wc * 64;
      const bool interior = cm0 + BM3 <= q.M && cn0 + BN3 <= q.N;
      if constexpr (EPI == EPI_GELU_GRAD || EPI == EPI_MUL) {
        if (interior) p8_epilogue_blocks<EPI, true, true>(q, acc, xb, mb, nb, lane_e, hook);
        else p8_epilogue_blocks<EPI, false, true>(q, acc, xb, mb, nb, lane_e, hook);
      } else if (EPI == EPI_BF16 && q.R != nullptr) {
        if (interior) p8_epilogue_blocks<EPI, true, true>(q, acc, xb, mb, nb, lane_e, hook);
        else p8_epilogue_blocks<EPI, false, true>(q, acc, xb, mb, nb, lane_e, hook);
      } else {
        if (interior) p8_epilogue_blocks<EPI, true, false>(q, acc, xb, mb, nb, lane_e, hook);
        else p8_epilogue_blocks<EPI, false, false>(q, acc, xb, mb, nb, lane_e, hook);
      }
    }
    P8_STAMP(4);
#ifdef P8_PROFILE
    ++prof_i;
#endif
    if (!has_next) break;
    vb = vbn;
    pre = pre_next;
    P8_SB();
  }
  if (wr == 0) P8_BAR();                             // balance the barrier count of the two wave rows
#undef P8_ZERO
#undef P8_STAMP
#undef P8_TILE
#undef P8_PROLOGUE
#undef P8_STG
#undef P8_STG_A
#undef P8_STG_W
#undef P8_RD_A
#undef P8_RD_W
#undef P8_MM
#undef P8_COMPUTE
#undef P8_BAR
#undef P8_SB
#undef P8_LGKM0
}

// ---- launchers.  The dynamic-LDS attribute of a kernel is raised exactly once (function-local static: thread-safe).
template <typename K>
int raise_lds(K kernel, int bytes, const char* what) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    spmm_set_error("spmm_gemm_nt: cannot raise the dynamic LDS of %s to %d: %s", what, bytes, hipGetErrorString(e));
    return SPMM_ERR_LAUNCH;
  }
  return SPMM_OK;
}

template <int EPI>
int launch_p8_one(const GemmP& p, hipStream_t st, bool persist) {
  static const int rc = raise_lds(gemm_nt_p8_kernel<EPI>, P8_LDS + P8_XLDS, "the 8-phase kernel");
  if (rc != SPMM_OK) return rc;
  const int nt = ((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3);
  static const int ncu = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n & ~7 : 256; }();
  // persistent: one workgroup per CU (a multiple of 8 so that a workgroup's tiles stay on its XCD's contiguous range)
  dim3 grid(persist && nt > ncu ? ncu : nt);
  hipLaunchKernelGGL((gemm_nt_p8_kernel<EPI>), grid, dim3(512), P8_LDS + P8_XLDS, st, p);
  return SPMM_OK;
}
int launch_p8(int epi, const GemmP& p, hipStream_t st, bool persist) {
  switch (epi) {
    case EPI_BF16: return launch_p8_one<EPI_BF16>(p, st, persist);
    case EPI_GELU: return launch_p8_one<EPI_GELU>(p, st, persist);
    case EPI_GELU_GRAD: return launch_p8_one<EPI_GELU_GRAD>(p, st, persist);
    case EPI_GELU_DERIV: return launch_p8_one<EPI_GELU_DERIV>(p, st, persist);
    case EPI_MUL: return launch_p8_one<EPI_MUL>(p, st, persist);
    default: return -1;
  }
}

template <int EPI>
int launch_v3_one(const GemmP& p, hipStream_t st) {
  static const int rc = raise_lds(gemm_nt_v3_kernel<EPI>, LDS3_BYTES, "the 256x256 kernel");
  if (rc != SPMM_OK) return rc;
  dim3 grid(((p.M + BM3 - 1) / BM3) * ((p.N + BN3 - 1) / BN3));
  hipLaunchKernelGGL((gemm_nt_v3_kernel<EPI>), grid, dim3(512), LDS3_BYTES, st, p);
  return SPMM_OK;
}
int launch_v3(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_v3_one<EPI_BF16>(p, st);
    case EPI_GELU: return launch_v3_one<EPI_GELU>(p, st);
    case EPI_GELU_GRAD: return launch_v3_one<EPI_GELU_GRAD>(p, st);
    case EPI_GELU_DERIV: return launch_v3_one<EPI_GELU_DERIV>(p, st);
    case EPI_MUL: return launch_v3_one<EPI_MUL>(p, st);
    default: return -1;
  }
}

template <int EPI>
int launch_v2_one(const GemmP& p, hipStream_t st) {
  static const int rc = raise_lds(gemm_nt_v2_kernel<EPI>, LDS2_BYTES, "the 256x128 kernel");
  if (rc != SPMM_OK) return rc;
  dim3 grid(((p.M + BM2 - 1) / BM2) * ((p.N + BN - 1) / BN));
  hipLaunchKernelGGL((gemm_nt_v2_kernel<EPI>), grid, dim3(512), LDS2_BYTES, st, p);
  return SPMM_OK;
}
int launch_v2(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch_v2_one<EPI_BF16>(p, st);
    case EPI_GELU: return launch_v2_one<EPI_GELU>(p, st);
    case EPI_F32: return launch_v2_one<EPI_F32>(p, st);
    case EPI_GELU_GRAD: return launch_v2_one<EPI_GELU_GRAD>(p, st);
    case EPI_GELU_DERIV: return launch_v2_one<EPI_GELU_DERIV>(p, st);
    case EPI_MUL: return launch_v2_one<EPI_MUL>(p, st);
    case EPI_F32_ACC: return launch_v2_one<EPI_F32_ACC>(p, st);
    default: return -1;
  }
}

template <typename K>
int pc_attr(K kernel) {                              // the dynamic-LDS attribute of a kernel is raised exactly once
  static const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS);
  return e == hipSuccess ? 0 : 1;
}
#define PC_LAUNCH(E)                                                                                   \
  case E:                                                                                              \
    if (pc_attr(gemm_nt_pc_kernel<E>)) { spmm_set_error("spmm_gemm_nt: cannot raise the LDS limit of the loader/compute kernel"); return SPMM_ERR_LAUNCH; } \
    hipLaunchKernelGGL((gemm_nt_pc_kernel<E>), grid, dim3(PC_THREADS), PC_LDS, st, p);                        \
    break
int launch_pc(int epi, const GemmP& p, dim3 grid, hipStream_t st) {
  switch (epi) {
    PC_LAUNCH(EPI_BF16); PC_LAUNCH(EPI_GELU); PC_LAUNCH(EPI_F32); PC_LAUNCH(EPI_F32_ATOMIC); PC_LAUNCH(EPI_GELU_GRAD); PC_LAUNCH(EPI_F32_ACC);
    PC_LAUNCH(EPI_GELU_DERIV); PC_LAUNCH(EPI_MUL);
    default: return -1;
  }
  return 0;
}
#undef PC_LAUNCH

int launch_v1(int epi, const GemmP& p, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: hipLaunchKernelGGL((gemm_nt_kernel<EPI_BF16>), grid, dim3(256), 0, st, p); break;
    case EPI_GELU: hipLaunchKernelGGL((gemm_nt_kernel<EPI_GELU>), grid, dim3(256), 0, st, p); break;
    case EPI_F32: hipLaunchKernelGGL((gemm_nt_kernel<EPI_F32>), grid, dim3(256), 0, st, p); break;
    case EPI_F32_ATOMIC: hipLaunchKernelGGL((gemm_nt_kernel<EPI_F32_ATOMIC>), grid, dim3(256), 0, st, p); break;
    case EPI_GELU_GRAD: hipLaunchKernelGGL((gemm_nt_kernel<EPI_GELU_GRAD>), grid, dim3(256), 0, st, p); break;
    case EPI_F32_ACC: hipLaunchKernelGGL((gemm_nt_kernel<EPI_F32_ACC>), grid, dim3(256), 0, st, p); break;
    case EPI_GELU_DERIV: hipLaunchKernelGGL((gemm_nt_kernel<EPI_GELU_DERIV>), grid, dim3(256), 0, st, p); break;
    case EPI_MUL: hipLaunchKernelGGL((gemm_nt_kernel<EPI_MUL>), grid, dim3(256), 0, st, p); break;
    default: return -1;
  }
  return SPMM_OK;
}

inline bool is_bf16_epi(int epi) { return epi_is_bf16(epi); }
// the 8-phase kernel walks K two tiles at a time, addresses its DMA sources with 32-bit byte offsets and stores whole 16-B
// chunks of a row (N % 8 == 0)
inline bool p8_ok(const GemmP& p, int epi) {
  return is_bf16_epi(epi) && p.K % 128 == 0 && p.N % 8 == 0 && (unsigned long)p.M * (unsigned long)p.lda * 2ul < (1ul << 32) &&
         (unsigned long)p.N * (unsigned long)p.ldw * 2ul < (1ul << 32);
}

// Which tile kernel for an M x N output?  Per-tile efficiency (bytes through the per-CU load path per FLOP) favours the big
// tile, wave quantisation on 256 CUs favours the small one: score = efficiency x fill of the last wave.  The 256x256 and
// 256x128 kernels hold one workgroup per CU (128 / 144 KiB LDS); the 128x128 kernel (32 KiB) runs several per CU, so its fill
// is smooth.  Constants from tools/bench_gemm_tiles.py.  Used for the decoder's shapes (M = beams x molecules < 6000), e.g.
// 5000x768x3072: 640 TF with 128x128 vs 472 with 256x128 vs 292 with 256x256.
int pick_tile(int M, int N, int epi) {
  const bool bf = is_bf16_epi(epi);
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
  // (1.20: with the 8-phase kernel behind choice 3 the big tile wins from a fill of 0.67 on -- 5000x2304x768 runs 25.6 us on it
  // against 33.0 on 128x128 tiles and 34.9 on the one-barrier 256x256 kernel the 1.00 was measured with; tools/gemm_small_m.py)
  const double s3 = bf ? 1.20 * fill(t3) : 0.0;
  const double s2 = (M >= 512 && epi != EPI_F32_ATOMIC) ? 0.85 * fill(t2) : 0.0;
  const double s1 = 0.80 * (t1 < 256 ? t1 / 256.0 : 1.0);
  if (s3 >= s2 && s3 >= s1) return 3;
  return s2 >= s1 ? 2 : 1;
}

}  // namespace

extern "C" int spmm_gemm_nt(const void* A, long lda, const void* W, long ldw, int M, int N, int K, int splits,
                            const float* bias, const float* div_ptr, float alpha, const void* R, long ldr,
                            const void* G, long ldg, void* C, long ldc, void* C2, long ldc2, int epi, float* colsum,
                            int kernel, const int* M_dev, hipStream_t stream) {
  SPMM_CHECK_SHAPE(M > 0 && N > 0 && K > 0, "spmm_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  SPMM_CHECK_SHAPE(K % 64 == 0, "spmm_gemm_nt: K=%d must be a multiple of 64", K);
  SPMM_CHECK_SHAPE(N % 4 == 0, "spmm_gemm_nt: N=%d must be a multiple of 4", N);
  SPMM_CHECK_SHAPE(lda % 8 == 0 && ldw % 8 == 0, "spmm_gemm_nt: lda/ldw must be multiples of 8 elements");
  SPMM_CHECK_SHAPE(ldc % 4 == 0, "spmm_gemm_nt: ldc must be a multiple of 4");
  SPMM_CHECK_SHAPE(!is_bf16_epi(epi) || (ldc % 8 == 0 && (!R || ldr % 8 == 0) && (!G || ldg % 8 == 0) && (!C2 || ldc2 % 8 == 0) && (uintptr_t)C % 16 == 0),
                   "spmm_gemm_nt: bf16 outputs need 16-B aligned rows (ldc/ldr/ldg/ldc2 multiples of 8)");
  SPMM_CHECK_SHAPE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0), "spmm_gemm_nt: A/W must be 16-B aligned");
  SPMM_CHECK_SHAPE(epi >= EPI_BF16 && epi <= EPI_MUL, "spmm_gemm_nt: unknown epilogue %d", epi);
  if (splits < 1) splits = 1;
  SPMM_CHECK_SHAPE(splits == 1 || epi == EPI_F32_ATOMIC, "spmm_gemm_nt: split-K needs the atomic epilogue");
  SPMM_CHECK_SHAPE((epi != EPI_GELU_GRAD && epi != EPI_MUL) || G != nullptr, "spmm_gemm_nt: the GELU-grad / multiply epilogues need G");
  SPMM_CHECK_SHAPE(colsum == nullptr || epi == EPI_BF16 || epi == EPI_GELU_GRAD || epi == EPI_MUL, "spmm_gemm_nt: colsum is only fused into the bf16 / GELU-grad / multiply epilogues");
  int ksplit = ((K / 64 + splits - 1) / splits) * 64;
  splits = (K + ksplit - 1) / ksplit;
  GemmP p;
  p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw;
  p.M = M; p.N = N; p.K = K; p.ksplit = ksplit; p.bias = bias; p.div_ptr = div_ptr; p.alpha = alpha;
  p.R = (const bf16*)R; p.ldr = ldr; p.G = (const bf16*)G; p.ldg = ldg; p.C = C; p.ldc = ldc;
  p.C2 = (bf16*)C2; p.ldc2 = ldc2; p.colsum = colsum;
  p.M_ptr = nullptr;
  // tile order (tile_of): K <= 1024 -> 2 (blocks of 8 row panels x up to 4 column tiles per XCD), longer K -> 0 (row-major).  Measured in
  // the cache state the step presents (a 256-MiB memset between launches, tools/gemm_bench sustain GEMM_BENCH_BETWEEN=1): 84256x2304x768
  // 1 072 TF/s with order 2 against 960 with the column-group order 3 used before (which had the lowest counter traffic with warm
  // caches, profiles/r01_pmc_nt_gemm.txt) and 1 040 row-major; x3072x768 1 113 / 1 105 / 1 050; x768x768 893 / 874 / 880;
  // K = 3072: row-major 1 313 against 1 268.
  p.order = K > 1024 ? 0 : 2;

  // kernel: 0 = choose (below); 1 = 128x128 (all epilogues, split-K); 2 = 256x128 three-stage ring (no atomics);
  // 3 = 256x256 one-barrier-per-k-step (bf16 outputs); 5 = 128x128 with loader + compute waves; 8 = 256x256 8-phase (bf16 outputs, K % 128 == 0)
  int k = kernel == SPMM_GEMM_AUTO_TILES ? 0 : kernel;
  if (k == 0) {
    const int tile = splits > 1 ? 1 : pick_tile(M, N, epi);
    k = tile == 3 ? (p8_ok(p, epi) ? (kernel == SPMM_GEMM_AUTO_TILES ? 9 : 8) : 3) : tile;
    if (k == 2 && (epi == EPI_F32_ATOMIC || M < 512)) k = 1;
  }
  // 5 = the 128x128 tile with loader + compute waves (any epilogue of kernel 1, split-K too; operands < 4 GiB: 32-bit DMA offsets)
  const bool pc_fits = (unsigned long)p.M * (unsigned long)p.lda * 2ul < (1ul << 32) && (unsigned long)p.N * (unsigned long)p.ldw * 2ul < (1ul << 32);
  const long t128 = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * splits;
  if (k == 1 && kernel != 1 && pc_fits && PC_AUTO && t128 <= 256 && ksplit >= 4 * BK) k = 5;
  SPMM_CHECK_SHAPE(k == 1 || k == 2 || k == 3 || k == 5 || k == 8 || k == 9, "spmm_gemm_nt: unknown kernel selector %d", kernel);
  SPMM_CHECK_SHAPE(k != 5 || pc_fits, "spmm_gemm_nt: kernel 5 addresses its operands with 32-bit byte offsets (< 4 GiB each)");
  SPMM_CHECK_SHAPE((k != 8 && k != 9) || p8_ok(p, epi), "spmm_gemm_nt: the 8-phase kernel needs a bf16-output epilogue, K %% 128 == 0, N %% 8 == 0 and operands < 4 GiB");
  SPMM_CHECK_SHAPE(k != 3 || is_bf16_epi(epi), "spmm_gemm_nt: the 256x256 kernel has bf16-output epilogues only");
  SPMM_CHECK_SHAPE(k != 2 || epi != EPI_F32_ATOMIC, "spmm_gemm_nt: the 256x128 kernel has no atomic epilogue");
  SPMM_CHECK_SHAPE(k == 1 || k == 5 || splits == 1, "spmm_gemm_nt: split-K runs on the 128x128 kernels only");
  p.M_ptr = M_dev;
  int rc;
  if (k == 8 || k == 9) rc = launch_p8(epi, p, stream, k == 8);     // 9: one workgroup per tile (A/B of the persistent walk)
  else if (k == 3) rc = launch_v3(epi, p, stream);
  else if (k == 2) rc = launch_v2(epi, p, stream);
  else if (k == 5) rc = launch_pc(epi, p, dim3(((M + BM - 1) / BM) * ((N + BN - 1) / BN), 1, splits), stream);
  else rc = launch_v1(epi, p, dim3(((M + BM - 1) / BM) * ((N + BN - 1) / BN), 1, splits), stream);
  if (rc > 0) return rc;
  if (rc < 0) { spmm_set_error("spmm_gemm_nt: epilogue %d is not built for kernel %d", epi, k); return SPMM_ERR_UNSUPPORTED; }
  SPMM_LAUNCH_CHECK("spmm_gemm_nt");
  return SPMM_OK;
}
