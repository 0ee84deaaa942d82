// bf16 "TN" GEMM for gfx950 -- the weight-gradient GEMM:  C[N,K] += A[M,N]^T * B[M,K]  (reduction over the ROW index m
// of both operands, A = dY and B = X token-major exactly as the forward pass left them: no transposed copies).
//
// Replaces autograd's weight-gradient matmul for every nn.Linear on the path (the backward of xbert.py:280-300,
// :370, :435, :448, :673, :695 and SPMM_models.py:31-42).
//
// v_mfma_f32_32x32x16_bf16 wants, per lane, 8 consecutive REDUCTION elements of one row of its operand; here the
// reduction index is the memory row, so fragments are fetched with the gfx950 LDS transpose read
// ds_read_b64_tr_b16: within a 16-lane group, lane i supplies the address of 4 contiguous bf16 (row i>>2,
// columns 4*(i&3)..+3 of a [4 x 16] block) and receives column i of that block (4 reduction rows).  Two such reads
// give the 8 k-slot values of a lane; MFMA sums its k-slots in any order and A and B use the same order.
// Tiles: 64 (reduction rows) x 128 (columns) per operand, 256-B LDS rows; the 16-B chunk index is XOR-swizzled with
// (row&3)<<2 so the four rows a read group touches sit in different 32-B bank spans (conflict-free for the 2x32-lane
// service groups of ds_read_b64_tr_b16).  Full tiles are staged by LDS-DMA (global_load_lds_dwordx4) with the
// swizzle applied on the source address; the ragged last tile goes through registers with zero fill.
// Output tile 128x128 per 256-thread workgroup (4 waves as 2x2), MFMA issued as D[k][n] so a lane owns 4 consecutive
// k -> 16-byte fp32 stores.  Split over M: every split writes its partial tile to a slab with plain stores (fp32
// atomics measured 20x slower than the same bytes as plain stores), a second pass reduces the slabs into C.
#include "common.h"
#include "../../include/spmm_hip.h"

namespace {

constexpr int BR = 64;                      // reduction rows per stage
constexpr int BT = 128;                     // output tile edge
constexpr int TILE_BYTES = BR * BT * 2;     // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

struct TnP {
  const bf16* A; long lda;    // [M, N]
  const bf16* B; long ldb;    // [M, K]
  int M, N, K;
  int rsplit;                 // reduction rows per grid.z slice (multiple of 64)
  float* C; long ldc;         // direct mode: C[n*ldc + k] += ...
  float* slab;                // slab mode: slab[z][n][k] (dense N*K)
  float alpha;
  int nsplit;                 // 8-phase kernel: > 0 = 1-D grid of tiles x nsplit workgroups in XCD-contiguous order (else grid.z = slice)
  int rlast, ndup;            // 8-phase kernel: the LAST slice is rows [M - rlast, M); its first ndup rows belong to the slice before (masked)
  const int* M_ptr;           // optional device-side row count <= M (8-phase kernel, nsplit > 1): the slices are re-cut on the device
};

// physical byte offset of logical 16-B chunk `c16` (8 columns) of row `row`
__device__ __forceinline__ int tn_off(int row, int c16) { return row * 256 + ((c16 ^ ((row & 3) << 2)) << 4); }

__device__ __forceinline__ void tn_stage_dma(const bf16* __restrict__ src, long ld, int m0, int col0, int ncols, char* tile, int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = c * 256 + tid;
    const int row = id >> 4, pc = id & 15;
    const int lc = pc ^ ((row & 3) << 2);
    int col = col0 + lc * 8;
    col = col < ncols ? col : ((ncols - 1) & ~7);      // chunks wholly past the last column re-read the last valid chunk (never stored)
    const bf16* g = src + (long)(m0 + row) * ld + col;
    const int wave_base = __builtin_amdgcn_readfirstlane((c * 256 + (tid & ~63)) * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(tile + wave_base), 16, 0, 0);
  }
}
__device__ __forceinline__ void tn_stage_regs(const bf16* __restrict__ src, long ld, int m0, int M, int col0, int ncols, char* tile, int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = c * 256 + tid;
    const int row = id >> 4, pc = id & 15;
    const int lc = pc ^ ((row & 3) << 2);
    int col = col0 + lc * 8;
    col = col < ncols ? col : ((ncols - 1) & ~7);
    bf16x8 v;
    if (m0 + row < M) {
      v = *(const bf16x8*)(src + (long)(m0 + row) * ld + col);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
    }
    *(bf16x8*)(tile + id * 16) = v;
  }
}

// LDS byte address this lane SUPPLIES for the transpose read of rows mb..mb+3 (h=0) / mb+4..mb+7 (h=1), columns colb..colb+15
__device__ __forceinline__ unsigned tn_addr(const char* tile, int mb, int colb, int i16, int h) {
  const int col = colb + (i16 & 3) * 4;
  const int r = mb + 4 * h + (i16 >> 2);
  return (unsigned)(size_t)(tile + tn_off(r, col >> 3) + (col & 7) * 2);
}
__device__ __forceinline__ bf16x8 tn_join(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// Fragments of one 16-row k-slice for this wave (two 32-column blocks of A and of B = 8 transpose reads), software
// pipelined: tn_issue() only ISSUES the reads; tn_wait<N>() waits until at most N LDS operations are outstanding (LDS
// returns in order, so N = 8 means "the older slice has landed") and names the destination registers so that no consumer
// is scheduled above it.  Lane i of a 16-lane group receives column i, 4 reduction rows per read.
struct TnFrag { bf16x4 r[8]; };
__device__ __forceinline__ void tn_issue(const char* As, const char* Bs, int mb, int ncol, int kcol, int i16, TnFrag& f) {
  const unsigned a00 = tn_addr(As, mb, ncol, i16, 0), a01 = tn_addr(As, mb, ncol, i16, 1);
  const unsigned a10 = tn_addr(As, mb, ncol + 32, i16, 0), a11 = tn_addr(As, mb, ncol + 32, i16, 1);
  const unsigned b00 = tn_addr(Bs, mb, kcol, i16, 0), b01 = tn_addr(Bs, mb, kcol, i16, 1);
  const unsigned b10 = tn_addr(Bs, mb, kcol + 32, i16, 0), b11 = tn_addr(Bs, mb, kcol + 32, i16, 1);
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15"
      : "=&v"(f.r[0]), "=&v"(f.r[1]), "=&v"(f.r[2]), "=&v"(f.r[3]), "=&v"(f.r[4]), "=&v"(f.r[5]), "=&v"(f.r[6]), "=&v"(f.r[7])
      : "v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(b00), "v"(b01), "v"(b10), "v"(b11)
      : "memory");
}
template <int N>
__device__ __forceinline__ void tn_wait(TnFrag& f) {
  if constexpr (N == 0)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]), "+v"(f.r[3]), "+v"(f.r[4]), "+v"(f.r[5]), "+v"(f.r[6]), "+v"(f.r[7]) :: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]), "+v"(f.r[3]), "+v"(f.r[4]), "+v"(f.r[5]), "+v"(f.r[6]), "+v"(f.r[7]) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <bool SLAB>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int i16 = lane & 15, j = lane >> 4;
  const int ntn = (p.N + BT - 1) / BT, ntk = (p.K + BT - 1) / BT, nt = ntn * ntk;
  int t;
  {
    const int b = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = b & 7, i = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int n0 = (t / ntk) * BT, k0 = (t % ntk) * BT;
  if (p.M_ptr) { const int m_ = *p.M_ptr; p.M = m_ < p.M ? m_ : p.M; }      // device-side row count: slices past it reduce nothing
  const int mbeg = blockIdx.z * p.rsplit;
  const int mend = min(p.M, mbeg + p.rsplit);
  const int nsteps = (mend - mbeg + BR - 1) / BR;

  f32x16 acc[2][2];   // [ki][ni]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto stage = [&](int s, char* buf) {
    const int m0 = mbeg + s * BR;
    if (m0 + BR <= p.M) {
      tn_stage_dma(p.A, p.lda, m0, n0, p.N, buf, tid);
      tn_stage_dma(p.B, p.ldb, m0, k0, p.K, buf + TILE_BYTES, tid);
    } else {
      tn_stage_regs(p.A, p.lda, m0, p.M, n0, p.N, buf, tid);
      tn_stage_regs(p.B, p.ldb, m0, p.M, k0, p.K, buf + TILE_BYTES, tid);
    }
  };
  if (nsteps > 0) stage(0, smem);
  for (int s = 0; s < nsteps; ++s) {
    __syncthreads();
    char* cur = smem + (s & 1) * STAGE_BYTES;
    if (s + 1 < nsteps) stage(s + 1, smem + ((s + 1) & 1) * STAGE_BYTES);
    const char* As = cur;
    const char* Bs = cur + TILE_BYTES;
    const int ncol = wn * 64 + (j & 1) * 16, kcol = wk * 64 + (j & 1) * 16, mg = (j >> 1) * 8;
    TnFrag fa, fb;
    tn_issue(As, Bs, mg, ncol, kcol, i16, fa);
#pragma unroll
    for (int kk = 0; kk < 4; kk += 2) {
      tn_issue(As, Bs, (kk + 1) * 16 + mg, ncol, kcol, i16, fb);
      tn_wait<8>(fa);
      {
        const bf16x8 a0 = tn_join(fa.r[0], fa.r[1]), a1 = tn_join(fa.r[2], fa.r[3]), b0 = tn_join(fa.r[4], fa.r[5]), b1 = tn_join(fa.r[6], fa.r[7]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 2 < 4) {
        tn_issue(As, Bs, (kk + 2) * 16 + mg, ncol, kcol, i16, fa);
        tn_wait<8>(fb);
      } else {
        tn_wait<0>(fb);
      }
      {
        const bf16x8 a0 = tn_join(fb.r[0], fb.r[1]), a1 = tn_join(fb.r[2], fb.r[3]), b0 = tn_join(fb.r[4], fb.r[5]), b1 = tn_join(fb.r[6], fb.r[7]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // acc[ki][ni][r] = D[k][n]: n = lane&31 (column), k = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = SLAB ? p.slab + (long)blockIdx.z * p.N * p.K : p.C;
  const long ldo = SLAB ? p.K : p.ldc;
#pragma unroll
  for (int ki = 0; ki < 2; ++ki)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
      if (n >= p.N) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = k0 + wk * 64 + ki * 32 + 8 * g + 4 * (lane >> 5);
        if (k >= p.K) continue;
        f32x4* dst = (f32x4*)(out + (long)n * ldo + k);
        f32x4 v = {acc[ki][ni][g * 4] * p.alpha, acc[ki][ni][g * 4 + 1] * p.alpha, acc[ki][ni][g * 4 + 2] * p.alpha,
                   acc[ki][ni][g * 4 + 3] * p.alpha};
        if constexpr (!SLAB) {
          const f32x4 o = *dst;
          v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
        *dst = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// p8: the weight-gradient GEMM on the 8-phase schedule of gemm.hip's NT kernel (same LDS ring, same phases, same counted
// waits; cdna_hip_programming.md "The 256^2 8-phase template").  Output tile 256 (n) x 256 (k) per 512-thread workgroup,
// 8 waves as 2 (n) x 4 (k); the reduction runs over the memory ROW index m in steps of 64 rows.
//  * A step stages FOUR 16-KiB half-tiles -- columns [0,128) and [128,256) of the A (dY) and of the B (X) row block, i.e. one
//    whole 256-B piece of every row -- one per phase, by LDS-DMA; the three newest stay in flight across the raw barriers.
//  * Wave (wr, wc) owns n-columns h*128 + wr*64 + [0,64) and k-columns h*128 + wc*32 + [0,32) of half h = 0, 1: its four
//    quadrants (hn, hk) are 64 x 32 outputs = 8 accumulators of v_mfma_f32_16x16x32_bf16, 16 MFMA per quadrant per step.
//  * Fragments come from ds_read_b64_tr_b16 (inline asm): lane i of a 16-lane group addresses 4 contiguous bf16 of row
//    mb + (i >> 2), columns cb + 4*(i & 3), and receives column cb + i of the four rows -- two reads = the 8 reduction values
//    of its k-slot.  LDS rows are 256 B; the 16-B chunk index is XOR-ed with ((row & 3) << 2) ^ (((row >> 3) & 1) << 1): the
//    32 lanes of a service group (rows mb..mb+3 and mb+8..mb+11 of one 16-column block) then touch all 64 banks once.
//  * Every split covers a multiple of 128 rows (an even number of steps); the launcher gives the < 128 leftover rows of M to the
//    128x128 kernel (LDS-DMA cannot zero-fill a ragged last step).
// Output: D[k][n] per MFMA, a lane owns 4 consecutive k of one n -> 16-B fp32 stores into the split's slab (or += C).
constexpr int TP_HT = 64 * 128 * 2;               // half-tile: 64 rows x 128 columns = 16 KiB
constexpr int TP_BUF = 4 * TP_HT;                 // one step: A-h0 | A-h1 | B-h0 | B-h1 = 64 KiB
constexpr int TP_LDS = 2 * TP_BUF;                // 128 KiB

template <int OFF>
__device__ __forceinline__ void tp_tr(bf16x4& d, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

// (the body of the 8-phase kernel as a function of the problem and the workgroup's index in it)
template <bool SLAB>
__device__ __forceinline__ void tn_p8_body(const TnP& p, const int bidx, const int bz) {
  extern __shared__ __attribute__((aligned(16))) char smem_tp[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (p.N + 255) / 256, ntk = (p.K + 255) / 256;
  // Which tile and which row slice?  Every tile re-reads the A panel of its n-tile and the B panel of its k-tile for its slice of
  // rows, so the workgroups that share panels must share an L2.  Workgroup b runs on XCD b % 8 (round-robin dispatch): each XCD is
  // given a CONTIGUOUS range of the (slice-major, then row-major tile) order, i.e. its ~32 resident workgroups are one slice's
  // consecutive tiles -- e.g. 32 tiles of a 12 x 3 grid touch 11 + 3 panels instead of 64.  (With the tiles of a slice dealt
  // round-robin over the eight XCDs every panel crossed the fabric once per tile: 25-30 % of the launch, GEMM_BENCH_LDA0 in
  // tools/gemm_bench.)
  int tile, zsplit;
  if (p.nsplit > 0) {
    const int tiles = ntn * ntk, G = tiles * p.nsplit;
    const int b = bidx, q = G >> 3, r = G & 7, xcd = b & 7, i = b >> 3;
    if (i >= (xcd < r ? q + 1 : q)) return;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    zsplit = L / tiles;
    tile = L - zsplit * tiles;
  } else {
    tile = bidx; zsplit = bz;
  }
  const int n0 = (tile / ntk) * 256, k0 = (tile % ntk) * 256;
  // Device-side row count (rows of a batch whose tail only the device knows): the launcher's slicing rule applied here to *M_ptr, with
  // the nsplit slices the grid was sized for; slices left without rows write a zero tile into their slab.
  int Mrows = p.M, rsplit = p.rsplit, rlast_ = p.rlast, ndup_ = p.ndup, nslices = p.nsplit;
  int mreal = p.M;                                   // rows that really exist: below 256 the slicing runs over 256 rows and the rest is masked (TP_MASK_TAIL)
  if (p.M_ptr != nullptr && p.nsplit > 1) {
    int m_ = *p.M_ptr;
    m_ = m_ < 0 ? 0 : m_;
    mreal = m_ < p.M ? m_ : p.M;
    m_ = m_ < p.M ? (m_ > 256 ? m_ : 256) : p.M;
    const int rs = ((((m_ + 127) >> 7) + p.nsplit - 1) / p.nsplit) << 7;
    const int ns = (m_ + rs - 1) / rs;               // >= 2 for m_ > rs; rows [0, m_) cut like the launcher cuts [0, M)
    const int over = ns * rs - m_;
    Mrows = m_; rsplit = rs; nslices = ns; rlast_ = rs - ((over >> 7) << 7); ndup_ = over & 127;
    if (ns == 1) { rlast_ = (m_ >> 7) << 7; ndup_ = 0; }      // (cannot happen for m_ > 256 with nsplit >= 2; kept total)
  }
  const bool empty = p.nsplit > 0 && zsplit >= nslices;
  // Row slices are multiples of 128 rows (pairs of 64-row steps).  M is not: the last slice is moved back to END at row M, and the
  // ndup rows it then shares with the slice before are zeroed in LDS (A side) after they land -- LDS-DMA cannot zero-fill, and
  // reading past row M is not an option.
  const bool lastz = p.nsplit > 0 && zsplit == nslices - 1;
  const int mbeg = empty ? 0 : (lastz ? Mrows - rlast_ : zsplit * rsplit);
  const int nk = empty ? 0 : (lastz ? rlast_ : min(Mrows, mbeg + rsplit) - mbeg) / 64;     // steps of 64 rows: even and >= 2 (0: a slice without rows)
  const int ndup = lastz ? ndup_ : 0;
  (void)ntn;

  f32x4 acc[2][4][4];                                  // [n half][n block mi][k half * 2 + k block]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- LDS-DMA sources.  DMA instruction i (0/1) of a half-tile: chunk id = i*512 + tid -> row id >> 4 (0..63), physical chunk
  // id & 15, logical chunk (8 columns) = physical ^ swizzle(row).  32-bit byte offsets from (operand + step * 64 rows); chunks
  // wholly past the last column re-read the last valid chunk (they feed outputs that are never stored).
  uint32_t offA[2][2], offB[2][2];                     // [half][instruction]
  {
    const int pc = tid & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 32 + (tid >> 4);
      const int lc = pc ^ ((row & 3) << 2) ^ (((row >> 3) & 1) << 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int ca = n0 + h * 128 + lc * 8, cb = k0 + h * 128 + lc * 8;
        ca = ca + 8 <= p.N ? ca : ((p.N - 8) & ~7);
        cb = cb + 8 <= p.K ? cb : ((p.K - 8) & ~7);
        offA[h][i] = (uint32_t)(mbeg + row) * (uint32_t)(p.lda * 2) + (uint32_t)(ca * 2);
        offB[h][i] = (uint32_t)(mbeg + row) * (uint32_t)(p.ldb * 2) + (uint32_t)(cb * 2);
      }
    }
  }
  const char* gA = (const char*)p.A;
  const char* gB = (const char*)p.B;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem_tp;
  // LDS-DMA in its SGPR-base + 32-bit-VGPR-offset form (inline asm: the builtin materialises a 64-bit address per load -- 16
  // VGPRs of loop-invariant pointers here -- and its loads are tracked by the compiler's waitcnt pass, which this schedule
  // counts by hand anyway).  M0 = wave-uniform LDS destination; one wait state between the M0 write and its use.
#define TP_STG(GB, O, SLOT)                                                                                               \
  do {                                                                                                                    \
    const uint32_t d_ = lds0 + (SLOT) + wave * 1024;                                                                      \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((O)[0]), "s"(GB), "s"(d_) : "memory", "m0");          \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"((O)[1]), "s"(GB), "s"(d_ + 8192) : "memory", "m0");   \
  } while (0)
#define TP_STG_A(H, BUF, KT) TP_STG((const char*)(gA + (size_t)(KT) * 128 * p.lda), offA[H], (BUF) * TP_BUF + (H) * TP_HT)
#define TP_STG_B(H, BUF, KT) TP_STG((const char*)(gB + (size_t)(KT) * 128 * p.ldb), offB[H], (BUF) * TP_BUF + (2 + (H)) * TP_HT)

  // ---- fragment addresses.  16x16x32 operand: lane l -> column (l & 15) of the block, reduction rows kk*32 + (l >> 4)*8 .. +7,
  // fetched as two transpose reads of 4 rows.  This lane SUPPLIES the address of row mb + 4*rh + (i >> 2), columns cb + 4*(i & 3).
  uint32_t aad[2][2];                                  // [kk][row half]: byte address inside an A half-tile for column block 0 of the wave
  {
    const int i16 = lane & 15, j = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        const int row = kk * 32 + j * 8 + rh * 4 + (i16 >> 2);
        const int sw = ((row & 3) << 2) ^ (((row >> 3) & 1) << 1);
        const int ca = wr * 64 + (i16 & 3) * 4;        // column inside the half-tile (block f adds 16)
        aad[kk][rh] = lds0 + (uint32_t)(row * 256 + (((ca >> 3) ^ sw) << 4) + (ca & 7) * 2);
      }
  }
  // the B address of the same lane differs only in the wave's base chunk (wc*4 instead of wr*8): a wave-uniform XOR
  const uint32_t bxor = (uint32_t)(((wr * 8) ^ (wc * 4)) << 4);
  // column block f of the wave's run: +16 columns = chunk index + 2f; the wave's base chunk is a multiple of 4 (B) / 8 (A) and the
  // swizzle touches bits 1-3, so the block index enters the address as XOR (f << 5), not as an immediate
  // (`opq` is an always-zero value the compiler cannot see through, refreshed before every group of reads: without it the 48
  // XOR-ed addresses of both buffers are hoisted out of the loop and spill)
  uint32_t opq = 0;
  bf16x4 ta[4][2][2], tb0[2][2][2], tb1[2][2][2];     // [block][kk][row half]
#define TP_RD_A(BUF, H)                                                                                              \
  do {                                                                                                               \
    asm volatile("" : "+v"(opq));                                                                                    \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_) {                                                            \
      _Pragma("unroll") for (int f_ = 0; f_ < 4; ++f_)                                                               \
      _Pragma("unroll") for (int rh_ = 0; rh_ < 2; ++rh_)                                                            \
        tp_tr<(H) * TP_HT>(ta[f_][kk_][rh_], ((aad[kk_][rh_] ^ (uint32_t)(f_ << 5)) ^ opq) + (BUF) * TP_BUF);       \
      __builtin_amdgcn_sched_barrier(0);           /* at most 8 address temporaries alive at a time */               \
    }                                                                                                                \
  } while (0)
#define TP_RD_B(BUF, H, TB)                                                                                          \
  do {                                                                                                               \
    asm volatile("" : "+v"(opq));                                                                                    \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                                              \
    _Pragma("unroll") for (int f_ = 0; f_ < 2; ++f_)                                                                 \
    _Pragma("unroll") for (int rh_ = 0; rh_ < 2; ++rh_)                                                              \
      tp_tr<(2 + (H)) * TP_HT>(TB[f_][kk_][rh_], ((aad[kk_][rh_] ^ bxor ^ (uint32_t)(f_ << 5)) ^ opq) + (BUF) * TP_BUF); \
  } while (0)
#define TP_MM(H, KH, TB)                                                                                             \
  do {                                                                                                               \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                                                              \
    _Pragma("unroll") for (int mi_ = 0; mi_ < 4; ++mi_)                                                              \
    _Pragma("unroll") for (int ni_ = 0; ni_ < 2; ++ni_)                                                              \
      acc[H][mi_][(KH) * 2 + ni_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                        \
          tn_join(TB[ni_][kk_][0], TB[ni_][kk_][1]), tn_join(ta[mi_][kk_][0], ta[mi_][kk_][1]), acc[H][mi_][(KH) * 2 + ni_], 0, 0, 0); \
  } while (0)
#define TP_BAR() __builtin_amdgcn_s_barrier()
#define TP_SB() __builtin_amdgcn_sched_barrier(0)
#define TP_COMPUTE(H, KH, TB)                              \
  do {                                                     \
    TP_BAR();                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    TP_SB();                                               \
    __builtin_amdgcn_s_setprio(1);                         \
    TP_MM(H, KH, TB);                                      \
    __builtin_amdgcn_s_setprio(0);                         \
    TP_SB();                                               \
    TP_BAR();                                              \
    TP_SB();                                               \
  } while (0)

  if (nk > 0) {                                         // (wave- and workgroup-uniform)
  // ---- prologue: step 0 complete, the first three half-tiles of step 1 in flight
  TP_STG_A(0, 0, 0); TP_STG_B(0, 0, 0); TP_STG_B(1, 0, 0); TP_STG_A(1, 0, 0);
  TP_STG_B(0, 1, 1); TP_STG_A(0, 1, 1); TP_STG_B(1, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  // rows of step S (buffer S & 1) this thread staged: i * 32 + (tid >> 4) for DMA instruction i, both column halves of A
#define TP_MASK(S)                                                                                                   \
  do {                                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                 \
      if ((S) * 64 + i_ * 32 + (tid >> 4) < ndup) {                                                                  \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                             \
          *(LDS_AS f32x4*)(uintptr_t)(lds0 + ((S) & 1) * TP_BUF + h_ * TP_HT + wave * 1024 + lane * 16 + i_ * 8192) = \
              f32x4{0.f, 0.f, 0.f, 0.f};                                                                             \
      }                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
  } while (0)
  // Fewer than 256 real rows under a device-side row count: the two 128-row slices the grid then runs hold rows past the count, whose
  // contents nobody wrote (NaN bit patterns included: 0 x NaN would poison the product), so BOTH operands' rows are zeroed after they land.
#define TP_MASK_TAIL(S)                                                                                              \
  do {                                                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                 \
      if (mbeg + (S) * 64 + i_ * 32 + (tid >> 4) >= mreal) {                                                         \
        _Pragma("unroll") for (int h_ = 0; h_ < 4; ++h_)                                                             \
          *(LDS_AS f32x4*)(uintptr_t)(lds0 + ((S) & 1) * TP_BUF + h_ * TP_HT + wave * 1024 + lane * 16 + i_ * 8192) = \
              f32x4{0.f, 0.f, 0.f, 0.f};                                                                             \
      }                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
  } while (0)
  const bool tail = mreal < Mrows;                     // (workgroup-uniform; implies Mrows == 256: two slices of two steps)
  if (ndup > 0) TP_MASK(0);                            // (wave-uniform; step 0 has landed: this thread's own pieces)
  if (tail) TP_MASK_TAIL(0);
  TP_BAR();
  TP_SB();
  if (wr == 1) TP_BAR();                               // the second wave row runs one barrier behind the first

  for (int kt = 0; kt < nk; kt += 2) {
    const bool more = kt + 2 < nk;
    // ---------------- step kt (buffer 0).  Hazards exactly as in gemm_nt_p8_kernel: B-h0 (8 reads, issued first, retired by the
    // lgkmcnt before the first barrier) is restaged one phase later, everything else two phases after its last read.
    TP_RD_B(0, 0, tb0); TP_SB(); TP_RD_A(0, 0);                                   // phase 1: 8 + 16 transpose reads
    TP_STG_A(1, 1, kt + 1);
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");                           // 24 issued, <= 15 pending: the 8 B-h0 reads are back
    TP_COMPUTE(0, 0, tb0);
    TP_RD_B(0, 1, tb1);                                                           // phase 2
    if (more) TP_STG_B(0, 0, kt + 2);
    TP_COMPUTE(0, 1, tb1);
    TP_RD_A(0, 1);                                                                // phase 3
    if (more) TP_STG_A(0, 0, kt + 2);
    TP_COMPUTE(1, 1, tb1);
    if (more) {                                                                   // phase 4
      TP_STG_B(1, 0, kt + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (kt == 0 && ndup > 64) TP_MASK(1);                                         // step 1 has landed; it is read from phase 5 on
    if (kt == 0 && tail) {                                                        // (nk == 2 here: `more` is false, vmcnt(0) above)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TP_MASK_TAIL(1);
    }
    TP_COMPUTE(1, 0, tb0);
    // ---------------- step kt+1 (buffer 1)
    TP_RD_B(1, 0, tb0); TP_SB(); TP_RD_A(1, 0);                                   // phase 5
    if (more) TP_STG_A(1, 0, kt + 2);
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    TP_COMPUTE(0, 0, tb0);
    TP_RD_B(1, 1, tb1);                                                           // phase 6
    if (more) TP_STG_B(0, 1, kt + 3);
    TP_COMPUTE(0, 1, tb1);
    TP_RD_A(1, 1);                                                                // phase 7
    if (more) TP_STG_A(0, 1, kt + 3);
    TP_COMPUTE(1, 1, tb1);
    if (more) {                                                                   // phase 8
      TP_STG_B(1, 1, kt + 3);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    TP_COMPUTE(1, 0, tb0);
  }
  if (wr == 0) TP_BAR();
  }
#undef TP_MASK
#undef TP_STG
#undef TP_STG_A
#undef TP_STG_B
#undef TP_RD_A
#undef TP_RD_B
#undef TP_MM
#undef TP_COMPUTE
#undef TP_BAR
#undef TP_SB

  // ---- epilogue: acc[h][mi][kh*2+ni][j] = D[k][n], n = n0 + h*128 + wr*64 + mi*16 + (lane & 15),
  //      k = k0 + kh*128 + wc*32 + ni*16 + 4*(lane >> 4) + j
  // Straight from the accumulators a store instruction would write 16 rows x 64 B (12 B/clk/CU, tools/store_bench: 22 k cycles
  // for the 256-KiB tile); the ring is free now, so each quarter (64 output rows x 256 columns) is transposed through LDS
  // (row pitch 260 floats: the 16 rows of a lane group land in disjoint banks) and leaves as whole 1-KiB rows, 8 rows per wave.
  float* out = SLAB ? p.slab + (long)zsplit * p.N * p.K : p.C;
  const long ldo = SLAB ? p.K : p.ldc;
  float* tl = (float*)smem_tp;
  constexpr int TPITCH = 260;
  const int kcol = k0 + lane * 4;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      __syncthreads();                                   // the ring (first pass) / the previous quarter's rows have been read
      if (wr == r) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = acc[h][mi][q];
            v[0] *= p.alpha; v[1] *= p.alpha; v[2] *= p.alpha; v[3] *= p.alpha;
            *(f32x4*)(tl + (mi * 16 + (lane & 15)) * TPITCH + (q >> 1) * 128 + wc * 32 + (q & 1) * 16 + 4 * (lane >> 4)) = v;
          }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i;                    // row of the quarter
        const int n = n0 + h * 128 + r * 64 + row;
        if (n < p.N && kcol < p.K) {
          f32x4 v = *(const f32x4*)(tl + row * TPITCH + lane * 4);
          f32x4* dst = (f32x4*)(out + (long)n * ldo + kcol);
          if constexpr (!SLAB) {
            const f32x4 o = *dst;
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
          }
          *dst = v;
        }
      }
    }
}

template <bool SLAB>
__global__ __launch_bounds__(512) void gemm_tn_p8_kernel(TnP p) { tn_p8_body<SLAB>(p, blockIdx.x, blockIdx.z); }

// C[n*ldc + k] += sum_z slab[z][n][k]
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int splits, int N, int K4, float* __restrict__ C,
                                                          long ldc) {
  const long total = (long)N * K4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / K4), k = (int)(i - (long)n * K4) * 4;
    f32x4 s = *(const f32x4*)(slab + i * 4);
    for (int z = 1; z < splits; ++z) {
      const f32x4 v = *(const f32x4*)(slab + (long)z * total * 4 + i * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    f32x4* d = (f32x4*)(C + (long)n * ldc + k);
    f32x4 o = *d;
    o[0] += s[0]; o[1] += s[1]; o[2] += s[2]; o[3] += s[3];
    *d = o;
  }
}

// out[c] += sum_r x[r][c]   (bias gradients).  A workgroup reduces a CS_ROWS-row x 128-column strip: 16 column groups of
// 8 bf16 (16-byte loads) x 16 row lanes, eight loads in flight per thread, LDS tree over the row lanes, one atomicAdd per column.
constexpr int CS_ROWS = 256;    // (1024-row strips were measured SLOWER: 37 vs 26 us per launch -- fewer workgroups, less memory-level parallelism)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, long ld, int R, int C, float* __restrict__ out,
                                                     const int* __restrict__ R_ptr) {
  __shared__ float red[16][129];
  if (R_ptr) { const int r_ = *R_ptr; R = r_ < R ? r_ : R; }
  if ((int)blockIdx.y * CS_ROWS >= R) return;
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 128 + cg * 8;
  const int r0 = blockIdx.y * CS_ROWS;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const bool full = c + 8 <= C;
    if (full && r0 + CS_ROWS <= R) {
      const bf16* xp = x + (long)(r0 + rl) * ld + c;
      for (int i = 0; i < CS_ROWS / 16; i += 8) {
        bf16x8 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const bf16x8*)(xp + (long)(i + u) * 16 * ld);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += (float)v[u][e];
      }
    } else
#pragma unroll 4
    for (int i = 0; i < CS_ROWS / 16; ++i) {
      const int r = r0 + rl + i * 16;
      if (r < R) {
        if (full) {
          const bf16x8 v = *(const bf16x8*)(x + (long)r * ld + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        } else {
          for (int e = 0; e < 8 && c + e < C; ++e) s[e] += (float)x[(long)r * ld + c + e];
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int cc = blockIdx.x * 128 + threadIdx.x;
    if (cc < C) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
      atomicAdd(out + cc, t);
    }
  }
}

}  // namespace

extern "C" long spmm_gemm_tn_workspace_bytes(int M, int N, int K, int splits) {
  (void)M;
  return splits > 1 ? (long)splits * N * K * 4 : 0;
}

// target number of workgroups (tiles x splits) of the 128x128 kernel: ~2 full rounds of the 2 x 256 resident workgroups (measured:
// 640 -> 1000 = +20..45 % on the training step's shapes)
constexpr int TN_TARGET_WGS = 1000;

// kernel: 0 = chosen from the shape, 1 = 128x128 tiles (two workgroups per CU), 8 = 256x256 tiles on the 8-phase schedule (one
// workgroup per CU; long reductions, N and K multiples of 8)
static int tn_pick(int M, int N, int K, int kernel) {
  if (kernel == 1 || kernel == 8) return kernel;
  const bool ok8 = N % 8 == 0 && K % 8 == 0 && N >= 256 && K >= 256;
  const long tiles8 = (long)((N + 255) / 256) * ((K + 255) / 256);
  // the big tile needs enough reduction rows per workgroup to amortise its 256-KiB fp32 tile store and 14-load prologue: measured
  // cross-over (tools/gemm_bench tntime, k1 vs k8) between M x tiles = 124 k (6912 x 18, 13824 x 9: 128x128 wins) and 147 k (16384 x 9)
  return (ok8 && M >= 4096 && (long)M * tiles8 >= 136L * 1024) ? 8 : 1;
}

extern "C" int spmm_gemm_tn_splits(int M, int N, int K, int kernel) {
  if (tn_pick(M, N, K, kernel) == 8) {
    const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
    int s = 256 / tiles;                              // one workgroup per CU
    const int maxs = M / 1024 > 0 ? M / 1024 : 1;     // >= 16 steps per split
    if (s > maxs) s = maxs;
    return s < 1 ? 1 : s;
  }
  const int tiles = ((N + BT - 1) / BT) * ((K + BT - 1) / BT);
  int s = (TN_TARGET_WGS + tiles / 2) / tiles;
  const int maxs = (M / BR) / 16 > 0 ? (M / BR) / 16 : 1;   // at least 16 reduction steps per split
  if (s > maxs) s = maxs;
  return s < 1 ? 1 : s;
}

static void tn_reduce_launch(const float* ws, int ns, int N, int K, float* C, long ldc, hipStream_t stream) {
  long blocks = ((long)N * (K / 4) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, ns, N, K / 4, C, ldc);
}

static int tn_run(const void* A, long lda, const void* B, long ldb, int M, int N, int K, int splits, float alpha, float* C, long ldc,
                  float* workspace, int kernel, const int* M_dev, hipStream_t stream) {
  SPMM_CHECK_SHAPE(M > 0 && N > 0 && K > 0, "spmm_gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
  SPMM_CHECK_SHAPE(N % 4 == 0 && K % 4 == 0 && ldc % 4 == 0, "spmm_gemm_tn: N=%d K=%d ldc=%ld must be multiples of 4", N, K, ldc);
#ifndef P8_PROFILE   // (the profiling build of tools/ aliases all rows onto row 0 with lda = ldb = 0: cache-resident operands)
  SPMM_CHECK_SHAPE(lda % 8 == 0 && ldb % 8 == 0 && lda >= ((N + 7) & ~7) && ldb >= ((K + 7) & ~7),
                   "spmm_gemm_tn: lda=%ld / ldb=%ld must be multiples of 8 covering the 8-column chunks of N=%d / K=%d", lda, ldb, N, K);
#endif
  SPMM_CHECK_SHAPE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0), "spmm_gemm_tn: A/B must be 16-B aligned");
  SPMM_CHECK_SHAPE(kernel == 0 || kernel == 1 || kernel == 8, "spmm_gemm_tn: unknown kernel selector %d", kernel);
  SPMM_CHECK_SHAPE(kernel != 8 || (N % 8 == 0 && K % 8 == 0 && N >= 8 && K >= 8), "spmm_gemm_tn: the 8-phase kernel needs N %% 8 == 0 and K %% 8 == 0");
  if (splits < 1) splits = 1;
  SPMM_CHECK_SHAPE(splits == 1 || workspace != nullptr, "spmm_gemm_tn: split reduction needs a workspace");
  const int k8 = tn_pick(M, N, K, kernel) == 8 && M >= 128;
  SPMM_CHECK_SHAPE(!k8 || ((unsigned long)M * (unsigned long)lda * 2ul < (1ul << 32) && (unsigned long)M * (unsigned long)ldb * 2ul < (1ul << 32)),
                   "spmm_gemm_tn: the 8-phase kernel addresses its operands with 32-bit byte offsets (< 4 GiB)");
  TnP p;
  p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.M = M; p.N = N; p.K = K;
  p.C = C; p.ldc = ldc; p.slab = workspace; p.alpha = alpha; p.nsplit = 0; p.M_ptr = nullptr;
  auto reduce = [&](int nsplit) { tn_reduce_launch(workspace, nsplit, N, K, C, ldc, stream); };
  if (k8) {
    static const hipError_t attr_rc = [] {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_p8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS);
      if (e != hipSuccess) return e;
      return hipFuncSetAttribute((const void*)gemm_tn_p8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS);
    }();
    if (attr_rc != hipSuccess) {
      spmm_set_error("spmm_gemm_tn: cannot raise dynamic LDS to %d: %s", TP_LDS, hipGetErrorString(attr_rc));
      return SPMM_ERR_LAUNCH;
    }
    // slices of rs rows (multiples of 128); the last one ends at row M and is masked where it overlaps the one before
    int rs = ((((M + 127) / 128) + splits - 1) / splits) * 128;
    const int ns = (M + rs - 1) / rs;
    const int over = ns * rs - M;                       // rows the uniform slicing would run past M
    p.rsplit = rs; p.nsplit = ns;
    p.rlast = rs - (over / 128) * 128;                  // whole 128-row pairs past M are dropped, the rest is overlap
    p.ndup = over % 128;
    if (ns == 1) { p.rlast = ((M + 127) / 128) * 128 > M ? (M / 128) * 128 : M; p.ndup = 0; }   // a single slice cannot overlap anything
    const int single_left = ns == 1 ? M - p.rlast : 0;
    dim3 grid(((N + 255) / 256) * ((K + 255) / 256) * ns, 1, 1);
    if (M_dev != nullptr && ns == 1) {                 // one slice + remainder pass: the 128x128 kernel serves a device-side row count at any M
      p.nsplit = 0; p.M_ptr = M_dev;
      p.rsplit = ((M + BR - 1) / BR) * BR;
      hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(((N + BT - 1) / BT) * ((K + BT - 1) / BT), 1, 1), dim3(256), 0, stream, p);
      SPMM_LAUNCH_CHECK("spmm_gemm_tn(device rows, 128x128)");
      return SPMM_OK;
    }
    p.M_ptr = M_dev;
    if (ns > 1) {
      hipLaunchKernelGGL(gemm_tn_p8_kernel<true>, grid, dim3(512), TP_LDS, stream, p);
      reduce(ns);
    } else {
      hipLaunchKernelGGL(gemm_tn_p8_kernel<false>, grid, dim3(512), TP_LDS, stream, p);
    }
    if (single_left > 0) {                              // one slice = rows [M % 128, M); the first M % 128 rows: one pass of the 128x128 kernel, accumulated into C
      TnP q = p;
      q.nsplit = 0;
      q.M = single_left; q.rsplit = ((single_left + BR - 1) / BR) * BR;
      hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(((N + BT - 1) / BT) * ((K + BT - 1) / BT), 1, 1), dim3(256), 0, stream, q);
    }
    SPMM_LAUNCH_CHECK("spmm_gemm_tn(8-phase)");
    return SPMM_OK;
  }
  p.M_ptr = M_dev;
  int rsplit = (((M + BR - 1) / BR + splits - 1) / splits) * BR;
  splits = (M + rsplit - 1) / rsplit;
  p.rsplit = rsplit;
  const int tiles = ((N + BT - 1) / BT) * ((K + BT - 1) / BT);
  dim3 grid(tiles, 1, splits);
  if (splits > 1) {
    hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, stream, p);
    reduce(splits);
  } else {
    hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, stream, p);
  }
  SPMM_LAUNCH_CHECK("spmm_gemm_tn");
  return SPMM_OK;
}

extern "C" int spmm_gemm_tn(const void* A, long lda, const void* B, long ldb, int M, int N, int K, int splits, float alpha,
                            float* C, long ldc, float* workspace, int kernel, const int* M_dev, spmm_stream_t stream) {
  return tn_run(A, lda, B, ldb, M, N, K, splits, alpha, C, ldc, workspace, kernel, M_dev, stream);
}

extern "C" int spmm_gemm_tn_reduce(const float* ws, int ns, int N, int K, float* C, long ldc, spmm_stream_t stream) {
  SPMM_CHECK_SHAPE(ws != nullptr && C != nullptr && ns >= 1 && N > 0 && K > 0 && K % 4 == 0 && ldc % 4 == 0,
                   "spmm_gemm_tn_reduce: ns=%d N=%d K=%d ldc=%ld", ns, N, K, ldc);
  tn_reduce_launch(ws, ns, N, K, C, ldc, stream);
  SPMM_LAUNCH_CHECK("spmm_gemm_tn_reduce");
  return SPMM_OK;
}

extern "C" int spmm_colsum_bf16(const void* x, long ld, int R, int C, float* out, const int* R_dev, spmm_stream_t stream) {
  SPMM_CHECK_SHAPE(R > 0 && C > 0 && ld % 8 == 0 && ((uintptr_t)x % 16 == 0), "spmm_colsum_bf16: R=%d C=%d ld=%ld (ld %% 8, 16-B aligned)", R, C, ld);
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 127) / 128, (R + CS_ROWS - 1) / CS_ROWS), dim3(256), 0, stream, (const bf16*)x, ld, R, C, out, R_dev);
  SPMM_LAUNCH_CHECK("spmm_colsum_bf16");
  return SPMM_OK;
}
