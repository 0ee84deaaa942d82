// Tile helpers shared by the attention kernels (attention.hip) and the fused cross-attention block (xattn.hip):
// [L][64] bf16 head tiles in LDS (XOR-swizzled 16-B chunks), LDS-DMA staging, transpose reads, MFMA 32x32x16 wrappers.
#pragma once
#include "common.h"

namespace {

constexpr int HD = 64;            // head dim
constexpr int ROWB = 128;         // bytes per row of a row-major [L][64] bf16 LDS tile
constexpr int TILE = 128 * ROWB;  // 16 KiB

// Row-major [128][64] bf16 tile, 16-B chunk index XOR-swizzled with rotr3((row>>1)&7):
//  * ds_read_b128 of 16 rows (distinct mod 16) at one k-slot -> 16 distinct 16-B slots of the 256-B bank window;
//  * ds_read_b64_tr_b16 of 4 rows x 64 B per 32-lane half -> rows r,r+1 sit in different window halves and rows r,r+2 in
//    different 64-B spans (bit 2 of the chunk index flips with bit 1 of the row): conflict-free as well.
__device__ __forceinline__ int frot(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int swz(int row, int c16) { return row * ROWB + ((c16 ^ frot(row)) << 4); }

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// reference mask arithmetic: (1 - causal*mask) * -10000 (self) / (1 - mask) * finfo.min (cross)
__device__ __forceinline__ float score_bias(int mask_kv, bool causal, int q, int kv, float mask_neg) {
  const bool ok = mask_kv && (!causal || kv <= q);
  return ok ? 0.f : mask_neg;
}

// LDS-DMA staging of the [L][64] head slice: global_load_lds_dwordx4 writes lane-linear, so the swizzle is applied on
// the SOURCE chunk.  Rows >= L re-read row L-1 (finite data; such rows are masked / never stored).
__device__ __forceinline__ void stage_head(const bf16* __restrict__ src, long ld, int L, char* tile, int tid, int nthreads,
                                           int rows = 128) {
  // only the 32-row tiles the MFMAs will touch are staged (`rows` = tiles * 32): a 54-token sequence moves 64 rows, not 128
  for (int id = tid; id < rows * 8; id += nthreads) {
    const int row = id >> 3, pc = id & 7;
    const int lc = pc ^ frot(row);
    const int grow = row < L ? row : L - 1;
    const bf16* g = src + (long)grow * ld + lc * 8;
    const int wave_base = __builtin_amdgcn_readfirstlane((id & ~63) * 16);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(tile + wave_base), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 ld_rm(const char* tile, int row, int slot) { return *(const bf16x8*)(tile + swz(row, slot)); }

__device__ __forceinline__ bf16x8 join8(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// A-operand fragments X^T[d][rows] for both 32-wide d blocks, straight from the row-major tile with the LDS transpose read:
// lane l (d = dt*32 + (l&31), g = l>>5) receives X[rb + 4g + {0..3, 8..11}][d] -- the k-slot order the probabilities use.
__device__ __forceinline__ void ld_tr2(const char* tile, int rb, int lane, bf16x8 (&out)[2]) {
  const int i16 = lane & 15, j = lane >> 4;
  const int row0 = rb + 4 * (j >> 1) + (i16 >> 2);
  const int col = (j & 1) * 16 + (i16 & 3) * 4;
  const unsigned a00 = (unsigned)(size_t)(tile + swz(row0, col >> 3) + (col & 7) * 2);
  const unsigned a01 = (unsigned)(size_t)(tile + swz(row0 + 8, col >> 3) + (col & 7) * 2);
  const unsigned a10 = (unsigned)(size_t)(tile + swz(row0, (col + 32) >> 3) + (col & 7) * 2);
  const unsigned a11 = (unsigned)(size_t)(tile + swz(row0 + 8, (col + 32) >> 3) + (col & 7) * 2);
  bf16x4 r0, r1, r2, r3;
  asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a00), "v"(a01), "v"(a10), "v"(a11) : "memory");
  out[0] = join8(r0, r1);
  out[1] = join8(r2, r3);
  __builtin_amdgcn_sched_barrier(0);
}
// the same for TWO 16-row k-blocks (rb, rb + 16) with a single wait: eight transpose reads in flight instead of four
__device__ __forceinline__ void ld_tr2x2(const char* tile, int rb, int lane, bf16x8 (&o0)[2], bf16x8 (&o1)[2]) {
  const int i16 = lane & 15, j = lane >> 4;
  const int row0 = rb + 4 * (j >> 1) + (i16 >> 2);
  const int col = (j & 1) * 16 + (i16 & 3) * 4;
  unsigned a[8];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    a[b * 4 + 0] = (unsigned)(size_t)(tile + swz(row0 + 16 * b, col >> 3) + (col & 7) * 2);
    a[b * 4 + 1] = (unsigned)(size_t)(tile + swz(row0 + 16 * b + 8, col >> 3) + (col & 7) * 2);
    a[b * 4 + 2] = (unsigned)(size_t)(tile + swz(row0 + 16 * b, (col + 32) >> 3) + (col & 7) * 2);
    a[b * 4 + 3] = (unsigned)(size_t)(tile + swz(row0 + 16 * b + 8, (col + 32) >> 3) + (col & 7) * 2);
  }
  bf16x4 r[8];
  asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\tds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
               "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\tds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
  o0[0] = join8(r[0], r[1]); o0[1] = join8(r[2], r[3]);
  o1[0] = join8(r[4], r[5]); o1[1] = join8(r[6], r[7]);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int hf) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (bf16)v[hf * 8 + e];
  return r;
}
// two floats -> one register of two bf16 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pk2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v; v[0] = (bf16)a; v[1] = (bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

}  // namespace
