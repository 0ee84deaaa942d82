// Standalone check + timing of spmm_gemm_nt's tile kernels through the C ABI (no Python, no torch: starts in a second on a
// fresh GPU box).  Build: make -C tools  ->  build/gemm_bench.   Usage:
//   gemm_bench check                       every kernel vs a host fp64 reference on sampled rows, ragged and full shapes
//   gemm_bench time M N K [epi] [rounds]   interleaved rounds of kernels 3 (256x256, one barrier per k-step) and 8 (8-phase)
// Operands are uniform random in [-1, 1) (cdna_hip_programming.md 5.4 rule 25: never time on zeros).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <math.h>
#include <vector>
#include <array>
#include <algorithm>
#include "../include/spmm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float urand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (float)((rng_state >> 40) & 0xffffff) / 8388608.0f - 1.0f; }

struct Buf {
  void* d = nullptr; size_t bytes = 0;
  void alloc(size_t b) { bytes = b; CK(hipMalloc(&d, b)); }
  ~Buf() { if (d) (void)hipFree(d); }
};
static std::vector<uint16_t> rand_bf16(size_t n, float scale) { std::vector<uint16_t> v(n); for (auto& x : v) x = f2bf(urand() * scale); return v; }

static int run(int kernel, int epi, const Buf& A, const Buf& W, int M, int N, int K, const float* bias, const Buf* R, const Buf* G, Buf& C, Buf* C2,
               float* colsum, hipStream_t st) {
  // GEMM_BENCH_LDA0=1: every row of A is row 0 (lda = 0): the A operand is cache-resident, what remains is the schedule itself
  // (an upper bound for what hiding the first-touch latency of A could buy)
  static const long lda = getenv("GEMM_BENCH_LDA0") ? 0 : -1;
  return spmm_gemm_nt(A.d, lda < 0 ? K : lda, W.d, K, M, N, K, 1, bias, nullptr, 1.0f, R ? R->d : nullptr, N, G ? G->d : nullptr, N, C.d, N, C2 ? C2->d : nullptr, N,
                      epi, colsum, kernel, nullptr, st);
}

static double gelu(double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752)); }
static double gelu_grad(double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752)) + x * 0.3989422804014327 * exp(-0.5 * x * x); }

static int check_one(int kernel, int epi, int M, int N, int K) {
  auto hA = rand_bf16((size_t)M * K, 1.0f), hW = rand_bf16((size_t)N * K, 0.05f), hR = rand_bf16((size_t)M * N, 1.0f), hG = rand_bf16((size_t)M * N, 1.0f);
  std::vector<float> hb(N);
  for (auto& x : hb) x = urand();
  Buf A, W, R, G, C, C2, bias, cs;
  A.alloc(hA.size() * 2); W.alloc(hW.size() * 2); R.alloc(hR.size() * 2); G.alloc(hG.size() * 2); C.alloc((size_t)M * N * 2); C2.alloc((size_t)M * N * 2);
  bias.alloc(N * 4); cs.alloc(N * 4);
  CK(hipMemcpy(A.d, hA.data(), A.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(W.d, hW.data(), W.bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(R.d, hR.data(), R.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(G.d, hG.data(), G.bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias.d, hb.data(), bias.bytes, hipMemcpyHostToDevice));
  CK(hipMemset(C.d, 0x7f, C.bytes)); CK(hipMemset(C2.d, 0x7f, C2.bytes)); CK(hipMemset(cs.d, 0, cs.bytes));
  const bool two = epi == SPMM_EPI_GELU || epi == SPMM_EPI_GELU_DERIV, needs_g = epi == SPMM_EPI_GELU_GRAD || epi == SPMM_EPI_MUL;
  const bool use_r = epi == SPMM_EPI_BF16, use_cs = !two;
  int rc = run(kernel, epi, A, W, M, N, K, (const float*)bias.d, use_r ? &R : nullptr, needs_g ? &G : nullptr, C,
               two ? &C2 : nullptr, use_cs ? (float*)cs.d : nullptr, 0);
  if (rc) { printf("  kernel %d epi %d %dx%dx%d: rc=%d %s\n", kernel, epi, M, N, K, rc, spmm_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> hC((size_t)M * N), hC2((size_t)M * N);
  std::vector<float> hcs(N);
  CK(hipMemcpy(hC.data(), C.d, C.bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hC2.data(), C2.d, C2.bytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hcs.data(), cs.d, cs.bytes, hipMemcpyDeviceToHost));
  // host reference on a sample of rows (all columns): first / last rows of every 256-row tile edge plus random rows
  std::vector<int> rows;
  for (int r : {0, 1, 15, 16, 63, 64, 127, 128, 129, 191, 192, 255, 256, 257}) if (r < M) rows.push_back(r);
  for (int r = M - 3; r < M; ++r) if (r >= 0) rows.push_back(r);
  for (int i = 0; i < 40; ++i) rows.push_back((int)(fabsf(urand()) * (M - 1)));
  std::sort(rows.begin(), rows.end()); rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
  double maxerr = 0; long bad = 0;
  for (int m : rows)
    for (int n = 0; n < N; ++n) {
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hW[(size_t)n * K + k]);
      acc += hb[n];
      double ref = acc, ref2 = 0; bool has2 = false;
      if (epi == SPMM_EPI_BF16) ref = (double)bf2f(f2bf((float)acc)) + bf2f(hR[(size_t)m * N + n]);   // the kernel rounds to bf16 before adding R
      else if (epi == SPMM_EPI_GELU) { ref = gelu(acc); ref2 = acc; has2 = true; }
      else if (epi == SPMM_EPI_GELU_GRAD) ref = acc * gelu_grad((double)bf2f(hG[(size_t)m * N + n]));
      else if (epi == SPMM_EPI_GELU_DERIV) { ref = gelu(acc); ref2 = gelu_grad(acc); has2 = true; }
      else if (epi == SPMM_EPI_MUL) ref = (double)bf2f(f2bf((float)acc)) * bf2f(hG[(size_t)m * N + n]);
      const double got = bf2f(hC[(size_t)m * N + n]);
      double err = fabs(got - ref), tol = 2e-2 + 1e-2 * fabs(ref);
      if (has2) { const double e2 = fabs(bf2f(hC2[(size_t)m * N + n]) - ref2); if (e2 > 2e-2 + 1e-2 * fabs(ref2)) { ++bad; } if (e2 > maxerr) maxerr = e2; }
      if (err > tol) { if (bad < 5) printf("    bad C[%d][%d] = %g, ref %g\n", m, n, got, ref); ++bad; }
      if (err > maxerr) maxerr = err;
    }
  // column sums against the DEVICE output itself (every row, so untouched / doubly written rows show up here)
  double cserr = 0;
  if (use_cs) {
    std::vector<double> s(N, 0.0);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) s[n] += bf2f(hC[(size_t)m * N + n]);
    for (int n = 0; n < N; ++n) { double e = fabs(s[n] - hcs[n]) / (1.0 + fabs(s[n])); if (e > cserr) cserr = e; }
    if (cserr > 2e-3) ++bad;
  }
  // every element must have been written exactly with a finite value (0x7f7f pattern = 3.39e38 would blow the sums up)
  long untouched = 0;
  for (size_t i = 0; i < hC.size(); ++i) if (hC[i] == 0x7f7f) ++untouched;
  if (untouched) ++bad;
  printf("  kernel %d epi %d %6dx%5dx%5d: max err %.3g  colsum rel err %.2g  untouched %ld  %s\n", kernel, epi, M, N, K, maxerr, cserr, untouched, bad ? "FAIL" : "ok");
  return bad ? 1 : 0;
}

static int cmd_check() {
  int fails = 0;
  const int shapes[][3] = {{256, 256, 128}, {512, 768, 768}, {1000, 2304, 128}, {216, 304, 256}, {513, 520, 3072}, {6912, 768, 768}, {3000, 3072, 768}};
  for (int kernel : {8, 9, 3, 2, 1})
    for (auto& s : shapes)
      for (int epi : {SPMM_EPI_BF16, SPMM_EPI_GELU, SPMM_EPI_GELU_GRAD, SPMM_EPI_GELU_DERIV, SPMM_EPI_MUL}) fails += check_one(kernel, epi, s[0], s[1], s[2]);
  // repeated launches of one shape must agree bit for bit (a racy schedule shows up as run-to-run differences)
  {
    const int M = 8192, N = 3072, K = 768;
    auto hA = rand_bf16((size_t)M * K, 1.0f), hW = rand_bf16((size_t)N * K, 0.05f);
    Buf A, W, C0, C1; A.alloc(hA.size() * 2); W.alloc(hW.size() * 2); C0.alloc((size_t)M * N * 2); C1.alloc((size_t)M * N * 2);
    CK(hipMemcpy(A.d, hA.data(), A.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(W.d, hW.data(), W.bytes, hipMemcpyHostToDevice));
    run(1, SPMM_EPI_BF16, A, W, M, N, K, nullptr, nullptr, nullptr, C0, nullptr, nullptr, 0);
    std::vector<uint16_t> ref((size_t)M * N), got((size_t)M * N);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(ref.data(), C0.d, C0.bytes, hipMemcpyDeviceToHost));
    long diffs = 0, worst = 0;
    for (int it = 0; it < 20; ++it) {
      CK(hipMemset(C1.d, 0, C1.bytes));
      run(8, SPMM_EPI_BF16, A, W, M, N, K, nullptr, nullptr, nullptr, C1, nullptr, nullptr, 0);
      CK(hipDeviceSynchronize()); CK(hipMemcpy(got.data(), C1.d, C1.bytes, hipMemcpyDeviceToHost));
      long d = 0;
      for (size_t i = 0; i < got.size(); ++i) if (fabsf(bf2f(got[i]) - bf2f(ref[i])) > 0.02f + 0.01f * fabsf(bf2f(ref[i]))) ++d;
      diffs += d; worst = std::max(worst, d);
    }
    printf("  8-phase vs 128x128 kernel, full %dx%dx%d output, 20 launches: %ld elements off (worst launch %ld)  %s\n", M, N, K, diffs, worst, diffs ? "FAIL" : "ok");
    fails += diffs != 0;
  }
  printf("%s\n", fails ? "CHECK FAILED" : "CHECK OK");
  return fails ? 1 : 0;
}

static int cmd_time(int M, int N, int K, int epi, int rounds) {
  auto hA = rand_bf16((size_t)M * K, 1.0f), hW = rand_bf16((size_t)N * K, 0.05f);
  Buf A, W, C, C2, G, bias;
  A.alloc(hA.size() * 2); W.alloc(hW.size() * 2); C.alloc((size_t)M * N * 2); C2.alloc((size_t)M * N * 2); G.alloc((size_t)M * N * 2); bias.alloc(N * 4);
  CK(hipMemcpy(A.d, hA.data(), A.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(W.d, hW.data(), W.bytes, hipMemcpyHostToDevice));
  { auto hG = rand_bf16((size_t)M * N, 1.0f); CK(hipMemcpy(G.d, hG.data(), G.bytes, hipMemcpyHostToDevice)); }
  CK(hipMemset(bias.d, 0, bias.bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int kernels[] = {3, 9, 8};     // 256x256 one barrier per k-step | 8-phase, one workgroup per tile | 8-phase persistent
  const int NKER = 3;
  std::vector<float> best(NKER, 1e30f), med[NKER];
  const int iters = 10;
  for (int r = 0; r < rounds; ++r)
    for (int ki = 0; ki < NKER; ++ki) {
      for (int w = 0; w < 2; ++w) run(kernels[ki], epi, A, W, M, N, K, (const float*)bias.d, nullptr, (epi == SPMM_EPI_GELU_GRAD || epi == SPMM_EPI_MUL) ? &G : nullptr, C, (epi == SPMM_EPI_GELU || epi == SPMM_EPI_GELU_DERIV) ? &C2 : nullptr, nullptr, 0);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) {
        int rc = run(kernels[ki], epi, A, W, M, N, K, (const float*)bias.d, nullptr, (epi == SPMM_EPI_GELU_GRAD || epi == SPMM_EPI_MUL) ? &G : nullptr, C, (epi == SPMM_EPI_GELU || epi == SPMM_EPI_GELU_DERIV) ? &C2 : nullptr, nullptr, 0);
        if (rc) { printf("rc=%d %s\n", rc, spmm_last_error()); return 1; }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      best[ki] = std::min(best[ki], ms); med[ki].push_back(ms);
    }
  const double fl = 2.0 * M * N * K;
  printf("%6d x %5d x %5d epi %d :", M, N, K, epi);
  for (int ki = 0; ki < NKER; ++ki) {
    std::sort(med[ki].begin(), med[ki].end());
    const float m = med[ki][med[ki].size() / 2];
    printf("  k%d %8.1f us %7.1f TF (best %7.1f)", kernels[ki], m * 1e3, fl / (m * 1e-3) / 1e12, fl / (best[ki] * 1e-3) / 1e12);
  }
  printf("\n");
  return 0;
}


// ---- TN (weight-gradient) GEMM: C[N,K] += A[M,N]^T B[M,K]
static int tn_run(int kernel, const Buf& A, const Buf& B, int M, int N, int K, Buf& C, Buf& ws, int* splits_out = nullptr) {
  const int splits = spmm_gemm_tn_splits(M, N, K, kernel);
  if (splits_out) *splits_out = splits;
  if ((size_t)spmm_gemm_tn_workspace_bytes(M, N, K, splits) > ws.bytes) { printf("workspace too small\n"); return 1; }
  // GEMM_BENCH_LDA0=1: all rows of both operands alias row 0 (cache-resident operands: what the schedule does without HBM / fabric)
  static const bool ld0 = getenv("GEMM_BENCH_LDA0") != nullptr;
  return spmm_gemm_tn(A.d, ld0 ? 0 : N, B.d, ld0 ? 0 : K, M, N, K, splits, 1.0f, (float*)C.d, K, (float*)ws.d, kernel, nullptr, 0);
}
static int cmd_tncheck() {
  int fails = 0;
  const int shapes[][3] = {{4096, 768, 768}, {5000, 768, 3072}, {8320, 2304, 768}, {4224, 520, 264}, {16500, 3072, 768}, {300, 256, 256}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    auto hA = rand_bf16((size_t)M * N, 1.0f), hB = rand_bf16((size_t)M * K, 1.0f);
    Buf A, B, C1, C8, ws; A.alloc(hA.size() * 2); B.alloc(hB.size() * 2); C1.alloc((size_t)N * K * 4); C8.alloc((size_t)N * K * 4); ws.alloc((size_t)64 * N * K * 4);
    CK(hipMemcpy(A.d, hA.data(), A.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(B.d, hB.data(), B.bytes, hipMemcpyHostToDevice));
    std::vector<float> init((size_t)N * K);
    for (auto& x : init) x = urand();
    CK(hipMemcpy(C1.d, init.data(), C1.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(C8.d, init.data(), C8.bytes, hipMemcpyHostToDevice));
    int s1 = 0, s8 = 0;
    int rc = tn_run(1, A, B, M, N, K, C1, ws, &s1); CK(hipDeviceSynchronize());
    rc |= tn_run(8, A, B, M, N, K, C8, ws, &s8); CK(hipDeviceSynchronize());
    if (rc) { printf("  tn %dx%dx%d rc=%d %s\n", M, N, K, rc, spmm_last_error()); ++fails; continue; }
    std::vector<float> h1((size_t)N * K), h8((size_t)N * K);
    CK(hipMemcpy(h1.data(), C1.d, C1.bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(h8.data(), C8.d, C8.bytes, hipMemcpyDeviceToHost));
    double maxd = 0, maxref = 0; long bad = 0;
    for (size_t i = 0; i < h1.size(); ++i) { const double d = fabs((double)h1[i] - h8[i]); if (d > maxd) maxd = d; if (d > 2e-3 * sqrt((double)M) + 1e-4 * fabs(h1[i])) ++bad; }
    for (int t = 0; t < 400; ++t) {                     // fp64 reference on sampled outputs (incl. the last row / column)
      const int n = t < 8 ? N - 1 - t : (int)(fabsf(urand()) * (N - 1)), k = t < 8 ? K - 1 - 7 * t % K : (int)(fabsf(urand()) * (K - 1));
      double acc = init[(size_t)n * K + k];
      for (int m = 0; m < M; ++m) acc += (double)bf2f(hA[(size_t)m * N + n]) * (double)bf2f(hB[(size_t)m * K + k]);
      const double d = fabs(acc - h8[(size_t)n * K + k]);
      if (d > maxref) maxref = d;
      if (d > 2e-3 * sqrt((double)M) + 1e-4 * fabs(acc)) ++bad;
    }
    printf("  tn %6d x %5d x %5d: splits k1 %d k8 %d  max|k8 - k1| %.3g  max|k8 - fp64| %.3g  %s\n", M, N, K, s1, s8, maxd, maxref, bad ? "FAIL" : "ok");
    fails += bad != 0;
  }
  printf("%s\n", fails ? "TN CHECK FAILED" : "TN CHECK OK");
  return fails ? 1 : 0;
}
static int cmd_tntime(int M, int N, int K, int rounds) {
  auto hA = rand_bf16((size_t)M * N, 1.0f), hB = rand_bf16((size_t)M * K, 1.0f);
  Buf A, B, C, ws; A.alloc(hA.size() * 2); B.alloc(hB.size() * 2); C.alloc((size_t)N * K * 4); ws.alloc((size_t)64 * N * K * 4);
  CK(hipMemcpy(A.d, hA.data(), A.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(B.d, hB.data(), B.bytes, hipMemcpyHostToDevice));
  CK(hipMemset(C.d, 0, C.bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int kernels[] = {1, 8};
  std::vector<float> med[2];
  if (getenv("GEMM_BENCH_BETWEEN")) {                  // a 256-MiB memset between launches: the cache state inside the training step
    Buf scratch; scratch.alloc((size_t)256 << 20);
    for (int ki = 0; ki < 2; ++ki) {
      double tot = 0;
      for (int i = 0; i < 60; ++i) {
        CK(hipMemsetAsync(scratch.d, i & 255, scratch.bytes, 0));
        CK(hipEventRecord(e0, 0));
        if (tn_run(kernels[ki], A, B, M, N, K, C, ws)) { printf("%s\n", spmm_last_error()); return 1; }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 10) tot += ms;
      }
      med[ki].push_back((float)(tot / 50));
    }
    printf("tn %6d : %5d x %5d (dirty caches):", M, N, K);
    for (int ki = 0; ki < 2; ++ki) printf("  k%d %8.1f us %7.1f TF", kernels[ki], med[ki][0] * 1e3, 2.0 * M * N * K / (med[ki][0] * 1e-3) / 1e12);
    printf("\n");
    return 0;
  }
  for (int r = 0; r < rounds; ++r)
    for (int ki = 0; ki < 2; ++ki) {
      for (int w = 0; w < 2; ++w) tn_run(kernels[ki], A, B, M, N, K, C, ws);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 10; ++i) if (tn_run(kernels[ki], A, B, M, N, K, C, ws)) { printf("%s\n", spmm_last_error()); return 1; }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); med[ki].push_back(ms / 10);
    }
  printf("tn %6d : %5d x %5d :", M, N, K);
  for (int ki = 0; ki < 2; ++ki) { std::sort(med[ki].begin(), med[ki].end()); const float m = med[ki][med[ki].size() / 2]; printf("  k%d %8.1f us %7.1f TF", kernels[ki], m * 1e3, 2.0 * M * N * K / (m * 1e-3) / 1e12); }
  printf("\n");
  return 0;
}

// `sustain M N K [epi] [launches]`: back-to-back launches of one GEMM, TF/s per block of 100 -- does the rate of the first
// milliseconds (what `time` reports) hold once the chip has been at full MFMA load for a second?  With GEMM_BENCH_ROTATE=n the
// launches rotate over n different A / C buffer pairs (as the training step's GEMMs do) instead of re-using one.
static int cmd_sustain(int M, int N, int K, int epi, int launches) {
  const int nrot = getenv("GEMM_BENCH_ROTATE") ? atoi(getenv("GEMM_BENCH_ROTATE")) : 1;
  const bool rot_c_only = getenv("GEMM_BENCH_ROTATE_C_ONLY") != nullptr;
  const int kern = getenv("GEMM_BENCH_KERNEL") ? atoi(getenv("GEMM_BENCH_KERNEL")) : 8;      // 8 persistent 8-phase, 9 one workgroup per tile, 3 / 2 / 1     // A stays the same (cache-resident), only the output buffer rotates
  auto hA = rand_bf16((size_t)M * K, 1.0f), hW = rand_bf16((size_t)N * K, 0.05f);
  std::vector<Buf> A(nrot), C(nrot);
  Buf W, C2, G;
  for (int i = 0; i < nrot; ++i) { A[i].alloc(hA.size() * 2); C[i].alloc((size_t)M * N * 2); CK(hipMemcpy(A[i].d, hA.data(), A[i].bytes, hipMemcpyHostToDevice)); }
  W.alloc(hW.size() * 2); C2.alloc((size_t)M * N * 2); G.alloc((size_t)M * N * 2);
  CK(hipMemcpy(W.d, hW.data(), W.bytes, hipMemcpyHostToDevice)); CK(hipMemset(G.d, 0, G.bytes));
  const int nb = launches / 100;
  // GEMM_BENCH_BETWEEN=1: a 256-MiB memset between launches (a memory-bound kernel that also sweeps the caches);
  // GEMM_BENCH_BETWEEN=2: the GPU idles ~100 us between launches (the host waits); =3: a read-only sweep of 256 MiB (column sums:
  // the caches end up full of CLEAN lines).  Only the GEMMs are timed.
  const int between = getenv("GEMM_BENCH_BETWEEN") ? atoi(getenv("GEMM_BENCH_BETWEEN")) : 0;
  if (between) {
    Buf scratch, cs; scratch.alloc((size_t)(256 + 16) << 20); cs.alloc(768 * 4); CK(hipMemset(scratch.d, 0, scratch.bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot = 0; int n = 0;
    std::vector<double> blocks;
    for (int i = 0; i < launches; ++i) {
      const int r = i % nrot;
      if (between == 1) CK(hipMemsetAsync(scratch.d, i & 255, scratch.bytes, 0));
      else if (between == 3) { if (spmm_colsum_bf16(scratch.d, 768, 174762, 768, (float*)cs.d, nullptr, 0)) { printf("%s\n", spmm_last_error()); return 1; } }   // read-only sweep of 256 MiB
      else { CK(hipDeviceSynchronize()); struct timespec ts = {0, 100000}; nanosleep(&ts, nullptr); }
      CK(hipEventRecord(e0, 0));
      if (run(kern, epi, A[r], W, M, N, K, nullptr, nullptr, (epi == SPMM_EPI_GELU_GRAD || epi == SPMM_EPI_MUL) ? &G : nullptr, C[r],
              (epi == SPMM_EPI_GELU || epi == SPMM_EPI_GELU_DERIV) ? &C2 : nullptr, nullptr, 0)) { printf("%s\n", spmm_last_error()); return 1; }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; ++n;
      if (n == 100) { blocks.push_back(100 * 2.0 * M * N * K / (tot * 1e-3) / 1e12); tot = 0; n = 0; }
    }
    printf("%d x %d x %d epi %d, %d buffer pair(s), between=%d: TF/s per 100 launches:", M, N, K, epi, nrot, between);
    for (double b : blocks) printf(" %.0f", b);
    printf("\n");
    return 0;
  }
  std::vector<hipEvent_t> ev(nb + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  CK(hipEventRecord(ev[0], 0));
  for (int b = 0; b < nb; ++b) {
    for (int i = 0; i < 100; ++i) {
      const int r = (b * 100 + i) % nrot;
      if (run(8, epi, A[rot_c_only ? 0 : r], W, M, N, K, nullptr, nullptr, (epi == SPMM_EPI_GELU_GRAD || epi == SPMM_EPI_MUL) ? &G : nullptr, C[r],
              (epi == SPMM_EPI_GELU || epi == SPMM_EPI_GELU_DERIV) ? &C2 : nullptr, nullptr, 0)) { printf("%s\n", spmm_last_error()); return 1; }
    }
    CK(hipEventRecord(ev[b + 1], 0));
  }
  CK(hipDeviceSynchronize());
  printf("%d x %d x %d epi %d, %d buffer pair(s): TF/s per 100 launches:", M, N, K, epi, nrot);
  for (int b = 0; b < nb; ++b) { float ms; CK(hipEventElapsedTime(&ms, ev[b], ev[b + 1])); printf(" %.0f", 100 * 2.0 * M * N * K / (ms * 1e-3) / 1e12); }
  printf("\n");
  return 0;
}

// `contend M N K epi kernel nocc us [launches]`: the GEMM on stream 0 while `nocc` workgroups of a do-nothing kernel hold CUs for `us`
// microseconds on a second stream (what a long-running collective kernel does to a grid that counts on one workgroup per CU).
__global__ void __launch_bounds__(256) occupy_kernel(long long ticks) {
  __shared__ int pad[4096];                                  // 16 KiB of LDS: cannot share a CU with a 160-KiB GEMM workgroup
  pad[threadIdx.x] = 0;
  const long long t0 = wall_clock64();                       // 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static int cmd_contend(int M, int N, int K, int epi, int kern, int nocc, int us, int launches) {
  auto hA = rand_bf16((size_t)M * K, 1.0f), hW = rand_bf16((size_t)N * K, 0.05f);
  Buf A, W, C, C2, G;
  A.alloc(hA.size() * 2); W.alloc(hW.size() * 2); C.alloc((size_t)M * N * 2); C2.alloc((size_t)M * N * 2); G.alloc((size_t)M * N * 2);
  CK(hipMemcpy(A.d, hA.data(), A.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(W.d, hW.data(), W.bytes, hipMemcpyHostToDevice)); CK(hipMemset(G.d, 0, G.bytes));
  hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int i = 0; i < launches + 10; ++i) {
    if (nocc > 0) { occupy_kernel<<<nocc, 256, 0, s2>>>((long long)us * 100); struct timespec ts = {0, 30000}; nanosleep(&ts, nullptr); }   // the occupier is resident first
    CK(hipEventRecord(e0, 0));
    if (run(kern, epi, A, W, M, N, K, nullptr, nullptr, (epi == SPMM_EPI_GELU_GRAD || epi == SPMM_EPI_MUL) ? &G : nullptr, C,
            (epi == SPMM_EPI_GELU || epi == SPMM_EPI_GELU_DERIV) ? &C2 : nullptr, nullptr, 0)) { printf("%s\n", spmm_last_error()); return 1; }
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 10) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const float med = t[t.size() / 2];
  printf("%d x %d x %d epi %d kernel %d, %d occupier workgroups for %d us: GEMM median %.1f us (%.0f TF/s), p10 %.1f, p90 %.1f\n", M, N, K, epi, kern, nocc, us,
         med * 1e3, 2.0 * M * N * K / (med * 1e-3) / 1e12, t[t.size() / 10] * 1e3, t[t.size() * 9 / 10] * 1e3);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 9 && !strcmp(argv[1], "contend"))
    return cmd_contend(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8]), argc > 9 ? atoi(argv[9]) : 100);
  if (argc >= 5 && !strcmp(argv[1], "sustain")) return cmd_sustain(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 0, argc > 6 ? atoi(argv[6]) : 2000);
  if (argc >= 2 && !strcmp(argv[1], "check")) return cmd_check();
  if (argc >= 2 && !strcmp(argv[1], "tncheck")) return cmd_tncheck();
  if (argc >= 5 && !strcmp(argv[1], "tntime")) return cmd_tntime(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 5);
  if (argc >= 2 && !strcmp(argv[1], "tnstep")) {
    int rc = 0;
    for (auto& sh : std::vector<std::array<int, 3>>{{84256, 3072, 768}, {84256, 768, 768}, {84256, 768, 3072}, {84256, 2304, 768}, {28304, 768, 3072}, {13824, 3072, 768}, {13824, 768, 768}})
      rc |= cmd_tntime(sh[0], sh[1], sh[2], 5);
    return rc;
  }
  if (argc >= 5 && !strcmp(argv[1], "time"))
    return cmd_time(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), argc > 5 ? atoi(argv[5]) : 0, argc > 6 ? atoi(argv[6]) : 5);
  if (argc >= 2 && !strcmp(argv[1], "step")) {   // the training step's dominant shapes (B=128, Lt=128: ~84k fusion tokens)
    int rc = 0;
    const int M = 84000;
    rc |= cmd_time(M, 3072, 768, SPMM_EPI_GELU_DERIV, 5);
    rc |= cmd_time(M, 3072, 768, SPMM_EPI_MUL, 5);
    rc |= cmd_time(M, 3072, 768, SPMM_EPI_GELU, 5);
    rc |= cmd_time(M, 3072, 768, SPMM_EPI_GELU_GRAD, 5);
    rc |= cmd_time(M, 3072, 768, SPMM_EPI_BF16, 5);
    rc |= cmd_time(M, 768, 768, SPMM_EPI_BF16, 5);
    rc |= cmd_time(M, 2304, 768, SPMM_EPI_BF16, 5);
    rc |= cmd_time(M, 768, 3072, SPMM_EPI_BF16, 5);
    rc |= cmd_time(32768, 768, 768, SPMM_EPI_BF16, 5);
    rc |= cmd_time(13824, 3072, 768, SPMM_EPI_GELU, 5);
    rc |= cmd_time(4096, 4096, 4096, SPMM_EPI_BF16, 5);
    rc |= cmd_time(8192, 8192, 8192, SPMM_EPI_BF16, 3);
    return rc;
  }
  if (argc >= 5 && !strcmp(argv[1], "prof")) {    // needs the -DP8_PROFILE build (build/gemm_bench_prof): per-tile cycle stamps of wave 0
    const int M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), epi = argc > 5 ? atoi(argv[5]) : 0;
    auto hA = rand_bf16((size_t)M * K, 1.0f), hW = rand_bf16((size_t)N * K, 0.05f);
    Buf A, W, C, C2, G, prof;
    A.alloc(hA.size() * 2); W.alloc(hW.size() * 2); C.alloc((size_t)M * N * 2); C2.alloc((size_t)M * N * 2); G.alloc((size_t)M * N * 2);
    prof.alloc(256 * 16 * 8 * 8);
    CK(hipMemcpy(A.d, hA.data(), A.bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(W.d, hW.data(), W.bytes, hipMemcpyHostToDevice));
    CK(hipMemset(G.d, 0, G.bytes));
    const long lda = getenv("GEMM_BENCH_LDA0") ? 0 : K;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int it = 0; it < 3; ++it) {
      CK(hipMemset(prof.d, 0, prof.bytes));
      CK(hipEventRecord(e0, 0));
      int rc = spmm_gemm_nt(A.d, lda, W.d, K, M, N, K, 1, nullptr, nullptr, 1.0f, nullptr, N, (epi == SPMM_EPI_GELU_GRAD || epi == SPMM_EPI_MUL) ? G.d : nullptr, N, C.d, N,
                            (epi == SPMM_EPI_GELU || epi == SPMM_EPI_GELU_DERIV) ? C2.d : nullptr, N, epi, (float*)prof.d, 8, nullptr, 0);
      CK(hipEventRecord(e1, 0));
      if (rc) { printf("rc=%d %s\n", rc, spmm_last_error()); return 1; }
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(256 * 16 * 8);
    CK(hipMemcpy(h.data(), prof.d, prof.bytes, hipMemcpyDeviceToHost));
    // averages over every recorded (workgroup, tile): stamp 0 = tile start, 2 = main loop done, 4 = epilogue done
    double main_s = 0, epi_s = 0, tot = 0; long cnt = 0, cnt_tot = 0;
    for (int wg = 0; wg < 256; ++wg)
      for (int i = 0; i < 16; ++i) {
        const unsigned long long* t = &h[((size_t)wg * 16 + i) * 8];
        if (!t[0] || !t[4]) continue;
        main_s += (double)(t[2] - t[0]); epi_s += (double)(t[4] - t[2]);
        ++cnt;
        if (i < 15 && t[8]) { tot += (double)(t[8] - t[0]); ++cnt_tot; }
      }
    unsigned long long first = ~0ull, last = 0, first_max = 0, last_min = ~0ull;
    for (int wg = 0; wg < 256; ++wg) {
      unsigned long long f = 0, l = 0;
      for (int i = 0; i < 16; ++i) {
        const unsigned long long* t = &h[((size_t)wg * 16 + i) * 8];
        if (!t[0] || !t[4]) continue;
        if (!f) f = t[0];
        l = t[4];
      }
      if (!f) continue;
      first = std::min(first, f); first_max = std::max(first_max, f); last = std::max(last, l); last_min = std::min(last_min, l);
    }
    const double span = (double)(last - first);
    printf("avg over %ld tiles: mainloop %.0f  epilogue %.0f  | tile period %.0f (ticks)\n", cnt, main_s / cnt, epi_s / cnt, tot / (cnt_tot ? cnt_tot : 1));
    printf("launch %.1f us; stamps span %.0f ticks (=> >= %.0f ticks/us); first tile starts spread over %.0f ticks, last tile ends over %.0f ticks\n",
           ms * 1e3, span, span / (ms * 1e3), (double)(first_max - first), (double)(last - last_min));
    return 0;
  }
  fprintf(stderr, "usage: gemm_bench check | time M N K [epi] [rounds] | step | prof M N K [epi]\n");
  return 2;
}
