"""GEMM micro-benchmark at the shapes of the pretraining step (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
from spmm_amd._lib import lib

def bench(M, N, K, epi=ops.EPI_BF16, splits=1, iters=30):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    f32 = epi in (ops.EPI_F32, ops.EPI_F32_ATOMIC, ops.EPI_F32_ACC)
    C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    C2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == ops.EPI_GELU else None
    for _ in range(3):
        ops.gemm_nt(A, W, C, bias=None if epi == ops.EPI_F32_ATOMIC else bias, epi=epi, C2=C2, splits=splits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_nt(A, W, C, bias=None if epi == ops.EPI_F32_ATOMIC else bias, epi=epi, C2=C2, splits=splits)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9

if __name__ == "__main__":
    shapes = [(16384, 768, 768), (16384, 2304, 768), (16384, 3072, 768), (16384, 768, 3072), (6912, 2304, 768),
              (32768, 3072, 768), (93184, 3072, 768), (93184, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]
    for st in (1, 0):
        lib().cdll.spmm_gemm_set_staging(st)
        for (M, N, K) in shapes:
            ms, tf = bench(M, N, K)
            print(f"staging={'dma' if st else 'reg'} M={M} N={N} K={K}: {ms:.3f} ms  {tf:.1f} TFLOP/s", flush=True)
    lib().cdll.spmm_gemm_set_staging(1)
    ms, tf = bench(16384, 3072, 768, epi=ops.EPI_GELU); print(f"gelu epi 16384x3072x768: {ms:.3f} ms {tf:.1f} TF")
    for sp in (1, 4, 8, 16):   # wgrad-like: dW[768,768] += dY^T[768, 16384] X^T[768,16384]^T
        ms, tf = bench(768, 768, 16384, epi=ops.EPI_F32_ATOMIC, splits=sp); print(f"wgrad 768x768x16384 splits={sp}: {ms:.3f} ms {tf:.1f} TF")
    ms, tf = bench(3072, 768, 16384, epi=ops.EPI_F32_ATOMIC, splits=4); print(f"wgrad 3072x768x16384 splits=4: {ms:.3f} ms {tf:.1f} TF")
    # torch (hipBLASLt) reference point on the same shapes
    for (M, N, K) in [(16384, 3072, 768), (16384, 768, 3072), (8192, 8192, 8192)]:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
        for _ in range(3): torch.matmul(A, W.t())
        torch.cuda.synchronize(); t = time.time()
        for _ in range(20): torch.matmul(A, W.t())
        torch.cuda.synchronize(); ms = (time.time() - t) / 20 * 1e3
        print(f"torch.matmul {M}x{N}x{K}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TF")


def bench_tn(M, N, K, iters=20):
    A = torch.randn(M, N, device="cuda").to(torch.bfloat16); B = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    C = torch.zeros(N, K, device="cuda")
    for _ in range(3): ops.gemm_tn(A, B, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_tn(A, B, C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9
