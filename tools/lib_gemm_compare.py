"""The step's GEMM shapes on this package's kernels and on the vendor library (torch.nn.functional.linear / torch.mm = hipBLASLt /
rocBLAS on ROCm), same box, same buffers, HIP events around `reps` back-to-back launches.  A measuring tool only: the product
never calls the library.   python tools/lib_gemm_compare.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = 84256                                             # token rows of the largest merged batch of the benchmark step


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3           # us


print(f"{'kind':3s} {'M':>6s} {'N':>5s} {'K':>5s} | {'ours us':>8s} {'TF/s':>6s} | {'library us':>10s} {'TF/s':>6s} | ours/library time")
g = torch.Generator(device=dev).manual_seed(1)
for (N, K) in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
    A = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    b = torch.zeros(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t0 = timed(lambda: ops.gemm_nt(A, W, C, bias=b))
    t1 = timed(lambda: torch.mm(A, W.t(), out=C2))
    err = (C.float() - C2.float()).abs().max().item()
    fl = 2.0 * M * N * K
    print(f"nt  {M:6d} {N:5d} {K:5d} | {t0:8.1f} {fl / t0 / 1e6:6.0f} | {t1:10.1f} {fl / t1 / 1e6:6.0f} | {t0 / t1:5.2f}   (max |diff| {err:.3g}, bias is zero)")
for (N, K) in ((768, 768), (3072, 768), (768, 3072)):
    A = torch.randn(M, N, device=dev, generator=g).bfloat16()
    B = torch.randn(M, K, device=dev, generator=g).bfloat16()
    C = torch.zeros(N, K, device=dev)
    C2 = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    t0 = timed(lambda: ops.gemm_tn(A, B, C))
    t1 = timed(lambda: torch.mm(A.t(), B, out=C2))    # the library writes bf16 (fp32 accumulation inside); ours accumulates into fp32 C
    fl = 2.0 * M * N * K
    print(f"tn  {M:6d} {N:5d} {K:5d} | {t0:8.1f} {fl / t0 / 1e6:6.0f} | {t1:10.1f} {fl / t1 / 1e6:6.0f} | {t0 / t1:5.2f}")
