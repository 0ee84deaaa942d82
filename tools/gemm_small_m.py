"""The decoder's GEMM shapes (M = molecules x beams = 5 000 rows) on every NT tile kernel: what spmm_gemm_nt's automatic choice
(kernel 0) picks against the forced alternatives.  us per launch, 50 back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5000


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'N':>5s} {'K':>5s} {'epi':>4s} | " + " ".join(f"{'k' + str(k):>8s}" for k in (0, 1, 5, 2, 3, 8, 9)) + "   (us; TF/s of the automatic choice)")
for N, K, epi in ((2304, 768, 0), (768, 768, 0), (3072, 768, ops.EPI_GELU), (768, 3072, 0), (1536, 768, 0)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.zeros(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    for k in (0, 1, 5, 2, 3, 8, 9):
        try:
            row.append(timed(lambda: ops.gemm_nt(A, W, C, bias=b, epi=epi, kernel=k)))
        except RuntimeError:
            row.append(float("nan"))
    print(f"{N:5d} {K:5d} {epi:4d} | " + " ".join(f"{t:8.1f}" for t in row) + f"   {2.0 * M * N * K / row[0] / 1e6:6.0f}")
