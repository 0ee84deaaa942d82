"""Is the NT GEMM bound by the memory side?  Same launch with A and/or W rows aliased (row stride 0 => trivially cache resident)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
from spmm_amd._lib import lib
def run(M, N, K, aliasA, aliasW, iters=20):
    A = torch.randn(1 if aliasA else M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(1 if aliasW else N, K, device="cuda") * 0.05).to(torch.bfloat16)
    if aliasA: A = A.expand(M, K)
    if aliasW: W = W.expand(N, K)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm_nt(A, W, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_nt(A, W, C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return 2.0 * M * N * K / ms / 1e9
for (M, N, K) in [(93184, 3072, 768), (93184, 768, 3072), (8192, 8192, 8192)]:
    for var in (101, 100):
        lib().cdll.spmm_gemm_set_variant(var)
        r = [run(M, N, K, a, w) for (a, w) in ((False, False), (True, False), (False, True), (True, True))]
        print(f"{'v3' if var == 101 else 'v2'} M={M} N={N} K={K}: normal {r[0]:.0f}  A-aliased {r[1]:.0f}  W-aliased {r[2]:.0f}  both {r[3]:.0f} TF", flush=True)
