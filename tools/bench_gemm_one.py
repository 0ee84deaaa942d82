"""One GEMM shape, a few launches -- for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
if len(sys.argv) > 5:
    from spmm_amd._lib import lib
    lib().cdll.spmm_gemm_set_variant(int(sys.argv[5]))
A = torch.randn(M, K if kind == "nt" else N, device="cuda").to(torch.bfloat16)
if kind == "nt":
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16); bias = torch.randn(N, device="cuda")
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(6): ops.gemm_nt(A, W, C, bias=bias)
else:
    B = torch.randn(M, K, device="cuda").to(torch.bfloat16); C = torch.zeros(N, K, device="cuda")
    for _ in range(6): ops.gemm_tn(A, B, C)
torch.cuda.synchronize()
