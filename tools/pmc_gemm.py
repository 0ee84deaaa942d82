"""Tiny driver for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): the dominant NT GEMM shapes of the
step, 3 launches each, plus a 16 B/lane streaming copy of known size used to calibrate the counters (MI355X_MICROARCH.md, HBM)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmm_amd import ops
KERNEL = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # spmm_gemm_nt kernel selector (0 = default choice)

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(93184, 3072, 768, ops.EPI_GELU), (93184, 768, 3072, ops.EPI_BF16), (93184, 2304, 768, ops.EPI_BF16), (93184, 768, 768, ops.EPI_BF16)]
cal = torch.randn(93184, 768, device=dev).bfloat16()
for _ in range(3):
    cal2 = cal.clone()            # reads 143.1 MB, writes 143.1 MB
for M, N, K, epi in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
    b = torch.zeros(N, device=dev); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(A, W, C, bias=b, epi=epi, C2=torch.empty_like(C) if epi == ops.EPI_GELU else None, kernel=KERNEL)
    torch.cuda.synchronize()
    del A, W, C
print("ok")
