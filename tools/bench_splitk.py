import sys, os, torch
sys.path.insert(0, os.getcwd())
from spmm_amd import ops
BF = torch.bfloat16
M, N, K = 256, 256, 36992
A = torch.randn(M, K, device="cuda").to(BF); W = torch.randn(N, K, device="cuda").to(BF); C = torch.zeros(M, N, device="cuda")
temp = torch.full((1,), 0.07, device="cuda")
for sp in (8, 16, 24, 32, 48, 64, 96, 128):
    for _ in range(3): ops.gemm_nt(A, W, C, epi=ops.EPI_F32_ATOMIC, splits=sp, div=temp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_nt(A, W, C, epi=ops.EPI_F32_ATOMIC, splits=sp, div=temp)
    e1.record(); torch.cuda.synchronize()
    print(f"splits {sp:4d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
