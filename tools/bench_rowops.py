"""HBM-streaming row kernels at the S6 shape: GB/s against the algorithmic bytes of each call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
M, H = 84256, 768
dev = "cuda"
BF = torch.bfloat16
x = torch.randn(M, H, device=dev).to(BF); r = torch.randn(M, H, device=dev).to(BF)
g = torch.rand(H, device=dev) + 0.5; b = torch.randn(H, device=dev)
y = torch.empty_like(x); z = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)
dz = torch.empty_like(x); dx = torch.empty_like(x); dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); dxs = torch.zeros(H, device=dev)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

row = M * H * 2
cases = [
    ("ln_fwd  x+res -> y,z (train, dropout)", lambda: ops.ln_fwd(x, r, g, b, y, zout=z, mean=mean, rstd=rstd, dropout_p=0.1, seed=seed, salt=7), 4 * row),
    ("ln_fwd  x+res -> y (eval)", lambda: ops.ln_fwd(x, r, g, b, y), 3 * row),
    ("ln_bwd  dy,z -> dz,dx (dropout, dgamma, dxsum)", lambda: ops.ln_bwd(x, z, mean, rstd, g, dz, dx=dx, dgamma=dg, dbeta=db, dropout_p=0.1, seed=seed, salt=7, dxsum=dxs), 4 * row),
    ("ln_bwd  dy,z -> dz,dx (dropout, dgamma, dbeta, dxsum: the training combination)", lambda: ops.ln_bwd(x, z, mean, rstd, g, dz, dx=dx, dgamma=dg, dbeta=db, dropout_p=0.1, seed=seed, salt=7, dxsum=dxs), 4 * row),
    ("ln_bwd  dy,z -> dz (no dropout, dgamma)", lambda: ops.ln_bwd(x, z, mean, rstd, g, dz, dgamma=dg, dbeta=db), 3 * row),
]
ops.ln_fwd(x, r, g, b, y, zout=z, mean=mean, rstd=rstd)
for name, fn, bytes_ in cases:
    us = timeit(fn)
    print(f"{name:52s} {us:8.1f} us  {bytes_ / us / 1e6:7.2f} TB/s")
