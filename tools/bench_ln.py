"""LayerNorm backward at the benchmark shape: stored-sum form against the from-output form (spmm_ln_bwd beta_from_y); us per launch, GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
for rows in (84992, 28672, 13824):
    H = 768
    x = torch.randn(rows, H, device="cuda").bfloat16(); res = torch.randn(rows, H, device="cuda").bfloat16()
    gamma = (1 + 0.1 * torch.randn(H)).cuda(); beta = (0.1 * torch.randn(H)).cuda()
    y, z = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    seed = torch.full((1,), 7, dtype=torch.int64, device="cuda")
    dy = torch.randn(rows, H, device="cuda").bfloat16()
    dz, dx = torch.empty_like(x), torch.empty_like(x)
    dg, db, dxs = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    for name, kw_f, kw_b in (("stored sum", dict(zout=z), dict(z=z, mean=mean, beta_from_y=None)), ("from output", dict(zout=None), dict(z=y, mean=None, beta_from_y=beta))):
        def fwd(): ops.ln_fwd(x, res, gamma, beta, y, mean=mean, rstd=rstd, dropout_p=0.1, seed=seed, salt=3, **kw_f)
        def bwd(): ops.ln_bwd(dy, kw_b["z"], kw_b["mean"], rstd, gamma, dz, dx=dx, dgamma=dg, dbeta=db, dxsum=dxs, dropout_p=0.1, seed=seed, salt=3, beta_from_y=kw_b["beta_from_y"])
        out = []
        for fn in (fwd, bwd):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"rows {rows:6d}  {name:12s}  forward {out[0]:7.1f} us  backward {out[1]:7.1f} us")
