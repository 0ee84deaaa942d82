"""attn_bwd / attn_fwd at the self-text S6 shape, a few launches, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
BF = torch.bfloat16; dev = "cuda"; H, nH = 768, 12
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)
nseq, Lq, Lkv = 512, 128, 128
q = torch.randn(nseq * Lq, 3 * H, device=dev).to(BF)
Q, K, V = q[:, :H], q[:, H:2 * H], q[:, 2 * H:]
O = torch.empty(nseq * Lq, H, device=dev, dtype=BF); lse = torch.empty(nseq, nH, Lq, device=dev)
dO = torch.randn(nseq * Lq, H, device=dev).to(BF); dQKV = torch.empty_like(q)
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
kw = dict(nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, dropout_p=p, seed=seed, salt=3)
for _ in range(4):
    ops.attn_fwd(Q, K, V, O, lse, **kw)
    ops.attn_bwd(Q, K, V, O, lse, dO, dQKV[:, :H], dQKV[:, H:2 * H], dQKV[:, 2 * H:], **kw)
torch.cuda.synchronize()
print("ok")
