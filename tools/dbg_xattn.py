import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
from spmm_amd import ops
import test_kernels_gpu as T
BF = torch.bfloat16
nH, nseq, U, Lq, Lkv = 12, 6, 3, 54, 128
H = nH * 64
g = torch.Generator().manual_seed(nseq * 7 + Lq)
idx = torch.randint(0, U, (nseq,), generator=g); idx[:U] = torch.arange(U)
kmask = (torch.rand(nseq, Lkv, generator=g) > 0.25).int(); kmask[:, 0] = 1
lay = dict(kmask=kmask.cuda(), kv_seq=idx.to(torch.int32).cuda())
M, Mkv = nseq * Lq, U * Lkv
Q = T.rnd(M, H, seed=60, scale=1.5); KV = T.rnd(Mkv, 2 * H, seed=61); R = T.rnd(M, H, seed=62); Wo = T.rnd(H, H, seed=63, scale=0.04)
bo = torch.zeros(H).cuda(); gamma = torch.ones(H).cuda(); beta = torch.zeros(H).cuda()
seed = torch.tensor([1234567], dtype=torch.int64, device="cuda")
ref = T._xattn_composite(ops, Q, KV[:, :H], KV[:, H:], Wo, bo, R, gamma, beta, nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, eps=1e-12, pa=0.0, ph=0.0, seed=seed, salt_a=11, salt_h=22, row_base=0, **lay)
WoF = ops.xattn_pack_wo(Wo)
out = {k: torch.full_like(v, float("nan")) for k, v in ref.items()}
ops.xattn_fwd(Q, KV[:, :H], KV[:, H:], WoF, bo, R, gamma, beta, out["y"], nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, Z=out["z"], mean=out["mean"], rstd=out["rstd"], CTX=out["ctx"], lse=out["lse"], seed=seed, **lay)
torch.cuda.synchronize()
d = (out["ctx"].float() - ref["ctx"].float()).view(nseq, Lq, nH, 64)
bad = d.abs() > 0
print("ctx mismatch fraction", bad.float().mean().item(), "nan", torch.isnan(out["ctx"].float()).float().mean().item(), "max", d.abs().nan_to_num(9).max().item())
print("per head:", bad.float().mean(dim=(0, 1, 3)).tolist())
print("per seq:", bad.float().mean(dim=(1, 2, 3)).tolist())
print("per row (seq 0):", bad[0].float().mean(dim=(1, 2)).tolist())
print("per d (head of worst):", bad.float().mean(dim=(0, 1, 2)).tolist())
print("y max diff", (out["y"].float() - ref["y"].float()).abs().nan_to_num(9).max().item(), "z", (out["z"].float() - ref["z"].float()).abs().nan_to_num(9).max().item())
