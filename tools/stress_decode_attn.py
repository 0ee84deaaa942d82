"""decode_attn against fp32 torch over many random ancestry tables (a race screen: the kernel keeps blocks of keys and values in flight
behind counted waits): failures per configuration out of N trials."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
BF = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CFG = [(15, 12, 37, 64, 5, 0), (15, 12, 37, 64, 5, 6), (64, 4, 130, 160, 4, 0), (500, 12, 100, 100, 5, 0), (500, 12, 100, 100, 5, 6), (5000, 12, 60, 103, 5, 6),
       (18, 2, 33, 40, 6, 0), (12, 4, 70, 80, 2, 0), (6, 2, 256, 256, 3, 0)]
for R, nH, Lkv, Lmax, group, tail in CFG:
    H = nH * 64
    bad = 0
    worst = 0.0
    for trial in range(N):
        g = torch.Generator().manual_seed(1000 * trial + R + Lkv)
        q = torch.randn(R, H, generator=g).to(BF).cuda()
        Kc = torch.randn(R, Lmax, H, generator=g).to(BF).cuda()
        Vc = torch.randn(R, Lmax, H, generator=g).to(BF).cuda()
        anc = torch.randint(0, R, (R, Lmax), generator=g).to(torch.int32).cuda()
        if tail:
            if trial % 2:      # beams 2k and 2k+1 on the same rows in the tail (fewer distinct rows than beams)
                anc = anc.view(R // group, group, Lmax)[:, (torch.arange(group) // 2 * 2).tolist(), :].reshape(R, Lmax).contiguous()
            lead = anc.view(R // group, group, Lmax)[:, :1, :].expand(R // group, group, Lmax).reshape(R, Lmax)
            old = torch.arange(Lmax, device="cuda")[None, :] < max(Lkv - tail, 0)
            anc = torch.where(old, lead, anc).contiguous()
        out = torch.zeros(R, H, dtype=BF, device="cuda")
        ops.decode_attn(q, Kc, Vc, out, nH=nH, Lkv=Lkv, seq_stride=Lmax * H, tok_stride=H, anc=anc, group=group)
        j = torch.arange(Lkv, device="cuda")
        seq = anc[:, :Lkv].long()
        K = Kc.float()[seq, j[None, :]].view(R, Lkv, nH, 64)
        V = Vc.float()[seq, j[None, :]].view(R, Lkv, nH, 64)
        sc = torch.einsum("rhd,rjhd->rhj", q.float().view(R, nH, 64), K) * 0.125
        ref = torch.einsum("rhj,rjhd->rhd", torch.softmax(sc, -1), V).reshape(R, H)
        err = (out.float() - ref).abs()
        nb = int((err > 2e-2 + 2e-2 * ref.abs()).sum())
        bad += nb > 0
        worst = max(worst, float(err.max()))
    print(f"R={R} nH={nH} Lkv={Lkv} group={group} own-tail={tail}: {bad}/{N} trials with errors, worst |err| {worst:.3g}", flush=True)
