"""decode_attn (one wave per (molecule, head), k = 5 beams) against how much the beams' ancestries differ: all beams on the same cache rows,
the last `d` positions on their own rows (what beam search produces), everything different.  us per launch at 5 000 rows, 12 heads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
BF = torch.bfloat16
R, nH, H, k, Lmax = 5000, 12, 768, 5, 103
q = torch.randn(R, H, device="cuda").to(BF)
kc = torch.randn(R, nH, Lmax, 64, device="cuda").to(BF)
vc = torch.randn(R, nH, Lmax, 64, device="cuda").to(BF)
out = torch.empty(R, H, device="cuda", dtype=BF)
rows = torch.arange(R, dtype=torch.int32, device="cuda")
lead = (rows // k * k)[:, None].expand(R, Lmax)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    v = sorted(x.elapsed_time(y) for x, y in ev)
    return v[len(v) // 2] * 1e3


print(f"{'keys':>5s} {'own positions':>14s} {'us':>8s}")
for t in (25, 50, 100):
    for d in (0, 1, 2, 4, 6, 10, t):
        own = torch.arange(Lmax, device="cuda")[None, :] >= t - d
        anc = torch.where(own, rows[:, None].expand(R, Lmax), lead).to(torch.int32).contiguous()
        us = timeit(lambda: ops.decode_attn(q, kc, vc, out, nH=nH, Lkv=t, seq_stride=Lmax * H, tok_stride=64, head_stride=Lmax * 64, anc=anc, group=k))
        print(f"{t:5d} {d:14d} {us:8.1f}", flush=True)
