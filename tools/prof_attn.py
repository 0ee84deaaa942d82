"""Phase breakdown of the eight-wave attention backward (Lkv > 128) from s_memtime stamps (build/libspmm_hip_attprof.so, ATT_PROFILE):
   make -C tools ../build/libspmm_hip_attprof.so ; SPMM_HIP_LIB=build/libspmm_hip_attprof.so python tools/prof_attn.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmm_amd import ops
BF = torch.bfloat16
names = ["staging + bias vectors -> barrier", "phase A1: scores, dP, exp, partial D", "barrier + D exchange", "phase A2: dS, dQ MFMAs",
         "barrier (K/V dead)", "P~ / partial dQ to LDS, barrier, dQ store", "phase B: dV", "barrier", "dS to LDS + barrier", "phase B: dK",
         "barrier", "transposition + dK/dV stores"]
for (nseq, Lq, Lkv, p_drop) in ((512, 128, 256, 0.1), (512, 128, 256, 0.0), (512, 54, 256, 0.1)):
    nH, H = 12, 768
    q = torch.randn(nseq * Lq, H, device="cuda").to(BF); kv = torch.randn(nseq * Lkv, 2 * H, device="cuda").to(BF)
    O = torch.empty_like(q); lse = torch.empty(nseq, nH, Lq, device="cuda"); dO = torch.randn_like(q); dQ = torch.empty_like(q); dKV = torch.empty_like(kv)
    seed = torch.full((1,), 1234, dtype=torch.int64, device="cuda")
    kw = dict(nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, is_cross=True, dropout_p=p_drop, seed=seed, salt=3)
    ops.attn_fwd(q, kv[:, :H], kv[:, H:], O, lse, **kw)
    stamps = torch.zeros(nseq * nH * 8 * 16 * 2, device="cuda")           # uint64 slots viewed as float32 pairs
    for _ in range(3):
        ops.attn_bwd(q, kv[:, :H], kv[:, H:], O, lse, dO, dQ, dKV[:, :H], dKV[:, H:], dbuf=stamps, **kw)
    torch.cuda.synchronize()
    st = stamps.view(torch.int64).view(nseq * nH, 8, 16)[:, :, :13].double()
    d = st[:, :, 1:] - st[:, :, :-1]                                        # [blocks, waves, 12]
    tot = (st[:, :, 12] - st[:, :, 0]).mean().item()
    print(f"\n{nseq} sequences, {Lq} x {Lkv}, dropout {p_drop}: {tot:.0f} cycles per workgroup (mean over waves and workgroups)")
    for i, nm in enumerate(names):
        print(f"  {nm:46s} {d[:, :, i].mean().item():8.0f}  (slowest wave {d[:, :, i].max(dim=1).values.mean().item():8.0f})")
