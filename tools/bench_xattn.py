"""Fused cross-attention block (csrc/xattn.hip) against the composite it replaces (attention core + output GEMM + LayerNorm) at the
S6 shapes of the benchmark step: B = 128 unique sources, 512 query sequences per direction, training mode (both dropouts, every
tensor the backward needs is written).  Prints microseconds per call and the executed TFLOP/s of core + output projection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
BF = torch.bfloat16
dev = "cuda"
H, nH = 768, 12
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)


def timeit(fn, n=20, dirty=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if dirty is not None:
            dirty.zero_()                      # caches full of dirty lines, as after any producer kernel of the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) for a, b in ts)
    return v[len(v) // 2] * 1e3


dirty = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
print(f"{'shape':34s} {'mode':>8s} {'composite':>10s} {'fused':>8s} {'TF/s fused':>11s}   (us, median; dirty = 256-MiB memset before every call)")
for nm, nseq, U, Lq, Lkv in (("PV -> text 54x128, 512 seq", 512, 128, 54, 128), ("text -> PV 128x54, 512 seq", 512, 128, 128, 54),
                             ("text -> PV 128x54, 128 seq (S5)", 128, 128, 128, 54)):
    M = nseq * Lq
    Q = torch.randn(M, H, device=dev).to(BF); KV = torch.randn(U * Lkv, 2 * H, device=dev).to(BF); R = torch.randn(M, H, device=dev).to(BF)
    Wo = (torch.randn(H, H, device=dev) * 0.03).to(BF); bo = torch.zeros(H, device=dev); gm = torch.ones(H, device=dev); bt = torch.zeros(H, device=dev)
    idx = (torch.arange(nseq, device=dev) % U).to(torch.int32)
    WoF = ops.xattn_pack_wo(Wo)
    ctx, x, y, z = (torch.empty(M, H, device=dev, dtype=BF) for _ in range(4))
    lse = torch.empty(nseq, nH, Lq, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    flops = 4.0 * nseq * Lq * Lkv * H + 2.0 * M * H * H
    for mode, pa, ph, save in (("train", 0.1, 0.1, True), ("eval", 0.0, 0.0, False)):
        def comp():
            ops.attn_fwd(Q, KV[:, :H], KV[:, H:], ctx, lse if save else None, nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, is_cross=True, kv_seq=idx,
                         dropout_p=pa, seed=seed, salt=3)
            ops.gemm_nt(ctx, Wo, x, bias=bo)
            ops.ln_fwd(x, R, gm, bt, y, zout=x if save else None, mean=mean if save else None, rstd=rstd if save else None, dropout_p=ph,
                       seed=seed, salt=4)

        def fused():
            ops.xattn_fwd(Q, KV[:, :H], KV[:, H:], WoF, bo, R, gm, bt, y, nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, kv_seq=idx, Z=z if save else None,
                          mean=mean if save else None, rstd=rstd if save else None, CTX=ctx if save else None, lse=lse if save else None,
                          attn_dropout_p=pa, salt_a=3, hidden_dropout_p=ph, salt_h=4, seed=seed)
        for dn, d in (("", None), ("+dirty", dirty)):
            tc, tf = timeit(comp, dirty=d), timeit(fused, dirty=d)
            print(f"{nm:34s} {mode + dn:>8s} {tc:10.1f} {tf:8.1f} {flops / tf / 1e6:11.1f}", flush=True)
