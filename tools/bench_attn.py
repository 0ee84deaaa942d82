"""Attention kernels at the S6 shapes with and without probability dropout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
BF = torch.bfloat16
dev = "cuda"
H, nH = 768, 12
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"{'shape':26s} {'fwd p=0':>9s} {'fwd p=.1':>9s} {'bwd p=0':>9s} {'bwd p=.1':>9s}   (us)")
for nm, nseq, Lq, Lkv, cross in (("self PV 54x54", 512, 54, 54, False), ("self text 128x128", 512, 128, 128, False),
                                 ("cross PV->text 54x128", 512, 54, 128, True), ("cross text->PV 128x54", 512, 128, 54, True),
                                 ("self text 96x96", 512, 96, 96, False), ("self text 256x256", 512, 256, 256, False),
                                 ("self text 256x256 causal", 512, 256, 256, "causal"), ("self text 128x128 causal", 512, 128, 128, "causal"),
                                 ("cross PV->text 54x256", 512, 54, 256, True), ("cross text->PV 256x54", 512, 256, 54, True)):
    causal = cross == "causal"
    cross = cross is True
    q = torch.randn(nseq * Lq, 3 * H, device=dev).to(BF); kv = torch.randn(nseq * Lkv, 2 * H, device=dev).to(BF)
    Q, K, V = (q[:, :H], kv[:, :H], kv[:, H:]) if cross else (q[:, :H], q[:, H:2 * H], q[:, 2 * H:])
    O = torch.empty(nseq * Lq, H, device=dev, dtype=BF); lse = torch.empty(nseq, nH, Lq, device=dev)
    dO = torch.randn(nseq * Lq, H, device=dev).to(BF); dQ = torch.empty_like(O); dKV = torch.empty(nseq * Lkv, 2 * H, device=dev, dtype=BF)
    out = []
    if max(Lq, Lkv) > 128 and os.environ.get("SPMM_HIP_LIB", "").endswith("oldattn.so"):
        continue
    for p in (0.0, 0.1):
        kw = dict(nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, is_cross=cross, dropout_p=p, seed=seed, salt=3, causal_from=0 if causal else None)
        out.append(timeit(lambda: ops.attn_fwd(Q, K, V, O, lse, **kw)))
        out.append(timeit(lambda: ops.attn_bwd(Q, K, V, O, lse, dO, dQ, dKV[:, :H], dKV[:, H:], **kw)))
    print(f"{nm:26s} {out[0]:9.1f} {out[2]:9.1f} {out[1]:9.1f} {out[3]:9.1f}", flush=True)
