"""Phase breakdown of the fused cross-attention kernel from s_memtime stamps (build/libspmm_hip_xaprof.so, make -C tools):
   SPMM_HIP_LIB=build/libspmm_hip_xaprof.so python tools/prof_xattn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import ops
BF = torch.bfloat16
dev = "cuda"
H, nH = 768, 12
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)
for nm, nseq, U, Lq, Lkv in (("PV -> text 54x128", 512, 128, 54, 128), ("text -> PV 128x54", 512, 128, 128, 54)):
    M = nseq * Lq
    Q = torch.randn(M, H, device=dev).to(BF); KV = torch.randn(U * Lkv, 2 * H, device=dev).to(BF); R = torch.randn(M, H, device=dev).to(BF)
    Wo = (torch.randn(H, H, device=dev) * 0.03).to(BF); bo = torch.zeros(H, device=dev); gm = torch.ones(H, device=dev); bt = torch.zeros(H, device=dev)
    idx = (torch.arange(nseq, device=dev) % U).to(torch.int32)
    WoF = ops.xattn_pack_wo(Wo)
    ctx, y, z = (torch.empty(M, H, device=dev, dtype=BF) for _ in range(3))
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    npan = nseq * ((Lq + 63) // 64)
    for mode, pa, ph, save in (("train", 0.1, 0.1, True), ("eval", 0.0, 0.0, False)):
        prof = torch.zeros(npan * 32, dtype=torch.int64, device=dev)
        for _ in range(3):
            prof.zero_()
            ops.xattn_fwd(Q, KV[:, :H], KV[:, H:], WoF, bo, R, gm, bt, y, nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, kv_seq=idx, Z=z if save else None,
                          mean=mean if save else None, rstd=rstd if save else None, CTX=ctx if save else None, lse=prof.view(torch.float32),
                          attn_dropout_p=pa, salt_a=3, hidden_dropout_p=ph, salt_h=4, seed=seed)
        torch.cuda.synchronize()
        t = prof.view(npan, 32).double().cpu()
        t = t[t[:, 23] > 0]
        d = lambda a, b: float((t[:, b] - t[:, a]).mean())
        steps_att = sum(d(1 + 2 * s, 2 + 2 * s) for s in range(6)) / 6
        steps_proj = sum(d(2 + 2 * s, 3 + 2 * s) for s in range(6)) / 6
        print(f"{nm} {mode}: panels {len(t)}  cycles (s_memtime ticks, 100 MHz?) total {d(0, 23):.0f} | prologue {d(0, 1):.0f} | per step: attention {steps_att:.0f} "
              f"projection {steps_proj:.0f} | epilogue: bias -> LDS image {d(13, 20):.0f} LayerNorm rows (dropout, residual, statistics, z / y stores) {d(20, 23):.0f}")
        first = float(t[:, 0].min()); print(f"    launch span: first start -> last end {float(t[:, 23].max()) - first:.0f} ticks; per-step detail (panel 0): "
              + " ".join(f"{int(t[0, i + 1] - t[0, i])}" for i in range(0, 13)))
