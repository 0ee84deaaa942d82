"""Debug: wrap ops.attn_bwd inside a real tiny-model step and compare each call with torch autograd on the same tensors."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import spmm_oracle as O
from spmm_amd import ops
from spmm_amd.config import tiny_config
from spmm_amd.model import SPMM
from test_kernels_gpu import ref_attention

orig = ops.attn_bwd
def wrapped(Q, K, V, Oo, lse, dO, dQ, dK, dV, *, nseq, nH, Lq, Lkv, kmask=None, causal_from=None, is_cross=False, dropout_p=0.0, seed=None, salt=0):
    orig(Q, K, V, Oo, lse, dO, dQ, dK, dV, nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, kmask=kmask, causal_from=causal_from, is_cross=is_cross,
         dropout_p=dropout_p, seed=seed, salt=salt)
    H = nH * 64
    q = Q.float().reshape(nseq, Lq, H).clone().requires_grad_(True)
    k = K.float().reshape(nseq, Lkv, H).clone().requires_grad_(True)
    v = V.float().reshape(nseq, Lkv, H).clone().requires_grad_(True)
    cf = nseq if causal_from is None else causal_from
    with torch.enable_grad():
        ro, rl = ref_attention(q, k, v, kmask, nH, cf, is_cross)
        ro.backward(dO.float().reshape(nseq, Lq, H))
    def rel(a, b): return ((a.float().reshape(b.shape) - b).norm() / (b.norm() + 1e-12)).item()
    print(f"attn_bwd nseq={nseq} Lq={Lq} Lkv={Lkv} cf={cf} cross={is_cross}: O rel {rel(Oo, ro.detach()):.4f} lse {rel(lse, rl.detach()):.5f} "
          f"dQ {rel(dQ, q.grad):.4f} dK {rel(dK, k.grad):.4f} dV {rel(dV, v.grad):.4f} |dQ|={q.grad.norm():.4g} |dK|={k.grad.norm():.4g} |dV|={v.grad.norm():.4g} |dO|={dO.float().norm():.4g}")
ops.attn_bwd = wrapped

ocfg = O.tiny_cfg(); cfg = tiny_config()
for c in (cfg.text, cfg.prop): c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
sd = O.closed_form_state_dict(ocfg)
m = SPMM(config=None, spmm_config=cfg); m.load_state_dict(sd); m.train()
B, Lt = 8, 24
prop, ids, mask = O.synthetic_batch(B, Lt, seed=13)
mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(2))
neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
losses = m(prop, ids, mask, alpha=0.3, mpm_mask=mpm.cuda(), neg_idx=(neg[0].cuda(), neg[1].cuda()))
sum(losses).backward()
