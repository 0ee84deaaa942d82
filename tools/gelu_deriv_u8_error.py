"""What storing gelu'(x) in 8 bits would do to the gradients (EXPERIMENTS.md 1.5: the option was not built; this measures its error).
The forward keeps gelu' for the FFN backward (SPMM_EPI_GELU_DERIV); here the saved tensor is replaced by its 8-bit image before the
backward runs -- linear code over the function's range [-0.1298, 1.1298] (step 4.9e-3) -- at full depth (12+6 layers, H=768, B=32,
Lt=128, dropout off, fixed draws), and the whole gradient is compared with the unquantised run of the same step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd.config import BertConfig, SPMMConfig
from spmm_amd.engine import Engine
from spmm_amd.model import SPMM

LO, HI = -0.1298, 1.1298


def q8(t):
    s = (HI - LO) / 255.0
    return (torch.round((t.float() - LO) / s).clamp_(0, 255) * s + LO).to(t.dtype)


cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True),
                 prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1), embed_dim=256, queue_size=36864)
for c in (cfg.text, cfg.prop):
    c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20, 'cooldown_epochs': 0}
tc = {'embed_dim': 256, 'temp': 0.07, 'queue_size': 36864, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched,
      'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
B, Lt = 32, 128
g = torch.Generator().manual_seed(42)
prop = torch.randn(B, 53, generator=g).cuda()
ids = torch.zeros(B, Lt, dtype=torch.long)
lens = torch.randint(Lt // 2, Lt + 1, (B,), generator=g); lens[0] = Lt
for b in range(B):
    n = int(lens[b]); ids[b, 0] = 2; ids[b, 1:n - 1] = torch.randint(4, 300, (n - 2,), generator=g); ids[b, n - 1] = 3
ids = ids.cuda(); mask = (ids != 0).long()
mpm = (torch.rand(B, 53, generator=g) < 0.5).float().cuda()
neg = (torch.arange(B).roll(1).cuda(), torch.arange(B).roll(7).cuda())

orig = Engine._layer_fwd
res = {}
for mode in ("bf16", "u8"):
    torch.manual_seed(0)
    m = SPMM(config=tc, spmm_config=cfg, loader_len=1000)
    m.train()
    if mode == "u8":
        def patched(self, *a, **k):
            y, sv, y32 = orig(self, *a, **k)
            if sv is not None and sv.get("dact") is not None:
                sv["dact"].copy_(q8(sv["dact"]))
            return y, sv, y32
        Engine._layer_fwd = patched
    eng = m.engine
    eng.alpha.fill_(0.4); eng.gscale.fill_(1.0); m.store.grad.zero_()
    losses = eng.forward(prop, ids, mask, mpm_mask=mpm, neg_idx=neg)
    eng.backward()
    res[mode] = (losses.clone(), m.store.grad.clone(), m)
    Engine._layer_fwd = orig
g0, g1 = res["bf16"][1], res["u8"][1]
print("losses (identical forwards):", res["bf16"][0].tolist())
print(f"whole gradient: relative L2 error of the u8-gelu' backward {((g1 - g0).norm() / g0.norm()).item():.3e}  (|g| = {g0.norm().item():.4f})")
st = res["bf16"][2].store
worst = []
for name in st.order:
    a = st._view(g0, name).flatten(); b = st._view(g1, name).flatten()
    if a.norm() > 0:
        worst.append(((b - a).norm() / a.norm()).item())
worst = sorted(worst)
print(f"per tensor: median {worst[len(worst) // 2]:.3e}, p90 {worst[int(0.9 * len(worst))]:.3e}, max {worst[-1]:.3e} over {len(worst)} tensors")
for name in ("text_encoder.bert.encoder.layer.11.intermediate.dense.weight", "text_encoder.bert.encoder.layer.6.intermediate.dense.weight",
             "text_encoder.bert.encoder.layer.0.intermediate.dense.weight", "property_encoder.encoder.layer.0.intermediate.dense.weight",
             "text_encoder.bert.embeddings.word_embeddings.weight"):
    a = st._view(g0, name).flatten(); b = st._view(g1, name).flatten()
    print(f"  {name}: {((b - a).norm() / a.norm()).item():.3e}")
print("(for scale: the bf16 pipeline's whole gradient deviates from the fp32 oracle's by < 1.5e-2, tests/test_step_gpu.py::test_gradients_match_oracle)")
