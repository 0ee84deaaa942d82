"""The four losses of the product against the fp32 oracle and against the oracle's bf16 storage model at the FULL benchmark batch
(B = 128, Lt = 128, 12+6 layers, queue 36 864), recorded draws, dropout off.  The test suite runs the same comparison at B = 32
(tests/test_step_gpu.py::test_losses_match_the_bf16_storage_model_of_the_oracle); this is the one-off at B = 128 (~10 minutes of host time).
Imports oracle/ as the checker (test infrastructure), like the tests do."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import spmm_oracle as O
from spmm_amd.config import BertConfig, SPMMConfig
from spmm_amd.model import SPMM
B, Lt = int(os.environ.get("PARITY_B", "128")), 128
t = BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True)
p = BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1)
cfg = SPMMConfig(text=t, prop=p, embed_dim=256, queue_size=36864)
ocfg = O.full_cfg(); ocfg.queue_size = 36864
sd = O.init_state_dict(ocfg, seed=13)
prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
neg = (torch.arange(B).roll(1), torch.arange(B).roll(7))
torch.set_num_threads(min(64, os.cpu_count() or 8))
m = SPMM(config=None, spmm_config=cfg); m.load_state_dict({k: v.clone() for k, v in sd.items()}); m.eval()
with torch.no_grad():
    got = np.array([float(x) for x in m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=(neg[0].cuda(), neg[1].cuda()))])
    t0 = time.time()
    ref32 = np.array([float(x) for x in O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
    t1 = time.time()
    with O.bf16_storage():
        ref16 = np.array([float(x) for x in O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
print(f"B = {B}, Lt = {Lt}, full depth, queue 36864 (oracle forward {t1 - t0:.0f} s on {torch.get_num_threads()} threads)")
print("losses (mlm, 5 mpm, ita, itm)")
print("  product                         ", got)
print("  fp32 oracle                     ", ref32)
print("  bf16 storage model of the oracle", ref16)
print("  |product - fp32 oracle|         ", np.abs(got - ref32))
print("  |storage model - fp32 oracle|   ", np.abs(ref16 - ref32))
print("  |product - storage model|       ", np.abs(got - ref16))
