"""How far ahead of the GPU does the host run?  Host time to ENQUEUE one training step vs the GPU time to execute it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spmm_amd.config import BertConfig, SPMMConfig
from spmm_amd.model import SPMM
dev = torch.device("cuda:0")
cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True),
                 prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1), embed_dim=256, queue_size=36864)
sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20, 'cooldown_epochs': 0}
tc = {'embed_dim': 256, 'temp': 0.07, 'mlm_probability': 0.15, 'queue_size': 36864, 'momentum': 0.995, 'alpha': 0.4,
      'schedular': sched, 'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
torch.manual_seed(42)
model = SPMM(config=tc, spmm_config=cfg, loader_len=1000).train()
model.store.refresh_shadows()
batch = bench.synthetic_batch(128, 128, 42, dev)
for _ in range(3):
    model.fused_step(*batch[:3], 0.4, n_tokens=batch[3])
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(6):
    a = time.perf_counter()
    model.fused_step(*batch[:3], 0.4, n_tokens=batch[3])
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 6
print("host enqueue ms per step:", [round(h * 1e3, 1) for h in host], " wall ms per step:", round(wall * 1e3, 1))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
model.fused_step(*batch[:3], 0.4, n_tokens=batch[3])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
