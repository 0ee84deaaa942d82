"""Stability check: a few hundred optimiser steps of the full-size model on four fixed synthetic batches (train mode, dropout,
packed text passes, multi-stream schedule).  The model can only memorise them, so every loss should fall and nothing should go
non-finite.  Prints the four losses, the gradient norm and temp every `--every` steps."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spmm_amd.config import BertConfig, SPMMConfig
from spmm_amd.model import SPMM
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--every", type=int, default=25)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--seq-len", type=int, default=128, help="256: BASELINE configs[4]'s sequence length (the attention kernels' long-key forms)")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True),
                 prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1), embed_dim=256, queue_size=36864)
sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20, 'cooldown_epochs': 0}
tc = {'embed_dim': 256, 'temp': 0.07, 'mlm_probability': 0.15, 'queue_size': 36864, 'momentum': 0.995, 'alpha': 0.4,
      'schedular': sched, 'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
torch.manual_seed(42)
model = SPMM(config=tc, spmm_config=cfg, loader_len=1000).train()
model.store.refresh_shadows()
batches = [bench.synthetic_batch(a.batch, a.seq_len, 42 + i, dev) for i in range(4)]
opt = model.optimizers()
t0 = time.time()
print(f"{'step':>5s} {'mlm':>9s} {'5*mpm':>9s} {'ita':>9s} {'itm':>9s} {'|g|':>10s} {'temp':>7s}")
for s in range(a.steps):
    alpha = 0.4 * min(1.0, s / 1000)
    losses = model.fused_step(*batches[s % 4][:3], alpha, n_tokens=batches[s % 4][3])
    if s % a.every == 0 or s == a.steps - 1:
        l = losses.cpu().tolist()
        assert all(x == x and abs(x) < 1e6 for x in l), (s, l)
        print(f"{s:5d} {l[0]:9.4f} {l[1]:9.4f} {l[2]:9.4f} {l[3]:9.4f} {float(opt.grad_norm):10.3f} {float(model.temp):7.4f}"
              f"   alloc {torch.cuda.memory_allocated() / 2**30:6.1f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:6.1f} GiB", flush=True)
torch.cuda.synchronize()
print(f"{a.steps} steps in {time.time() - t0:.1f} s; nan flag {int(model.engine.nan_flag)}; queue_ptr {int(model.queue_ptr)}")
