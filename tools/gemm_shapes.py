"""Per-shape time of every NT / TN GEMM launch of one training step (HIP events around each launch), sorted by total time."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spmm_amd import ops
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
for _ in range(2):
    model.fused_step(*batch[:3], 0.4, n_tokens=batch[3])
model.engine.multi_stream = False
model.engine.wgrad_async = False
ev = []
stream = torch.cuda.current_stream()
orig_nt, orig_tn = ops.gemm_nt, ops.gemm_tn

def wrap(kind, fn):
    def w(A, B, C, **k):
        if kind == "nt":
            M, N, K = A.shape[0], B.shape[0], (k.get("K") or A.shape[1])
            key = ("nt", M, N, K, k.get("epi", 0))
        else:
            M, N, K = A.shape[0], A.shape[1], B.shape[1]
            key = ("tn", M, N, K, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); r = fn(A, B, C, **k); e1.record(stream)
        ev.append((key, e0, e1, 2.0 * M * N * K))
        return r
    return w

ops.gemm_nt, ops.gemm_tn = wrap("nt", orig_nt), wrap("tn", orig_tn)
model.fused_step(*batch[:3], 0.4, n_tokens=batch[3])
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl in ev:
    t = e0.elapsed_time(e1)
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += t; a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"{'kind':4s} {'M':>7s} {'N':>6s} {'K':>6s} epi {'n':>4s} {'ms':>8s} {'%':>6s} {'TF/s':>7s}")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{key[0]:4s} {key[1]:7d} {key[2]:6d} {key[3]:6d} {key[4]:3d} {a[0]:4d} {a[1]:8.3f} {100 * a[1] / tot:6.2f} {a[2] / a[1] / 1e9:7.1f}")
print(f"total {tot:.2f} ms")
