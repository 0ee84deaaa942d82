"""Is the GPU idle between the last backward kernel and the optimiser?  Events around eng.backward() / opt.step() inside an otherwise
normal sequence of steps (no profiler)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
batches = [bench.synthetic_batch(128, 128, 42 + i, dev) for i in range(4)]
eng, opt = model.engine, model.optimizers()
marks = []
orig_bwd, orig_step = eng.backward, opt.step


def bwd():
    orig_bwd()
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("bwd_end", e))


def step(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("opt_begin", e))
    orig_step(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("opt_end", e))


eng.backward, opt.step = bwd, step
for i in range(12):
    b = batches[i % 4]
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("step_begin", e))
    model.fused_step(*b[:3], 0.4, n_tokens=b[3])
torch.cuda.synchronize()
st = torch.cuda.memory_stats()
print("alloc retries", st.get("num_alloc_retries"), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 1))
by = {}
for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
    by.setdefault(f"{n0}->{n1}", []).append(e0.elapsed_time(e1))
for k, v in by.items():
    v = v[4:]
    print(f"{k:24s} median {sorted(v)[len(v)//2]:8.3f} ms   max {max(v):8.3f}")
