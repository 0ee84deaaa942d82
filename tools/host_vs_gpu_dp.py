"""One-rank data-parallel step: host enqueue time per step against GPU time per step, with and without the start-up stream probe
(the 80-ms mode of EXPERIMENTS.md 1.4 / 2.7b).   SPMM_PROBE_STREAMS=0|1 python tools/host_vs_gpu_dp.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
torch.cuda.set_device(0)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from spmm_amd.config import BertConfig, SPMMConfig
from spmm_amd.model import SPMM
from spmm_amd.options import EngineOptions
from spmm_amd.parallel import grad_sync_fn
import spmm_oracle as O
cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True), prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1),
                 embed_dim=256, queue_size=36864)
sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20, 'cooldown_epochs': 0}
tc = {'embed_dim': 256, 'temp': 0.07, 'queue_size': 36864, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched, 'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
# SPMM_TOUCH_ORDER="side0,side1,wgrad,rccl,x": first use of the pool streams / RCCL's stream / an unrelated extra stream in this order,
# before the model exists (which hardware queue a stream gets is decided at its first use)
from spmm_amd import streams as _streams
_extra = []
for name in [t for t in os.environ.get("SPMM_TOUCH_ORDER", "").split(",") if t]:
    if name == "rccl":
        torch.distributed.all_reduce(torch.zeros(4, device="cuda"), async_op=True).wait()
    elif name == "x":
        _extra.append(torch.cuda.Stream())
        with torch.cuda.stream(_extra[-1]):
            torch.zeros(8, device="cuda").add_(1.0)
    else:
        with torch.cuda.stream(_streams.get("cuda:0", name)):
            torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()
opts = EngineOptions.from_env(force_dist=True, schedule_check=False)
m = SPMM(config=tc, spmm_config=cfg, options=opts).train()
sync = grad_sync_fn(m.store, opts)
prop, ids, mask = O.synthetic_batch(128, 128, seed=42)
prop, ids, mask = prop.cuda(), ids.cuda(), mask.cuda()
nt = int(mask.sum())
for i in range(6):
    m.fused_step(prop, ids, mask, 0.4, n_tokens=nt, grad_sync=sync)
torch.cuda.synchronize()
host, N = [], 20
t00 = time.perf_counter()
for i in range(N):
    t0 = time.perf_counter()
    m.fused_step(prop, ids, mask, 0.4, n_tokens=nt, grad_sync=sync)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
tot = (time.perf_counter() - t00) * 1e3 / N
host.sort()
print(f"probe={os.environ.get('SPMM_PROBE_STREAMS', '0')} touch=[{os.environ.get('SPMM_TOUCH_ORDER', '')}]: wall {tot:.1f} ms per step; host time inside fused_step: median {host[N // 2]:.1f} ms, max {host[-1]:.1f} ms", flush=True)
# host enqueue time alone: queues drained before every step (no back-pressure from full hardware queues)
alone = []
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.fused_step(prop, ids, mask, 0.4, n_tokens=nt, grad_sync=sync)
    alone.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
alone.sort()
print(f"probe={os.environ.get('SPMM_PROBE_STREAMS', '0')}: host enqueue time of a step into drained queues: median {alone[4]:.1f} ms (min {alone[0]:.1f}, max {alone[-1]:.1f})", flush=True)
torch.distributed.destroy_process_group()
