"""Per-step wall time of the first steps of a fresh process (allocator pool growth), three streams then one stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from spmm_amd.config import BertConfig, SPMMConfig
from spmm_amd.model import SPMM
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import spmm_oracle as O
# SPMM_TOUCH_ORDER="wgrad,side0,x": first use of the pool streams (x = an unrelated stream) in this order before the model exists
from spmm_amd import streams as _streams
_extra = []
for name in [t for t in os.environ.get("SPMM_TOUCH_ORDER", "").split(",") if t]:
    st = torch.cuda.Stream() if name == "x" else _streams.get("cuda:0", name)
    _extra.append(st)
    with torch.cuda.stream(st):
        torch.zeros(8, device="cuda").add_(1.0)
    torch.cuda.synchronize()
torch.manual_seed(0)
cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True), prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1),
                 embed_dim=256, queue_size=36864)
sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20, 'cooldown_epochs': 0}
tc = {'embed_dim': 256, 'temp': 0.07, 'queue_size': 36864, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched, 'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
sync = None
if os.environ.get("SPMM_FORCE_DIST") == "1":         # the data-parallel code path with a one-rank RCCL group
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
m = SPMM(config=tc, spmm_config=cfg).train()
if os.environ.get("SPMM_FORCE_DIST") == "1":
    from spmm_amd.parallel import grad_sync_fn
    m.engine.opt = m.engine.opt.replace(schedule_check=False)
    sync = grad_sync_fn(m.store, m.engine.opt)
prop, ids, mask = O.synthetic_batch(128, 128, seed=42)
prop, ids, mask = prop.cuda(), ids.cuda(), mask.cuda()
nt = int(mask.sum())
for phase, one in (("three streams", False), ("one stream (streams)", True), ("three streams again", False)):
    m.engine.force_one_stream = one
    ts = []
    for i in range(int(os.environ.get("FIRST_STEPS_N", "5"))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.fused_step(prop, ids, mask, 0.4, n_tokens=nt, grad_sync=sync)
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(f"touch=[{os.environ.get('SPMM_TOUCH_ORDER', '')}]", phase, ts if len(ts) <= 6 else f"median of {len(ts)}: {sorted(ts)[len(ts) // 2]}", flush=True)
