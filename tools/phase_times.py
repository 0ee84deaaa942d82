"""Where the phases of a step sit in time WITHOUT a tracer (rocprofv3 slows the host enough to distort the three-stream schedule):
HIP events around every encoder-stack pass (PretrainStep.stack_fwd / stack_bwd, per stream) and the optimiser of steady-state steps of the
benchmark shape; prints, for the last step, start / end of every phase relative to the step's first kernel, and the stream it ran on.
    python tools/phase_times.py [steps=12]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from spmm_amd.config import BertConfig, SPMMConfig
from spmm_amd.model import SPMM
from spmm_amd.options import EngineOptions
from spmm_amd.step import PretrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True),
                 prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1), embed_dim=256, queue_size=36864)
sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20, 'cooldown_epochs': 0}
tc = {'embed_dim': 256, 'temp': 0.07, 'mlm_probability': 0.15, 'queue_size': 36864, 'momentum': 0.995, 'alpha': 0.4,
      'schedular': sched, 'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
torch.manual_seed(42)
model = SPMM(config=tc, spmm_config=cfg, loader_len=1000, options=EngineOptions.from_env())
model.store.refresh_shadows(); model.engine.invalidate_banks(); model.train(True)
batches = [bench.synthetic_batch(128, 128, 42 + i, dev) for i in range(4)]
log = []

def wrap(name):
    orig = getattr(PretrainStep, name)
    def w(self, pfx, c, layers, *a, **k):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); r = orig(self, pfx, c, layers, *a, **k); e1.record(st)
        ls = list(layers)
        log.append((f"{name[6:]} {pfx} layers {ls[0]}..{ls[-1]}" if ls else f"{name[6:]} {pfx} (none)", st.stream_id if hasattr(st, 'stream_id') else id(st), e0, e1))
        return r
    setattr(PretrainStep, name, w)
wrap("stack_fwd"); wrap("stack_bwd")


def wrap_any(name):             # PHASES_DETAIL=1: single layers outside the stacks (the top fusion layer), the heads
    orig = getattr(PretrainStep, name)
    def w(self, *a, **k):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); r = orig(self, *a, **k); e1.record(st)
        log.append((name + (" " + a[0] if a and isinstance(a[0], str) else ""), st.stream_id if hasattr(st, 'stream_id') else id(st), e0, e1))
        return r
    setattr(PretrainStep, name, w)
if os.environ.get("PHASES_DETAIL") == "1":
    for n_ in ("_layer_fwd", "_layer_bwd", "lm_head_fwd", "lm_head_bwd", "_s6_forward_cls", "_s6_backward_cls", "_banks"):
        wrap_any(n_)
if os.environ.get("PHASES_BOUNDARY") == "1":      # the step's head and tail: plan kernels, embeddings, whole forward / backward, joins, optimiser
    for n_ in ("_pack_plan", "embed_text", "embed_pv", "forward", "backward", "wgrad_join", "_embed_ln_bwd"):
        wrap_any(n_)
    _o = model.optimizers()
    _orig_step = _o.step
    def _step(*a, **k):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); r = _orig_step(*a, **k); e1.record(st)
        log.append(("optimiser step (sqnorm, AdamW, forward-side shadows; off-path launches excluded)", st.stream_id, e0, e1))
        return r
    _o.step = _step
opt = model.optimizer if hasattr(model, "optimizer") else None
marks = []
for i in range(steps):
    prop, ids, mask, ntok = batches[i % 4]
    del log[:]
    s0 = torch.cuda.Event(enable_timing=True); s0.record(torch.cuda.current_stream())
    model.fused_step(prop, ids, mask, 0.4, n_tokens=ntok)
    s1 = torch.cuda.Event(enable_timing=True); s1.record(torch.cuda.current_stream())
    marks.append((s0, s1, list(log)))
torch.cuda.synchronize()
s0, s1, lg = marks[-1]
print(f"step (main stream, first to last launch): {s0.elapsed_time(s1):.2f} ms; previous steps: " + " ".join(f"{a.elapsed_time(b):.2f}" for a, b, _ in marks[-5:-1]))
ids_ = {}
for name, sid, e0, e1 in lg:
    ids_.setdefault(sid, len(ids_))
    print(f"  stream {ids_[sid]}  {s0.elapsed_time(e0):7.2f} .. {s0.elapsed_time(e1):7.2f} ms  ({e0.elapsed_time(e1):6.2f})  {name}")
