"""AdamW / EMA / grad-norm kernels over a 144.4 M-parameter arena (us per launch, GB/s on algorithmic bytes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spmm_amd import ops
dev = "cuda"
n = 144_400_000 // 4 * 4
p = torch.randn(n, device=dev) * 0.02; gr = torch.randn(n, device=dev) * 1e-3; m1 = torch.zeros(n, device=dev); v1 = torch.zeros(n, device=dev)
sh = torch.empty(n, device=dev, dtype=torch.bfloat16); pm = p.clone(); shm = torch.empty_like(sh)
lr = torch.full((1,), 5e-5, device=dev); nsq = torch.zeros(1, device=dev); step = torch.zeros(1, dtype=torch.int32, device=dev); nan = torch.zeros(1, dtype=torch.int32, device=dev)
scal = torch.zeros(8, device=dev)
def t(fn, by, name, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:28s} {us:8.1f} us  {by / us / 1e3:7.0f} GB/s", flush=True)
def sq():
    nsq.zero_(); ops.grad_sqnorm(gr, nsq)
t(sq, n * 4, "grad sqnorm")
t(lambda: ops.adamw_step(p, gr, m1, v1, sh, lr=lr, normsq=nsq, step=step, nan_flag=nan, scalars=scal), n * 30, "adamw (+bf16 shadow)")
t(lambda: ops.ema_update(pm, p, shm, 0.995), n * 14, "ema (+bf16 shadow)")
