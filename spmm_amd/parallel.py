"""Data parallelism over RCCL/xGMI (torch.distributed backend "nccl" on ROCm), one process per GPU.

The path shards by batch rows only: every rank holds full weights and the full queue and runs the
whole step on its B_local rows.  Two exchanges per step:
  C1  all-gather of the momentum features before the enqueue (concat_all_gather SPMM_models.py:390-399, called from
      _dequeue_and_enqueue :273-274) -- one fused [B_local, E] gather per feature kind;
  C2  gradient averaging over the flat fp32 arena (what DDPStrategy does implicitly, SPMM_pretrain.py:36), in a few large
      buckets so each xGMI link carries long messages.
Buffers are NOT broadcast every step (DDP's broadcast_buffers, C3 in SURVEY.md): queues stay replica-identical because
every rank enqueues the same gathered features; `assert_replicas_identical` checks that."""
from __future__ import annotations

import torch
import torch.distributed as dist

BUCKET_ELEMS = 32 * 1024 * 1024      # 128 MiB of fp32 per all-reduce


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_gather_features(t: torch.Tensor) -> torch.Tensor:
    ws = world()
    if ws == 1:
        return t
    t = t.contiguous()
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out


def allreduce_mean_(flat: torch.Tensor, bucket_elems: int = BUCKET_ELEMS) -> torch.Tensor:
    """In-place mean over ranks of a flat gradient buffer, bucketed."""
    ws = world()
    if ws == 1:
        return flat
    use_avg = dist.get_backend() == "nccl"
    n = flat.numel()
    for lo in range(0, n, bucket_elems):
        chunk = flat[lo:min(n, lo + bucket_elems)]
        if use_avg:
            dist.all_reduce(chunk, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
            chunk.div_(ws)
    return flat


def grad_sync_fn():
    return allreduce_mean_ if world() > 1 else None


def broadcast_state_(tensors, src: int = 0):
    """C4: one-time broadcast of parameters and buffers after construction / checkpoint load."""
    if world() == 1:
        return
    for t in tensors:
        dist.broadcast(t, src=src)


def assert_replicas_identical(t: torch.Tensor, what: str = "tensor"):
    """Debug check replacing DDP's per-step buffer broadcast: every rank must hold the same values."""
    if world() == 1:
        return
    ref = t.detach().clone()
    dist.broadcast(ref, src=0)
    if not torch.equal(ref, t):
        raise RuntimeError(f"{what} diverged across data-parallel replicas")
