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

from typing import Optional

import torch
import torch.distributed as dist

from .options import EngineOptions

BUCKET_ELEMS = 32 * 1024 * 1024      # 128 MiB of fp32 per all-reduce


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_gather_features(t: torch.Tensor) -> torch.Tensor:
    """C1.  With a one-rank group (EngineOptions.force_dist) the collective still runs: the code path is the N>1 one."""
    ws = world()
    if not (dist.is_available() and dist.is_initialized()):
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


class OverlappedGradSync:
    """C2 overlapped with backward (what DDP's bucketed hooks do for the reference, SPMM_pretrain.py:36).

    Every encoder layer's gradients are final the moment its backward returns -- each parameter range is written by exactly
    one backward stage (text layers f..n-1 by S6, 0..f-1 by S2, the PV encoder by S1; spmm_amd/step.py) -- and a layer's
    tensors are contiguous in the flat arena, so `layer_done(prefix)` launches one asynchronous all-reduce over that slice
    (28-38 MB fp32: long messages for the point-to-point xGMI links).  RCCL runs it on its own stream behind an event on the
    compute stream while the next layers' backward kernels keep the CUs busy.  `finish()` reduces what no layer covered
    (embeddings, heads, projections: ~2 % of the bytes) and joins the streams before the optimiser.
    The collective's implicit dependency is the stream context `layer_done` is called in: the engine calls it from the stream that
    is ordered behind every writer of the slice (the weight-gradient stream when there is one, behind an event on the stream the
    layer's backward ran on -- S2's backward runs on a side stream; Engine._layer_done)."""

    def __init__(self, order, offset, total, wire=None, options: Optional[EngineOptions] = None):
        self.total = int(total)
        opt = options if options is not None else EngineOptions.from_env()
        names = list(order)
        ends = [offset[n] for n in names[1:]] + [self.total]
        self._ranges = {}
        import re
        for n, e in zip(names, ends):
            m = re.match(r"(.*encoder\.layer\.\d+\.)", n)
            if not m:
                continue
            lo, hi = self._ranges.get(m.group(1), (offset[n], offset[n]))
            if offset[n] != hi:
                raise ValueError(f"parameters of {m.group(1)} are not contiguous in the flat arena")
            self._ranges[m.group(1)] = (lo, e)
        self._grad, self._work, self._done = None, [], []
        # wire format of the gradient exchange: "fp32" = one all-reduce per slice on the arena itself; "bf16" = the slice is cast
        # to bf16, reduce-scattered and all-gathered (half the bytes on every xGMI link, sums in bf16: |rel err| <= 2^-8 per element)
        self.wire = opt.grad_wire if wire is None else wire
        if self.wire not in ("fp32", "bf16"):
            raise ValueError(f"grad_wire must be fp32 or bf16, not {self.wire!r}")
        # NT GEMMs of the backward as one workgroup per tile while slices are in flight (see SPMM.fused_step)
        self.tiles_under_comm = opt.nt_under_comm != "persistent"
        self.nt_auto = opt.nt_under_comm == "auto"     # SPMM._schedule_check_begin times the persistent launch too and keeps the faster form
        # The exchange is the only stream beside the backward chain: the asynchronous weight-gradient stream rests meanwhile
        # (three chip-filling streams side by side ran the step in 76-78 ms against 61-62)
        self.exclusive = not opt.dp_four_streams      # (four streams: no two share a hardware slot, the weight-gradient stream keeps running)
        self.wait_events = None      # set to [] to record, per step, an event pair around finish()'s wait on the compute stream: the
        #                              communication time the backward did NOT hide (bench.py `comm_exposed_ms`)

    def begin(self, grad: torch.Tensor):
        assert grad.numel() == self.total
        self._grad, self._work, self._done, self._staged = grad, [], [], []
        self._avg = dist.get_backend() == "nccl"

    def _reduce(self, lo, hi):
        if hi <= lo:
            return
        sl = self._grad[lo:hi]
        if self.wire == "fp32":
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            works = [dist.all_reduce(sl, op=op, async_op=True)]
        else:
            ws, n = world(), hi - lo
            npad = (n + ws - 1) // ws * ws
            buf = torch.zeros(npad, dtype=torch.bfloat16, device=sl.device) if npad != n else torch.empty(n, dtype=torch.bfloat16, device=sl.device)
            buf[:n].copy_(sl)
            shard = torch.empty(npad // ws, dtype=torch.bfloat16, device=sl.device)
            w1 = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, async_op=True)
            if not self._avg:
                w1.wait()                                        # gloo runs independent works concurrently: order them by hand
            works = [w1, dist.all_gather_into_tensor(buf, shard, async_op=True)]      # RCCL: same stream, in issue order
            self._staged.append((lo, hi, buf, shard))           # both stay alive until finish(): RCCL reads / writes them on its stream
        self._work.extend(works)
        self._done.append((lo, hi))

    def layer_done(self, prefix: str):
        if self._grad is None or prefix not in self._ranges:
            return
        self._reduce(*self._ranges[prefix])

    def finish(self):
        pos = 0
        for lo, hi in sorted(self._done) + [(self.total, self.total)]:
            for a in range(pos, lo, BUCKET_ELEMS):
                self._reduce(a, min(lo, a + BUCKET_ELEMS))
            pos = max(pos, hi)
        timed = self.wait_events is not None and self._grad is not None and self._grad.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._work:
            w.wait()
        if timed:
            e1.record()
            self.wait_events.append((e0, e1))
        for lo, hi, buf, shard in self._staged:
            self._grad[lo:hi].copy_(buf[:hi - lo])
            if buf.is_cuda:                                      # allocated in the issuing stream's pool, last read here
                buf.record_stream(torch.cuda.current_stream())
        if not self._avg or self.wire == "bf16":
            self._grad.div_(world())
        self._grad, self._work, self._staged = None, [], []


_placed = set()


def grad_sync_fn(store=None, options: Optional[EngineOptions] = None):
    """None on a single rank; otherwise the overlapped reducer when the parameter layout is given, else the plain bucketed one.
    On RCCL the first call also checks (once per device) that the compute streams and RCCL's stream sit on different hardware
    queues (spmm_amd/streams.py)."""
    opt = options if options is not None else EngineOptions.from_env()
    live = dist.is_available() and dist.is_initialized()
    if not live or (world() == 1 and not opt.force_dist):
        return None
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        # Stream order is part of the schedule (EXPERIMENTS.md 1.4): RCCL's internal stream first, the compute side streams after it
        # is the order that measures 60-61 ms per step with a one-rank group; side streams first-touched before RCCL's ran at 80 ms.
        # An asynchronous collective here creates the communicator's stream before spmm_amd/streams.py hands out the first side
        # stream (they are created lazily, at the first forward); a process that already holds side streams is told so.
        from . import streams
        dev = torch.cuda.current_device()
        if ("order", dev) not in _placed:
            _placed.add(("order", dev))
            late = bool(streams.handles(dev))
            t = torch.zeros(1, device=f"cuda:{dev}")
            dist.all_reduce(t, async_op=True).wait()
            torch.cuda.synchronize(dev)
            if not late:
                # ... and the streams of a data-parallel step take their hardware queues NOW, in the order measured fast (RCCL's stream,
                # side0, side1, wgrad).  A stream gets its queue at its first use and there are only a handful: orders that make
                # RCCL's stream share one with a stream it exchanges dependencies with run the step at 79-87 ms instead of 60.5
                # (EXPERIMENTS.md 2.7b, profiles/r04_stream_order.txt).
                order = ("side0", "wgrad") if opt.dp_four_streams else ("side0", "side1", "wgrad")
                streams.bind_in_order(dev, order)
            streams.note(f"cuda:{dev}: first collective issued " + ("AFTER the compute side streams existed (an order EXPERIMENTS.md 2.7b measured slow is possible)"
                                                                     if late else f"first, then {', '.join(order)} bound to hardware queues in that order"))
            if late:
                import warnings
                warnings.warn("spmm_amd: HIP side streams were created before the first RCCL collective; create the process group and "
                              "call grad_sync_fn() before the first forward")
    if opt.probe_streams and dist.get_backend() == "nccl" and torch.cuda.is_available():
        dev = torch.cuda.current_device()
        if dev not in _placed:
            _placed.add(dev)
            from . import streams
            streams.place_beside_collectives(dev)
    if store is not None and opt.grad_overlap:
        return OverlappedGradSync(store.order, store.offset, store.total, options=opt)
    return allreduce_mean_


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
