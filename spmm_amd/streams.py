"""Process-wide HIP streams of the step, created once, in a fixed order, and checked for hardware-queue collisions.

HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default; bench.py / pretrain.py ask for 8) and streams that
share a queue SERIALISE.  Which queue a stream gets depends on how many streams the process created before it, so every model of a
process draws its side streams from this one pool (two models = the same three streams, not six) in a fixed order: the names below.
A data-parallel run adds RCCL's internal stream (the one `async_op=True` collectives really run on), which this package cannot
create or see -- `place_beside_collectives` PROBES it: a long sleep kernel on a compute stream, a tiny collective issued from
an idle stream, and host-side polling of both completions; a collective that only completes after the sleep shares the compute
stream's queue, and that compute stream is re-drawn (at most a few times) until the probe passes.

The probe is a DIAGNOSTIC (EngineOptions.probe_streams, off by default).  Measured with a one-rank RCCL group (profiles/
r03_probe_variants.txt): the streams the probe itself touches change the order in which streams receive their hardware queues,
and some orders -- not explained by queue sharing: the probe reports every pair independent in them -- run the whole step at
80 ms instead of 60-61.  The order the product produces by itself (first collective = the start-up broadcast, then side0, side1)
is among the fast ones, so the product leaves it alone."""
from __future__ import annotations

import time
from typing import Dict, List, Optional

import torch

ORDER = ("side0", "side1", "wgrad")     # creation order = queue placement; never create them any other way

_pool: Dict[tuple, torch.cuda.Stream] = {}
_log: List[str] = []


def get(device, name: str) -> torch.cuda.Stream:
    """The process-wide stream `name` of `device` (all names before it in ORDER are created first)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if (idx, name) not in _pool:
        if name not in ORDER:
            raise KeyError(name)
        for n in ORDER[:ORDER.index(name) + 1]:
            if (idx, n) not in _pool:
                _pool[(idx, n)] = torch.cuda.Stream(device=idx)
    return _pool[(idx, name)]


def bind_in_order(device, names) -> None:
    """First use of the named pool streams, in this order (one trivial kernel each): a stream gets its hardware queue at its first
    use, there are only a handful of them, and which streams end up sharing one decides whether a data-parallel step takes 60 or 80 ms
    (EXPERIMENTS.md 2.7b, profiles/r04_stream_order.txt)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = torch.zeros(8, device=f"cuda:{idx}")
    for n in names:
        with torch.cuda.stream(get(idx, n)):
            t.add_(1.0)
    torch.cuda.synchronize(idx)


def handles(device=None) -> Dict[str, int]:
    return {n: s.cuda_stream for (i, n), s in _pool.items() if device is None or i == torch.device(device).index}


def log() -> List[str]:
    return list(_log)


def note(line: str) -> None:
    """Start-up facts about stream placement that bench.py / pretrain.py report (`stream_placement`)."""
    _log.append(line)


def _spin_until(pred, seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        if pred():
            return True
    return pred()


def serialised(busy: torch.cuda.Stream, issue, sleep_ms: float = 8.0) -> Optional[bool]:
    """True when work started by `issue()` (called with `busy` NOT current; returns an object with `.query()` or `.is_completed()`)
    only completes after a `sleep_ms` sleep kernel on `busy` has finished, i.e. both sit on one hardware queue.  None when the
    sleep was too short to tell (the probe's own host time exceeded it)."""
    cycles = int(sleep_ms * 2.0e6)                   # torch.cuda._sleep counts device clocks (~2 GHz): a few ms
    end = torch.cuda.Event()
    with torch.cuda.stream(busy):
        torch.cuda._sleep(cycles)
        end.record()
    w = issue()
    done = (lambda: w.is_completed()) if hasattr(w, "is_completed") else (lambda: w.query())
    if end.query():                                  # the sleep was over before the first look at the work: nothing can be told
        end.synchronize()
        _spin_until(done, 5.0)
        return None
    finished_first = _spin_until(lambda: done() or end.query(), 5.0) and done() and not end.query()
    end.synchronize()
    _spin_until(done, 5.0)
    if finished_first:
        return False
    return True


def probe_pair(a: torch.cuda.Stream, b: torch.cuda.Stream) -> bool:
    """Do two of our own streams serialise?  (a tiny kernel on `b` while `a` sleeps)"""
    x = torch.zeros(64, device=f"cuda:{a.device.index}")
    torch.cuda.synchronize()

    def issue():
        ev = torch.cuda.Event()
        with torch.cuda.stream(b):
            x.add_(1.0)
            ev.record()
        return ev
    for sleep_ms in (8.0, 32.0, 128.0):              # (None = the host was slower than the sleep: look again with a longer one)
        r = serialised(a, issue, sleep_ms)
        if r is not None:
            return r
    return False


def place_beside_collectives(device, names=("side0", "side1", "wgrad"), max_redraws: int = 6) -> Dict[str, str]:
    """Data-parallel start-up: make sure none of the compute streams (the caller's current stream included) shares a hardware
    queue with the stream RCCL runs asynchronous collectives on.  Returns {stream name: "ok" | "redrawn xN" | "COLLIDES"}; the
    current stream cannot be re-drawn, a collision there is reported (and printed once)."""
    import torch.distributed as dist
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = torch.zeros(256, device=f"cuda:{idx}")
    idle = torch.cuda.Stream(device=idx)             # the collective's implicit dependency: a stream with nothing on it

    def issue():
        with torch.cuda.stream(idle):
            return dist.all_reduce(t, async_op=True)

    torch.cuda.synchronize()
    with torch.cuda.stream(idle):
        dist.all_reduce(t)                           # communicator and RCCL's stream exist from here on
    torch.cuda.synchronize()
    out = {}
    cur = torch.cuda.current_stream(idx)
    out["current"] = "COLLIDES" if serialised(cur, issue) else "ok"
    for n in names:
        s, redraws = get(idx, n), 0
        while serialised(s, issue) and redraws < max_redraws:
            s = _pool[(idx, n)] = torch.cuda.Stream(device=idx)
            redraws += 1
        out[n] = "ok" if redraws == 0 else (f"redrawn x{redraws}" if not serialised(s, issue) else "COLLIDES")
    torch.cuda.synchronize()
    msg = f"stream placement vs RCCL (device {idx}): {out}; handles {handles(idx)}"
    _log.append(msg)
    if any(v == "COLLIDES" for v in out.values()):
        import sys
        print("[spmm_amd.streams] WARNING " + msg + " -- the gradient exchange will not overlap that stream's kernels "
              "(raise GPU_MAX_HW_QUEUES before the HIP runtime initialises)", file=sys.stderr, flush=True)
    return out
