"""Wall-clock / timeline properties of the step on a real MI355X.  They sort LAST (test_zz_*) so that a timing assertion can never
hide a parity test under `pytest -x`; every threshold is derived from a calibration made inside the test, and each test runs ONCE, in a
FRESH child process (stream -> hardware-queue placement and clocks differ from process to process; round 6: single attempt)."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from helpers_gpu import _cuda, _tiny_train_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _in_fresh_child(test_name, attempts=1, **env):
    """Runs `test_name` of this file in a fresh process (GPU_MAX_HW_QUEUES=8 like bench.py / pretrain.py), once (`attempts` exists for
    local diagnosis only); a failure's output is kept in gpurun_out/timing_failures.txt.  Returns the captured output of the passing run."""
    last = None
    for a in range(attempts):
        out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", os.path.abspath(__file__) + "::" + test_name],
                             capture_output=True, text=True, timeout=900, cwd=ROOT,
                             env=dict(os.environ, SPMM_TIMING_CHILD="1", GPU_MAX_HW_QUEUES="8", **env))
        if out.returncode == 0:
            return out.stdout
        last = out
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "timing_failures.txt"), "a") as f:
            f.write(f"==== {test_name} attempt {a}\n" + out.stdout[-3000:] + out.stderr[-1000:] + "\n")
    raise AssertionError(f"{test_name} failed in {attempts} fresh processes:\n" + last.stdout[-4000:] + last.stderr[-2000:])


def test_gradient_exchange_overlaps_backward(env, monkeypatch):
    """Timeline of the overlapped gradient exchange (spmm_amd/parallel.py OverlappedGradSync) with the collective replaced by one that
    behaves like ProcessGroupNCCL -- it waits for the issuing stream, runs ~5 ms on its OWN stream, and `work.wait()` only makes the
    caller's stream wait -- so the check needs no second GPU: every layer's slice is issued exactly once, the backward of the
    NEXT layer finishes on the compute stream while this layer's (slow) reduce is still running, the compute stream joins
    only in finish(), and the step's results are those of the run without any exchange."""
    if os.environ.get("SPMM_TIMING_CHILD") != "1":
        out = _in_fresh_child("test_gradient_exchange_overlaps_backward")
        probe = [l for l in out.splitlines() if l.startswith("[overlap-probe]")]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "overlap_probe.txt"), "a") as f:
            f.write("\n".join(probe) + "\n")
        return
    from spmm_amd import parallel, streams
    O = env[0]
    prop, ids, mask = O.synthetic_batch(4, 16, seed=7)
    prop, ids, mask = _cuda(prop, ids, mask)
    mpm = torch.zeros(4, 53).cuda()
    neg = tuple(_cuda(torch.arange(4).roll(1), torch.arange(4).roll(2)))
    ref = _tiny_train_model(env, dropout=False)
    for _ in range(2):                                            # (the measured step below is the model's SECOND: the first one warms everything up)
        want = [float(x) for x in ref.fused_step(prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)]
    want_flat = ref.store.flat.clone()

    # Stream -> hardware-queue placement, as the product checks it at data-parallel start-up (spmm_amd/streams.py): the stand-in
    # for RCCL's stream must not serialise with any compute stream; one that does is re-drawn, and what was seen is logged.
    comm = torch.cuda.Stream()
    compute = {"current": torch.cuda.current_stream(), **{n: streams.get("cuda:0", n) for n in ("side0", "side1", "wgrad")}}
    for attempt in range(6):
        clash = [n for n, s in compute.items() if streams.probe_pair(s, comm)]
        print(f"[overlap-probe] attempt {attempt}: comm stream {comm.cuda_stream:#x} vs {({n: hex(s.cuda_stream) for n, s in compute.items()})}: "
              f"{'serialises with ' + ','.join(clash) if clash else 'independent of all compute streams'} "
              f"(GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')})", flush=True)
        if not clash:
            break
        comm = torch.cuda.Stream()
    assert not clash, f"no communication stream independent of {clash} found"
    issued_from, records, posts, host_t = [], [], [], []
    scratch = torch.zeros(64, device="cuda")
    sleep_cycles = [1000]                                         # the warm-up step's collectives are short
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    SLEEP = 40_000_000                                           # ~20 ms: many times a tiny layer's (host-bound) backward (at ~10 ms one attempt in ~20 lost to a host hiccup)
    torch.cuda._sleep(1000)
    c0.record(); torch.cuda._sleep(SLEEP); c1.record()
    torch.cuda.synchronize()
    link_ms = c0.elapsed_time(c1)                                # what the fake collective below costs (a few ms)

    class Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    m = _tiny_train_model(env, dropout=False)

    def slow_all_reduce(t, op=None, async_op=False):
        # The stand-in owns BOTH timestamps of a slice: `issue` on the stream the product issues the collective from (behind the last
        # writer of the slice), `done` on its own stream.  Nothing else waits for `done` before finish() (rounds 3-5 let the product park an
        # "observer" stream behind every collective to take that timestamp; it is gone -- and it was not what made GPUTEST_r05 red, see below).
        assert async_op
        issued_from.append(torch.cuda.current_stream().cuda_stream)
        host_t.append(time.perf_counter())
        issue = torch.cuda.Event(enable_timing=True)
        issue.record()
        with torch.cuda.stream(comm):
            comm.wait_event(issue)
            torch.cuda._sleep(sleep_cycles[0])                  # "link time"
            t.mul_(1.0)                                          # one rank: the mean is the value itself
            done = torch.cuda.Event(enable_timing=True)
            done.record()
        lo = (t.data_ptr() - m.store.grad.data_ptr()) // t.element_size()
        records.append((lo, lo + t.numel(), issue, done))
        # diagnostic timestamps on the issuing stream: right behind the issue, and behind one more tiny kernel
        p1 = torch.cuda.Event(enable_timing=True); p1.record()
        scratch.add_(1.0)
        p2 = torch.cuda.Event(enable_timing=True); p2.record()
        posts.append((issue, p1, p2))
        return Work(done)

    monkeypatch.setattr(parallel.dist, "all_reduce", slow_all_reduce)
    monkeypatch.setattr(parallel.dist, "get_backend", lambda *a: "nccl")
    sync = parallel.OverlappedGradSync(m.store.order, m.store.offset, m.store.total, wire="fp32")
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # Step 1 = warm-up THROUGH the stand-in collective: round 6 found the host stalling 40-90 ms inside the stand-in's first call (lazy
    # code-object load of its `mul_` kernel; 9 ms on a good day) in 3 of 25 fresh processes -- the 20-ms "link time" was then over before
    # the host issued the next layer's backward, and the timeline read "no overlap" (GPUTEST_r05's red test; gpurun_out/r06_2).
    m.fused_step(prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, grad_sync=sync)
    torch.cuda.synchronize()
    for l in (issued_from, records, posts, host_t):
        l.clear()
    sleep_cycles[0] = SLEEP
    t0.record()
    got = m.fused_step(prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, grad_sync=sync)
    t1.record()
    torch.cuda.synchronize()
    got = [float(x) for x in got]
    trace = list(records)
    print("[overlap-probe] per slice, ms from its issue to (the issuing stream's next event, one tiny kernel later): "
          + " ".join(f"({a.elapsed_time(b):.2f} {a.elapsed_time(c):.2f})" for a, b, c in posts), f"issued from {[hex(x) for x in issued_from]}", flush=True)
    nl = m.cfg.text.num_hidden_layers + m.cfg.prop.num_hidden_layers
    layer_slices = trace[:nl]                                    # per-layer slices come first, the sweep of the rest follows
    assert len(trace) > nl and len(issued_from) == len(trace)
    covered = sorted((lo, hi) for lo, hi, _, _ in trace)
    assert covered[0][0] == 0 and covered[-1][1] == m.store.total
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))            # every element exactly once
    gaps, host_gaps = [], []
    for i, ((lo, hi, issue, done), (_, _, issue_next, _)) in enumerate(zip(layer_slices, layer_slices[1:])):
        assert issue.elapsed_time(done) >= 0.8 * link_ms                       # the fake collective really takes its time
        gaps.append(issue_next.elapsed_time(done))
        host_gaps.append((host_t[i + 1] - host_t[i]) * 1e3)
    print("[overlap-probe] ms between the NEXT layer's issue and this layer's done (positive = overlapped): " + " ".join(f"{g:.2f}" for g in gaps)
          + " | host ms between the two issues: " + " ".join(f"{g:.2f}" for g in host_gaps) + f" | link {link_ms:.1f} ms", flush=True)
    # A pair only counts when the HOST itself reached the next issue well inside the link time (a tiny model's backward is host-bound: a
    # host that takes longer than the collective proves nothing about the streams); the warmed-up step leaves every pair countable.
    counted = [g for g, h in zip(gaps, host_gaps) if h < 0.5 * link_ms]
    assert counted, f"the host needed {host_gaps} ms between issues against a {link_ms:.1f}-ms collective: nothing to judge"
    assert min(counted) > 0.0, f"the next layer's backward did not finish before this layer's reduce ended: no overlap ({gaps}; host {host_gaps})"
    # all slices ran back to back on the communication stream; the compute stream joined once, at the end
    assert t0.elapsed_time(t1) < len(trace) * link_ms * 1.5 + 200.0
    np.testing.assert_allclose(got, want, rtol=2e-4)             # (fp32 atomic sums: two runs of two steps agree to rounding, not bit for bit)
    assert (m.store.flat - want_flat).abs().max().item() < 5e-3              # two AdamW steps at lr 1e-3: sign flips of ~0 gradients move a weight by <= 2 lr each


def test_hipgraph_replay_host_cost(env):
    """Host side of `fused_step_graphed`: replaying the captured step must cost a small fraction of enqueueing it eagerly (the
    reason the mode exists).  Calibration = the eager step's own host time on this box, same process."""
    if os.environ.get("SPMM_TIMING_CHILD") != "1":
        _in_fresh_child("test_hipgraph_replay_host_cost")
        return
    O = env[0]
    prop, ids, mask = _cuda(*O.synthetic_batch(8, 24, seed=100))
    eager, graphed = _tiny_train_model(env, dropout=False), _tiny_train_model(env, dropout=False)
    eager.engine.pack_text = False
    t_e, t_g = [], []
    for i in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eager.fused_step(prop, ids, mask, 0.2)
        t_e.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        graphed.fused_step_graphed(prop, ids, mask, 0.2)
        t_g.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    med_e, med_g = sorted(t_e[2:])[3], sorted(t_g[2:])[3]          # steps 3..8 replay; the first two warm up / capture
    print(f"host ms per step: eager {med_e * 1e3:.2f}, graph replay {med_g * 1e3:.2f}")
    assert med_g < 0.5 * med_e, (t_e, t_g)


def test_data_parallel_code_path_costs_little_on_one_gpu(env):
    """The benchmark step through the N>1 code path with a one-rank RCCL group -- per-layer exchanges issued during the backward,
    RCCL's own stream, its reduce kernel per slice (with one rank the mean all-reduce is still a kernel) -- against the plain step on
    the same GPU, same process conditions.  Guards the schedule of DESIGN.md section 6: with the asynchronous weight-gradient stream left
    running beside the exchange this ratio was 1.32 (78 vs 59 ms); it is 1.04-1.06 as shipped.  Bound: 1.15 of the plain step measured
    in the same test (every measurement is a fresh child process; one repeat of the pair)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-kernel-timing"]
    for attempt in range(2):
        ms = _dp_cost_pair(cmd)
        if ms["1"] < 1.15 * ms["0"]:
            break
    assert ms["1"] < 1.15 * ms["0"], f"data-parallel code path {ms['1']:.1f} ms vs plain step {ms['0']:.1f} ms"


def _dp_cost_pair(cmd):
    ms = {}
    for force in ("1", "0"):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        envv = dict(os.environ, SPMM_FORCE_DIST=force, MASTER_PORT=str(port), SPMM_BENCH_WATCHDOG="400")
        for k in ("GPU_MAX_HW_QUEUES", "SPMM_NT_UNDER_COMM", "SPMM_GRAD_OVERLAP", "SPMM_STREAMS"):
            envv.pop(k, None)
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=envv, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        js = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert all(np.isfinite(js["losses"]))
        ms[force] = js["step_ms"]["median"]
    return ms
