"""Headline benchmark: SPMM pretraining step throughput (molecules/s) on N MI355X, one process per GPU over RCCL.

  python bench.py                      (1 GPU, 50 timed steps after 10 warm-up steps: SURVEY.md section 8d protocol)
  python bench.py --gpus N             (starts `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` as a child)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = zero_grad + SPMM.forward (12 encoder passes, 4 losses) + backward + [gradient all-reduce + feature all-gather]
+ clip + AdamW + EMA on one synthetic batch of the pretrain shape (BASELINE.json configs[1]: full 12+6-layer / H=768 model,
per-GPU batch 128, seq_len 128, queue 36864, train mode with dropout, bf16 compute with fp32 accumulation).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto 4 hardware queues by default and streams sharing a queue serialise; the step uses five
# (main, two side streams, weight gradients, RCCL): ask for 8 before the runtime initialises.
# Not when several ranks share ONE GPU (the gloo test vehicle, tests/test_step_gpu.py): two processes x 8 queues oversubscribe the
# hardware scheduler -- the same 2-rank run takes 2 s with 4 queues per process, 12x longer with 16, and stalled for minutes with 8.
if os.environ.get("SPMM_DIST_BACKEND") != "gloo":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Watchdog: a run still going after this many seconds dumps every thread's Python stack and EXITS NON-ZERO (never a re-exec, never a
# silent hang of the driver's scaling run).  Default for N>1: 600 s -- 60 default steps take ~4 s, the rest is start-up (the first
# `import torch` of eight processes on a fresh box alone can take two minutes).
# (long runs get more: 2 s per step asked for beyond the default 50; SPMM_BENCH_WATCHDOG=0 turns it off)
def _asked_steps():
    for i, a in enumerate(sys.argv):
        if a == "--steps" and i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit():
            return int(sys.argv[i + 1])
        if a.startswith("--steps=") and a[8:].isdigit():
            return int(a[8:])
    return 50


_wd = os.environ.get("SPMM_BENCH_WATCHDOG") or (str(600 + 2 * max(0, _asked_steps() - 50)) if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "")
if _wd and float(_wd) > 0:
    import faulthandler
    faulthandler.dump_traceback_later(float(_wd), exit=True)
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF dense, 2495 TF measured)
PEAK_HBM_GBS = 8000.0


def smi_sampler(period_s=0.05, in_process_only=False):
    """Shader clock / socket power sampler (tools/smi_sampler.py: amdsmi, then rocm-smi; read-only).  Measurement plumbing only.
    Inside the headline timed region only the in-process amdsmi reader is allowed (a rocm-smi subprocess every 50 ms would perturb the
    host-enqueue-bound step it annotates); the device index goes through HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from smi_sampler import Sampler
    return Sampler(period_s=period_s, index=int(os.environ.get("LOCAL_RANK", "0")), in_process_only=in_process_only)


def synthetic_batch(B, Lt, seed, device):
    """SURVEY.md section 8d recipe (same as oracle.synthetic_batch; restated so the product never imports oracle/)."""
    g = torch.Generator().manual_seed(seed)
    prop = torch.randn(B, 53, generator=g)
    ids = torch.zeros(B, Lt, dtype=torch.long)
    lens = torch.randint(max(Lt // 2, 3), Lt + 1, (B,), generator=g)
    lens[0] = Lt
    for b in range(B):
        n = int(lens[b])
        ids[b, 0] = 2
        ids[b, 1:n - 1] = torch.randint(4, 300, (n - 2,), generator=g)
        ids[b, n - 1] = 3
    return prop.to(device), ids.to(device), (ids != 0).long().to(device), int(lens.sum())      # (token count: host metadata of the batch)


def step_flops(B, Lt, Lp=54, H=768, I=3072, V=300, E=256, Q=36864, n_text=12, fusion=6, n_pv=6):
    """Algorithmic FLOPs of one training step, SURVEY.md section 8d formulas (multiply-add = 2):
    3 x forward of the gradient-carrying passes + 1 x forward of the no-grad (momentum) passes."""
    def self_layer(tokens, L):
        return tokens * (8 * H * H + 4 * L * H + 4 * H * I)

    def fusion_layer(tq, Lq, tkv, Lkv):
        return tq * (12 * H * H + 4 * Lq * H + 4 * Lkv * H + 4 * H * I) + tkv * 4 * H * H

    nf = n_text - fusion
    tp, tt = B * Lp, B * Lt
    grad = (2 * n_pv * self_layer(tp, Lp)            # P1, P11
            + 2 * fusion * self_layer(tt, Lt)        # P2, P10a
            + nf * (fusion_layer(tp, Lp, tt, Lt) * 2 + fusion_layer(2 * tp, Lp, 2 * tt, Lt))      # P5, P12, P7
            + nf * (fusion_layer(tt, Lt, tp, Lp) * 2 + fusion_layer(2 * tt, Lt, 2 * tp, Lp))      # P6, P10b, P8
            + tt * (2 * H * H + 2 * H * V) + 4 * 2 * B * E * (B + Q))
    nograd = (n_pv * self_layer(tp, Lp) + 2 * fusion * self_layer(tt, Lt) + nf * fusion_layer(tt, Lt, tp, Lp)
              + tt * (2 * H * H + 2 * H * V) + 4 * 2 * B * E * (B + Q))
    return 3 * grad + nograd


def shared_kv_saving(B, Lt, Lp=54, H=768, n_text=12, fusion=6):
    """FLOPs the build does NOT execute: the seven student fusion passes cross-attend to only B unique text and B unique PV
    sequences, so the K/V projections (forward, data gradient, weight gradient) run on B*(Lt+Lp) source tokens per layer
    instead of the reference's 4B*(Lt+Lp) (spmm_amd/step.py, Group.share_kv).  Reported so that utilisation can be judged on
    executed work as well as on the reference's algorithmic work."""
    return 3 * (n_text - fusion) * 3 * B * (Lt + Lp) * 4 * H * H


def padding_saving(B, Lt, n_valid, Lp=54, H=768, I=3072, n_text=12, fusion=6):
    """FLOPs not executed because the passes that only read position 0 (P2, P4, P6, first half of P8) run on the n_valid
    real tokens of the batch instead of B*Lt (spmm_amd/step.py::_pack_plan); per-token costs as in step_flops."""
    pad = B * Lt - n_valid
    self_tok = 8 * H * H + 4 * Lt * H + 4 * H * I
    fus_tok = 12 * H * H + 4 * Lt * H + 4 * Lp * H + 4 * H * I
    # P2 (x3: forward + backward), P4 and P9a (teacher, forward only), P6 + P8a (x3), P9b (teacher fusion layers)
    return pad * (3 * fusion * self_tok + 2 * fusion * self_tok + 3 * 2 * (n_text - fusion) * fus_tok + (n_text - fusion) * fus_tok)


def cls_top_saving(B, Lt, n_valid, Lp=54, H=768, I=3072):
    """FLOPs not executed in the LAST fusion layer: the three ITM pass pairs feed only position 0 of their last hidden states to a loss
    (SPMM_models.py:199-201), so that layer runs its query / output / FFN work for those sequences on position 0 alone and keeps only
    the self-attention key / value projections of their other rows (spmm_amd/step.py::_s6_forward_cls).  Rows as executed: PV queries
    3B x Lp, packed text queries 2 x n_valid, dense text negatives B x Lt; forward + backward = 3 x forward, as in step_flops."""
    dead = 3 * B * (Lp - 1) + 2 * (n_valid - B) + B * (Lt - 1)
    gemm = dead * (8 * H * H + 4 * H * I)                      # self Q + self out + cross Q + cross out + FFN; self K / V stay
    core = 4 * H * (3 * B * (Lp - 1) * (Lp + Lt) + 3 * B * (Lt - 1) * (Lt + Lp))     # attention cores of the dead queries (upper bound: dense Lt)
    return 3 * (gemm + core)


def neg_pack_saving(B, Lt, n_valid, Lp=54, H=768, I=3072, n_text=12, fusion=6):
    """FLOPs not executed because the text hard negatives re-enter the fusion layers as PACKED query rows (their total length is device
    data: the launches over the batch take their row count from device memory, spmm_amd/step.py::_s6_forward_cls) instead of a dense
    [B, Lt] block with zero rows: per fusion layer below the top one, expected padding = B*Lt - n_valid rows (a negative is as long as an
    average sequence)."""
    pad = B * Lt - n_valid
    fus_tok = 12 * H * H + 4 * Lt * H + 4 * Lp * H + 4 * H * I
    return 3 * pad * (n_text - fusion - 1) * fus_tok


def cross_attn_unit_flops(nseq, Lq, Lkv, H=768):
    """Fused cross-attention unit (Q/K/V projections + core + out-proj), BASELINE.md section 3."""
    return nseq * (4 * H * H * Lq + 4 * H * H * Lkv + 4 * Lq * Lkv * H)


def cross_attn_executed_flops(groups, q_rows, H=768):
    """FLOPs the build executes for one cross-attention block: Q and output projections on the rows actually present (packed),
    K/V projections once per unique source row, the attention core per (query sequence, its key/value source)."""
    kv_rows, seen, core = 0, set(), 0.0
    for g in groups:
        if g.src is not None:                      # shared source: unique rows, counted once however many groups attend it
            if id(g.src) not in seen:
                seen.add(id(g.src))
                kv_rows += g.src.kv.shape[0]
        else:
            kv_rows += g.kv.shape[0] if g.kv is not None else g.nseq * g.Lkv
        core += 4.0 * g.nseq * g.L * g.Lkv * H     # upper bound for packed groups (q_len <= L)
    return 4.0 * H * H * q_rows + 4.0 * H * H * kv_rows + core


def cpu_baseline(B, Lt, seconds_budget=150.0):
    """The oracle (a plain-PyTorch fp32 port of the reference's step) timed on this box's host cores: 3 timed steps after 1 warm-up
    (BASELINE.md section 4), fewer only on a box so slow that they would not fit `seconds_budget` (~19 s per step on 64 cores)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import spmm_oracle as O
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    cfg = O.full_cfg()
    sd = O.init_state_dict(cfg, seed=0)
    sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20,
             'cooldown_epochs': 0}
    tr = O.OracleTrainer(sd, cfg, sched, {'lr': 5e-5, 'weight_decay': 0.02}, loader_len=1000)
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    t0 = time.time()
    tr.step(prop, ids, mask, 0, 100, train=True)              # warm-up
    warm = time.time() - t0
    n, t0 = 0, time.time()
    while n < 1 or (time.time() - t0 + warm < seconds_budget and n < 3):
        tr.step(prop, ids, mask, 0, 101 + n, train=True)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 3), "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": f"oracle/spmm_oracle.py OracleTrainer, full 12+6-layer H=768 model fp32, B={B}, Lt={Lt}, {n} timed step(s) after 1 warm-up"}


def child_bench(flags, keep, timeout=420, env=None):
    """One more measurement of this file in a child process (never an exec from a process that has touched the GPU); -> the named keys
    of its JSON line, or {"error": ...}.  `env`: variables added to the child's environment (the parent's SPMM_* ones are not inherited)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + flags, capture_output=True, text=True, timeout=timeout,
                           env=dict({k: v for k, v in os.environ.items() if not k.startswith("SPMM_")}, **(env or {})))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"rc={r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
        d = json.loads(line[-1])
        return {k: d[k] for k in keep if k in d}
    except Exception as e:            # noqa: BLE001  (a report, not a failure of the headline measurement)
        return {"error": repr(e)[:300]}


def decode_bench(args):
    """BASELINE.json configs[3]: PV -> SMILES k-beam decode (d_pv2smiles_batched.py:18-59) on synthetic PVs with the batched
    K/V-cache decoder (spmm_amd/decode.py), full-size random-init model.  A "step" is one chunk of molecules decoded to
    `--decode-steps` positions (with random weights [SEP] rarely wins, so nearly every molecule runs all positions: the worst case).
    Roofline: the `decode_attn` launches (single-query attention over the K/V cache, HBM gather) -- algorithmic bytes = the K and V
    rows every beam reads once + q + out, against their HIP-event time in one instrumented (non-graph) chunk."""
    torch.cuda.set_device(0)
    from spmm_amd import decode, ops
    from spmm_amd.config import BertConfig, SPMMConfig
    from spmm_amd.model import SPMM
    torch.manual_seed(0)
    cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True),
                     prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1), embed_dim=256, queue_size=36864)
    m = SPMM(spmm_config=cfg, no_train=True).eval()
    if args.sep_bias:
        # random-init weights never rank [SEP] among the k best successors, so every molecule runs all positions (the worst case, and the
        # default here).  A bias on the [SEP] logit gives the searches a spread of lengths and exercises the early exit.
        m.store.w("text_encoder.cls.predictions.bias")[3] += float(args.sep_bias)
    m.store.refresh_shadows()
    N, chunk, k, T = args.molecules, args.chunk, args.beams, args.decode_steps
    props = torch.randn(N, 53, generator=torch.Generator().manual_seed(42))
    chunks = [props[i:i + chunk] for i in range(0, N, chunk)]
    decode.beam_search_batched(m, props[:min(8, N)], k=k, max_steps=4)                      # warm-up (kernel attributes, allocator)
    for c in chunks[:args.warmup]:
        decode.beam_search_batched(m, c, k=k, max_steps=T, graph={'auto': None, 'on': True, 'off': False}[args.decode_graph])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nfin = 0
    for c in chunks:
        nfin += sum(len(r) for r in decode.beam_search_batched(m, c, k=k, max_steps=T, graph={'auto': None, 'on': True, 'off': False}[args.decode_graph]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "PV->SMILES k-beam decode molecules/sec", "value": round(N / dt, 2), "unit": "molecules/s", "n_gpus": 1, "steps": len(chunks),
           "warmup": min(args.warmup, len(chunks)), "ms_per_step": round(dt / len(chunks) * 1e3, 2), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"d_pv2smiles_batched.py: {N} synthetic PVs, k={k} beams, <= {T} positions, chunks of {chunk} molecules, 12-layer "
                                  "causal text encoder with cross-attention to the 54-token PV embeddings, K/V cache"
                                  + (", one hipGraph replay per position" if args.decode_graph == "on" else ""), "global_batch": chunk, "seq_len": T},
           "ms_per_position": round(dt / len(chunks) / (T + 1) * 1e3, 3), "finished_hypotheses": nfin, "sep_logit_bias": args.sep_bias,
           "last_chunk": dict(decode.last_run)}      # positions run, compactions of the batch (finished molecules dropped) and its final size
    # ---- instrumented chunk (eager, single stream): HIP events around every decode_attn / GEMM / LayerNorm launch
    ev, other = [], {"gemm": [], "layernorm": []}
    orig, orig_gemm, orig_ln = ops.decode_attn, ops.gemm_nt, ops.ln_fwd
    stream = torch.cuda.current_stream()

    def timed(q, K, V, o, *, nH, Lkv, seq_stride, tok_stride, anc=None, kv_div=1, group=1, **kw):
        R, H = q.shape[0], nH * 64
        if anc is not None:
            # self-attention: the DISTINCT cache rows the beams of a molecule reference at each position (a row shared by several
            # beams is loaded once by the molecule's wave) -- counted from the ancestry table, outside the timed interval
            a = anc.view(R // group, group, -1)[:, :, :Lkv].sort(dim=1).values
            kv_rows = int((a[:, 1:] != a[:, :-1]).sum()) + (R // group) * Lkv
        else:
            kv_rows = (R // kv_div) * Lkv                                   # cross-attention: one source per molecule
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r = orig(q, K, V, o, nH=nH, Lkv=Lkv, seq_stride=seq_stride, tok_stride=tok_stride, anc=anc, kv_div=kv_div, group=group, **kw)
        e1.record(stream)
        ev.append((e0, e1, 2.0 * (2 * kv_rows * H + 2 * R * H), 2.0 * (2 * (R * Lkv if anc is not None else kv_rows) * H + 2 * R * H)))
        return r

    def timed_other(kind, fn):
        def w(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            r = fn(*a, **k)
            e1.record(stream)
            other[kind].append((e0, e1))
            return r
        return w
    ops.decode_attn, ops.gemm_nt, ops.ln_fwd = timed, timed_other("gemm", orig_gemm), timed_other("layernorm", orig_ln)
    tot0, tot1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        tot0.record(stream)
        decode.beam_search_batched(m, chunks[0], k=k, max_steps=T, graph=False)
        tot1.record(stream)
    finally:
        ops.decode_attn, ops.gemm_nt, ops.ln_fwd = orig, orig_gemm, orig_ln
    torch.cuda.synchronize()
    tms = sum(a.elapsed_time(b) for a, b, *_ in ev)
    nbytes = sum(b for _, _, b, _ in ev)
    nbytes_rows = sum(b for *_, b in ev)
    tot_ms = tot0.elapsed_time(tot1)
    gms, lms = (sum(a.elapsed_time(b) for a, b in other[kk]) for kk in ("gemm", "layernorm"))
    out["roofline"] = {"bound": "hbm", "kernel": "decode_attn_mfma_kernel (csrc/decode.hip): one wave per (molecule, head) over the K/V cache, the beams as the N dimension of "
                                                         "16x16 MFMAs, keys / values streamed through an LDS-DMA ring; a row every beam shares is loaded once, a position where the "
                                                         "beams sit on different rows once per beam.  Bytes are counted as DISTINCT cache rows per molecule and position (from the "
                                                         "ancestry table) + q + out",
                       "achieved": round(nbytes / (tms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": round(nbytes / (tms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None, "launches": len(ev),
                       "avg_launch_us": round(tms * 1e3 / len(ev), 2), "algorithmic_bytes_per_launch": round(nbytes / len(ev)),
                       "bytes_per_launch_if_every_beam_row_read_its_own_prefix": round(nbytes_rows / len(ev)),
                       "measured": "HIP events around every decode_attn launch of one eager chunk (event-pair overhead ~2 us included)"}
    # HBM-side traffic of the same kernel from separate rocprofv3 --pmc passes (tools/pmc_decode.sh), quoted when taken on this workload
    pdir = os.path.join(ROOT, "profiles")
    ppath = next((os.path.join(pdir, f) for f in ("r06_pmc_decode_attn.json",) if os.path.exists(os.path.join(pdir, f))), "")
    if ppath:
        pmc = json.load(open(ppath))
        if pmc["workload"] == {"molecules": args.molecules, "beams": k, "positions": T, "chunk": args.chunk}:
            out["roofline"]["traffic"] = round(pmc["traffic_bytes_per_launch"])
            out["roofline"]["traffic_note"] = (f"bytes per launch, FETCH_SIZE x2 + WRITE_SIZE from profiles/{os.path.basename(ppath)} (L2<->fabric requests, "
                                               "Infinity-Cache hits included; average over self- and cross-attention launches of all chunks)")
    npos = T + 1
    out["position_breakdown_ms"] = {"note": "one eager single-stream chunk, HIP events around every launch of the three kernel families (event-pair overhead included); "
                                            "rest = the beam bookkeeping launch (spmm_beam_step), embedding, LM head tail, launch gaps and the event pairs themselves (the cache update is part of decode_attn since round 5)",
                                    "total": round(tot_ms / npos, 3), "gemm": round(gms / npos, 3), "gemm_launches": len(other["gemm"]) // npos,
                                    "decode_attn": round(tms / npos, 3), "layernorm": round(lms / npos, 3),
                                    "rest": round((tot_ms - gms - tms - lms) / npos, 3)}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import spmm_oracle as O
        import decode_oracle
        cores = min(os.cpu_count() or 1, 64)
        torch.set_num_threads(cores)
        ocfg = O.full_cfg()
        om = O.OracleModule(O.init_state_dict(ocfg, seed=0), ocfg)
        Tc = min(T, 24)
        t0 = time.perf_counter()
        decode_oracle.beam_search(om, props[0], k=k, max_steps=Tc)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(1.0 / dtc, 4), "unit": "molecules/s", "cores": cores, "kind": "port",
                               "sample": f"oracle/decode_oracle.py beam_search (the reference's one-molecule whole-prefix-per-step loop) on the fp32 "
                                         f"oracle model, 1 molecule, k={k}, {Tc} positions (the GPU figure runs {T}: the CPU cost grows ~quadratically "
                                         "with the positions, so this flatters the CPU)"}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--layers", type=str, default="12,6,6", help="text layers, fusion layer, PV layers")
    ap.add_argument("--queue", type=int, default=36864)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (NOT the benchmark configuration)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one hipGraph (dense text layout, one rank; implies --no-kernel-timing)")
    ap.add_argument("--check-replicas", action="store_true", help="after the run assert parameters / queues are identical on all ranks")
    ap.add_argument("--extra-streams", type=int, default=0, help="diagnostic: create N more HIP streams and run one tiny kernel on each before the "
                    "run (what the number of ACTIVE hardware queues costs; profiles/r03_hw_queues.txt)")
    ap.add_argument("--decode", action="store_true", help="BASELINE configs[3]: PV->SMILES k-beam decode throughput instead of the pretrain step")
    ap.add_argument("--molecules", type=int, default=1000)
    ap.add_argument("--chunk", type=int, default=1000, help="--decode: molecules decoded together (1000 x 5 beams = 5000 rows per GEMM: 2 010-2 090 "
                    "molecules/s against 943 at 250 and 1 469 at 500, profiles/r03_decode_bench.json)")
    ap.add_argument("--beams", type=int, default=5)
    ap.add_argument("--decode-steps", type=int, default=100)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short configs[3] (decode) and configs[4]-shape (B=512, Lt=256) "
                    "measurements the default one-GPU run appends to its JSON line (each in a child process, after the timed region)")
    ap.add_argument("--sep-bias", type=float, default=0.0, help="--decode: added to the [SEP] logit of the random-init LM head (0 = [SEP] never wins: every "
                    "molecule decodes all positions)")
    ap.add_argument("--decode-graph", choices=["auto", "on", "off"], default="auto", help="--decode: one hipGraph replay per position instead of eager launches (auto = off since round 5: replay is "
                    "opt-in, it pays only below decode.GRAPH_BELOW_ROWS beam rows per chunk)")
    args = ap.parse_args()
    # the other configs ride only on the default workload of one GPU (what the driver runs), not on every experiment
    default_run = (args.batch == 128 and args.seq_len == 128 and args.layers == "12,6,6" and args.queue == 36864 and not args.eval_mode and not args.graph
                   and args.gpus == 1 and not args.no_kernel_timing)
    if args.decode:
        if args.warmup == 10:
            args.warmup = 1
        return decode_bench(args)
    if args.graph:
        args.no_kernel_timing = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start one as a CHILD process (never exec from a process that may touch the GPU) and pass its
        # exit code on.  Nothing in this process has initialised HIP yet.
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("SPMM_DIST_BACKEND", "nccl")          # "gloo" lets two ranks share one GPU in tests
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(ndev, 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            torch.distributed.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
        if os.environ.get("SPMM_FORCE_DIST") == "1":        # (EngineOptions.force_dist; the process group must exist before the model)
            # single-rank RCCL group: drives the real collective code path (overlapped per-layer all-reduce on RCCL's stream,
            # feature all-gather) on a one-GPU box; used by tests/test_step_gpu.py
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    dev = torch.device(f"cuda:{torch.cuda.current_device()}")
    extra = [torch.cuda.Stream() for _ in range(args.extra_streams)]
    for st_ in extra:
        with torch.cuda.stream(st_):
            torch.zeros(8, device=dev).add_(1.0)
    torch.cuda.synchronize()

    from spmm_amd import ops
    from spmm_amd.config import BertConfig, SPMMConfig
    from spmm_amd.model import SPMM
    from spmm_amd.options import EngineOptions
    from spmm_amd.parallel import grad_sync_fn, broadcast_state_

    nt, f, npv = (int(x) for x in args.layers.split(","))
    cfg = SPMMConfig(text=BertConfig(num_hidden_layers=nt, fusion_layer=f, add_cross_attention=True),
                     prop=BertConfig(num_hidden_layers=npv, fusion_layer=f, vocab_size=1), embed_dim=256, queue_size=args.queue)
    sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20,
             'cooldown_epochs': 0}
    tc = {'embed_dim': 256, 'temp': 0.07, 'mlm_probability': 0.15, 'queue_size': args.queue, 'momentum': 0.995, 'alpha': 0.4,
          'schedular': sched, 'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
    torch.manual_seed(42)                                   # SPMM_pretrain.py:48 default seed; same init on every rank
    opts = EngineOptions.from_env()
    model = SPMM(config=tc, spmm_config=cfg, loader_len=1000, options=opts)
    broadcast_state_([model.store.flat, model.store.flat_m] + [model.store.buffers[k] for k in ("prop_queue", "text_queue")])
    model.store.refresh_shadows()
    model.engine.invalidate_banks()
    model.train(not args.eval_mode)
    B, Lt = args.batch, args.seq_len
    batches = [synthetic_batch(B, Lt, 42 + 1000 * rank + i, dev) for i in range(4)]
    sync = grad_sync_fn(model.store, opts)
    n_valid = sum(b[3] for b in batches) / len(batches)                # real (non-padding) text tokens per batch

    def one_step(i):
        prop, ids, mask, ntok = batches[i % len(batches)]
        if args.graph:                   # the step as one hipGraph replay (dense text layout, single rank): NOT the headline configuration
            return model.fused_step_graphed(prop, ids, mask, 0.4)
        return model.fused_step(prop, ids, mask, 0.4, grad_sync=sync, n_tokens=ntok)   # the data pipeline knows the token count (host mask sum)

    replica_report = {}

    def check_replicas(after, fatal):
        """Every rank must hold the same parameters, momentum parameters, queues and pointer (there is no per-step buffer broadcast).
        Collective on every rank; with fatal=False a divergence is REPORTED in the JSON line instead of ending the scaling run."""
        from spmm_amd.parallel import assert_replicas_identical
        bad = []
        for t, what in ((model.store.flat, "student parameters"), (model.store.flat_m, "momentum parameters"),
                        (model.store.buffers["prop_queue"], "prop_queue"), (model.store.buffers["text_queue"], "text_queue"),
                        (model.store.buffers["queue_ptr"], "queue_ptr")):
            try:
                assert_replicas_identical(t, what)
            except RuntimeError:
                if fatal:
                    raise
                bad.append(what)
        flag = torch.tensor([len(bad)], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        replica_report[f"after_{after}_steps"] = "identical" if int(flag) == 0 else f"DIVERGED on some rank (this rank: {bad or 'none'})"
        if rank == 0 and int(flag) == 0:
            print("replicas identical after", after, "steps; queue_ptr =", int(model.queue_ptr), flush=True)

    rccl_ranks = None
    if world > 1:
        # what every rank's communicator believes the job is: [world size seen] per rank, gathered through the collective itself
        seen = torch.tensor([torch.distributed.get_world_size()], device=dev, dtype=torch.int32)
        allseen = torch.empty(world, dtype=torch.int32, device=dev)
        torch.distributed.all_gather_into_tensor(allseen, seen)
        rccl_ranks = {"backend": torch.distributed.get_backend(), "world_size_seen_by_rank": allseen.cpu().tolist()}
    n_check = 0
    if sync is not None and opts.schedule_check and opts.multi_stream:
        # the model's own schedule check (spmm_amd/model.py: both stream schedules tried over its first 12 steps) runs BEFORE the warm-up:
        # extra untimed steps, reported as `schedule_check_steps`
        n_check = model.SCHEDULE_CHECK_STEPS
        for i in range(n_check):
            one_step(i)
    for i in range(args.warmup):
        losses = one_step(i)
        if world > 1 and i == 1:         # replicas must agree from the start (no per-step buffer broadcast): checked after 2 steps, untimed
            check_replicas(2, fatal=args.check_replicas)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step spread, no extra syncs
    host_lead = None
    if os.environ.get("SPMM_BENCH_HOST_LEAD") == "1":          # diagnostic (not the default run): how far ahead of the GPU the host enqueues
        host_lead, opt_, orig_step_ = [], model.optimizers(), model.optimizers().step

        def step_probe(*a, **k):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            reached = e0.query()                                # True: the GPU had already drained everything enqueued before the optimiser
            r = orig_step_(*a, **k)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            host_lead.append((reached, e0, e1))
            return r
        opt_.step = step_probe
    if hasattr(sync, "wait_events"):
        sync.wait_events = []             # event pair around the compute stream's wait in OverlappedGradSync.finish(), one per timed step
    smp = smi_sampler(in_process_only=True) if rank == 0 else None       # clock / power of this rank's GPU over the timed region (a 20-Hz in-process reader thread)
    if smp is not None:
        smp.__enter__()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        losses = one_step(args.warmup + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    power = None
    if smp is not None:
        smp.__exit__(None, None, None)
        power = smp.summary()
    comm_wait = None
    if hasattr(sync, "wait_events"):
        comm_wait = [a.elapsed_time(b) for a, b in (sync.wait_events or [])]
        sync.wait_events = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    if host_lead is not None:
        opt_.step = orig_step_
        el = sorted(a.elapsed_time(b) for _, a, b in host_lead)
        print(f"[host-lead] optimiser enqueued after the GPU had drained the backward in {sum(r for r, _, _ in host_lead)} of {len(host_lead)} steps; "
              f"backward end -> optimiser end on the GPU: median {el[len(el) // 2]:.3f} ms, min {el[0]:.3f}, max {el[-1]:.3f}", file=sys.stderr, flush=True)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    spread = {"median": round(per_step[len(per_step) // 2], 3), "p10": round(per_step[int(0.1 * (len(per_step) - 1))], 3),
              "p90": round(per_step[int(0.9 * (len(per_step) - 1) + 0.5)], 3), "note": "GPU-side interval between consecutive steps' last kernels (this rank)"}
    dp_diag = None
    if world > 1 or comm_wait is not None:
        # Per rank, gathered: median / p90 step, the time the compute stream waited for the exchange at the end of the backward
        # (communication the backward did not hide), so that a scaling run explains itself: which rank is slow, and whether it waits.
        mine = torch.tensor([spread["median"], spread["p90"], (sum(comm_wait) / len(comm_wait)) if comm_wait else -1.0,
                             max(comm_wait) if comm_wait else -1.0], dtype=torch.float32, device=dev)
        allr = mine[None, :]
        if world > 1:
            allr = torch.empty(world, 4, dtype=torch.float32, device=dev)
            torch.distributed.all_gather_into_tensor(allr, mine[None, :].contiguous())
        allr = allr.cpu().tolist()
        slow = max(range(len(allr)), key=lambda r_: allr[r_][0])
        dp_diag = {"per_rank": [{"rank": r_, "step_ms_median": round(v[0], 3), "step_ms_p90": round(v[1], 3),
                                 "comm_exposed_ms": None if v[2] < 0 else round(v[2], 3), "comm_exposed_ms_max": None if v[3] < 0 else round(v[3], 3)}
                                for r_, v in enumerate(allr)],
                   "slowest_rank": slow, "slowest_rank_step_ms": {"median": round(allr[slow][0], 3), "p90": round(allr[slow][1], 3)},
                   "note": "comm_exposed_ms = GPU time the compute stream spends in OverlappedGradSync.finish() waiting for the gradient exchange "
                           "(mean over the timed steps; includes the sweep of the ~2 % of the arena no layer covers)"}
    final_losses = [float(x) for x in losses.cpu()]
    mst = torch.cuda.memory_stats(dev)
    hbm = {"peak_allocated_gb": round(mst.get("allocated_bytes.all.peak", 0) / 2**30, 2), "peak_reserved_gb": round(mst.get("reserved_bytes.all.peak", 0) / 2**30, 2),
           "alloc_retries": int(mst.get("num_alloc_retries", 0)), "device_mallocs": int(mst.get("segment.all.allocated", 0))}

    # ---- per-kernel timing with HIP events on the launch stream: same step, every GEMM / cross-attention launch bracketed.
    roof, xattn = None, None
    if not args.no_kernel_timing:      # every rank runs the instrumented steps (they contain the collectives); rank 0 reports
        # (a failure in this report-only section must not cost the headline line: it is caught and reported on one rank; with several ranks
        #  the section contains collectives and an exception is fatal anyway)
        _saved_ops = (ops.gemm_nt, ops.attn_fwd, model.engine._attn_block_fwd, model.engine.opt, model.engine.multi_stream, model.engine.wgrad_async)
        try:
            ev = {"gemm": [], "xattn": []}
            orig_gemm, orig_attn = ops.gemm_nt, ops.attn_fwd
            stream = torch.cuda.current_stream()

            def timed(kind, fn, flops):
                def w(*a, **k):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    r = fn(*a, **k)
                    e1.record(stream)
                    ev[kind].append((e0, e1, flops(*a, **k)))
                    return r
                return w

            gemm_bytes = [0.0]
            shape_log = []

            def gemm_flops(A, W, C, **k):
                M, N, K = A.shape[0], W.shape[0], (k.get("K") or A.shape[1])
                # algorithmic bytes of the launch: each operand read once, each output written once
                b = 2.0 * (M * K + N * K) + M * N * C.element_size()
                for name in ("R", "G", "C2"):
                    if k.get(name) is not None:
                        b += M * N * k[name].element_size()
                gemm_bytes[0] += b
                shape_log.append((M, N, K, int(k.get("epi", 0)), C.dtype == torch.float32))
                return 2.0 * M * N * K

            model.engine.multi_stream = False     # per-launch events need one stream; concurrency would also smear the durations
            model.engine.wgrad_async = False      # (the weight-gradient side stream too)
            # What an event pair adds to the interval it brackets (command-processor time between the first event's timestamp and the
            # kernel's start, and between its end and the second timestamp): intervals around ONE and around TWO minimal kernels through
            # the same launch path, overhead = 2 I1 - I2 (the kernel's own cost cancels).  rocprofv3's kernel trace has no such term;
            # subtracting it is what makes `avg_launch_us` comparable with profiles/*kernel_stats*.
            cal = torch.zeros(1, device=dev)

            def interval(n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(n):
                    ops.clamp_scalar(cal, 0.0, 1.0)
                e1.record(stream)
                return e0, e1
            pairs = [(interval(1), interval(2)) for _ in range(200)]
            torch.cuda.synchronize()
            i1 = sorted(a.elapsed_time(b) for (a, b), _ in pairs)[100]
            i2 = sorted(a.elapsed_time(b) for _, (a, b) in pairs)[100]
            ev_overhead_ms = max(0.0, 2 * i1 - i2)
            ops.gemm_nt = timed("gemm", orig_gemm, gemm_flops)
            nsteps = min(3, args.steps)
            for i in range(nsteps):
                one_step(i)
            torch.cuda.synchronize()
            ops.gemm_nt = orig_gemm
            # sustained shader clock of the single-stream schedule these intervals come from: the same steps again, un-instrumented, under
            # the sampler (the events' host work would otherwise thin the load the clock responds to)
            smp1 = smi_sampler() if rank == 0 else None
            if smp1 is not None:
                with smp1:
                    for i in range(max(10, nsteps)):
                        one_step(i)
                    torch.cuda.synchronize()
                clk1 = smp1.summary()
            else:
                clk1 = None
                for i in range(max(10, nsteps)):
                    one_step(i)
                torch.cuda.synchronize()
            # the metric's kernel: the cross-attention unit (Q/K/V projections + softmax(QK^T)V + output projection, forward),
            # timed as a whole with events around BertAttention(cross) in separate instrumented steps
            eng = model.engine
            orig_blk = eng._attn_block_fwd

            def blk(pfx, c, X, groups, save, cross, X32=None):
                if not cross:
                    return orig_blk(pfx, c, X, groups, save, cross, X32=X32)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                r = orig_blk(pfx, c, X, groups, save, cross, X32=X32)
                e1.record(stream)
                fl = sum(cross_attn_unit_flops(g.nseq, g.L, g.Lkv) for g in groups)
                Hh = X.shape[1]
                core_b = 2.0 * (2 * X.shape[0] * Hh + sum(g.nseq * g.Lkv * 2 * Hh for g in groups))      # Q in, context out, K / V per query sequence
                ev["xattn"].append((e0, e1, (fl, cross_attn_executed_flops(groups, X.shape[0]), core_b, 2.0 * 4 * X.shape[0] * Hh, bool(save))))
                return r

            eng._attn_block_fwd = blk
            for i in range(nsteps):
                one_step(i)
            torch.cuda.synchronize()
            # ... and the other form beside the default one.  Default since round 6 (EngineOptions.fused_xattn = "nograd"): the passes that keep
            # no tape (the momentum fusion pass) run the block as ONE launch per query group (csrc/xattn.hip: core + output projection + dropout +
            # residual + LayerNorm), the taped student passes as the composite of launches; the other form here = the composite everywhere
            # ("off"; or "all" = the row-panel kernel everywhere when the run's default is "off")
            n_comp = len(ev["xattn"])
            other_mode = "off" if opts.fused_xattn != "off" else "all"
            eng.opt = opts.replace(fused_xattn=other_mode)
            one_step(0)                                          # untimed: the fused form's one-time self-check (a host sync) happens here
            torch.cuda.synchronize()
            n_comp, n_skip = len(ev["xattn"]), len(ev["xattn"]) - n_comp
            for i in range(nsteps):
                one_step(i)
            torch.cuda.synchronize()
            eng.opt = opts
            ev_other, ev["xattn"] = ev["xattn"][n_comp:], ev["xattn"][:n_comp - n_skip]
            eng._attn_block_fwd = orig_blk
            model.engine.multi_stream = opts.multi_stream
            model.engine.wgrad_async = opts.multi_stream and opts.wgrad_stream
            by_shape_early = {}
            for (a_, b_, _), (M_, N_, K_, epi_, f32_) in zip(ev["gemm"], shape_log):
                key_ = (round(M_ / 1024) * 1024 if M_ >= 2048 else M_, N_, K_, epi_, f32_)
                t_, f_ = by_shape_early.get(key_, (0.0, 0.0))
                by_shape_early[key_] = (t_ + a_.elapsed_time(b_) - ev_overhead_ms, f_ + 2.0 * M_ * N_ * K_)
            x_ms = sum(a.elapsed_time(b) for a, b, _ in ev["xattn"])
            x_alg = sum(fl[0] for _, _, fl in ev["xattn"])
            x_exe = sum(fl[1] for _, _, fl in ev["xattn"])
            xattn = {"unit": "cross-attention block forward (Q, K, V projections + softmax(QK^T/8 + mask)V + output projection + residual LN), "
                             "the 12 calls of a step (S6 student batch and S5 momentum batch)",
                     "algorithmic_tflops": round(x_alg / (x_ms * 1e-3) / 1e12, 1),
                     "executed_tflops": round(x_exe / (x_ms * 1e-3) / 1e12, 1),
                     "executed_frac_of_bf16_peak": round(x_exe / (x_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                     "calls_per_step": len(ev["xattn"]) // nsteps, "ms_per_step": round(x_ms / nsteps, 3),
                     "form": {"off": "composite: Q GEMM + K/V GEMM + attn_fwd per group + output GEMM + ln_fwd",
                              "all": "fused row-panel kernel (csrc/xattn.hip) + Q and K/V projection GEMMs",
                              "nograd": "no-grad passes (momentum fusion pass, 6 of the 12 calls): fused row-panel kernel (csrc/xattn.hip) + Q and K/V projection "
                                        "GEMMs; taped student passes: composite of Q GEMM + K/V GEMM + attn_fwd per group + output GEMM + ln_fwd"}[opts.fused_xattn],
                     "ms_per_step_taped_calls": round(sum(a.elapsed_time(b) for a, b, fl in ev["xattn"] if fl[4]) / nsteps, 3),
                     "ms_per_step_nograd_calls": round(sum(a.elapsed_time(b) for a, b, fl in ev["xattn"] if not fl[4]) / nsteps, 3),
                     "other_form": other_mode,
                     "other_form_ms_per_step": round(sum(a.elapsed_time(b) for a, b, _ in ev_other) / nsteps, 3),
                     "other_form_ms_per_step_taped_calls": round(sum(a.elapsed_time(b) for a, b, fl in ev_other if fl[4]) / nsteps, 3),
                     "other_form_ms_per_step_nograd_calls": round(sum(a.elapsed_time(b) for a, b, fl in ev_other if not fl[4]) / nsteps, 3),
                     "other_form_executed_frac_of_bf16_peak": round(sum(fl[1] for _, _, fl in ev_other) / (sum(a.elapsed_time(b) for a, b, _ in ev_other) * 1e-3)
                                                                    / 1e12 / PEAK_BF16_TFLOPS, 4),
                     "note": "algorithmic = the reference's work, nseq*(4H^2 Lq + 4H^2 Lkv + 4 Lq Lkv H) per query sequence; executed = what runs here "
                             "(K/V projected once per unique key/value source, packed rows); padded-tile waste excluded from both"}
            # What bounds the unit: it is 85 % projection FLOPs, which run at the rate the step's K = 768 GEMMs reach, plus a core and a residual
            # LayerNorm that are HBM passes.  Ceiling = executed FLOPs / (projection FLOPs / that GEMM rate + core bytes / HBM + LayerNorm bytes / HBM),
            # with the GEMM rate MEASURED in this run (the N = 768, K = 768 launches of `roofline.shapes`) and HBM at the 6.3 TB/s a streaming
            # kernel reaches on this part (MI355X_MICROARCH.md) -- i.e. core and LayerNorm AT their roofs; north_star's 0.50 would need the
            # projections alone to run above 0.55 of the sheet peak.
            x_core_b = sum(fl[2] for _, _, fl in ev["xattn"])
            x_ln_b = sum(fl[3] for _, _, fl in ev["xattn"])
            k768 = [(v[0], v[1]) for k_, v in by_shape_early.items() if k_[1] == 768 and k_[2] == 768 and not k_[4]] if by_shape_early else []
            if k768:
                g_rate = sum(f_ for _, f_ in k768) / (sum(t_ for t_, _ in k768) * 1e-3)           # FLOP/s of the K = 768, N = 768 launches
                t_floor = x_exe * 0.85 / g_rate + (x_core_b + x_ln_b) / 6.3e12
                xattn["ceiling"] = {"formula": "executed FLOPs / (0.85 x executed FLOPs / R_gemm + (core bytes + LayerNorm bytes) / 6.3 TB/s)",
                                    "R_gemm_tflops_measured_N768_K768": round(g_rate / 1e12, 1), "core_bytes": x_core_b / nsteps, "layernorm_bytes": x_ln_b / nsteps,
                                    "ceiling_ms_per_step": round(t_floor * 1e3 / nsteps, 3),
                                    "ceiling_frac_of_bf16_peak": round(x_exe / t_floor / 1e12 / PEAK_BF16_TFLOPS, 4),
                                    "measured_over_ceiling": round(t_floor * 1e3 / x_ms, 3),
                                    "note": "the unit cannot beat the GEMMs it is made of: with its core and LayerNorm at the HBM roof it would reach this fraction; "
                                            "the >= 0.50 of north_star needs projections above 0.55 of the sheet peak, which no K = 768 GEMM reaches on this part "
                                            "(profiles/r05_power.txt: 0.49-0.53 of the peak AT THE SUSTAINED CLOCK)"}
            raw_ms = sum(a.elapsed_time(b) for a, b, _ in ev["gemm"])
            tot_fl = sum(fl for _, _, fl in ev["gemm"])
            n_launch = len(ev["gemm"])
            tot_ms = raw_ms - n_launch * ev_overhead_ms          # kernel time: the event pairs' own share removed (calibrated above)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "NT GEMM family, all launches of the step: gemm_nt_p8_kernel (persistent 256x256 8-phase, bf16 MFMA "
                                               "16x16x32) for the large shapes, gemm_nt_v2/v1 for fp32 outputs and small problems", "achieved": round(ach, 1),
                    "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                    "launches_per_step": n_launch // nsteps, "avg_launch_us": round(tot_ms * 1e3 / n_launch, 2),
                    "flops_per_step": tot_fl / nsteps, "gemm_ms_per_step": round(tot_ms / nsteps, 3),
                    "algorithmic_bytes_per_launch": round(gemm_bytes[0] / n_launch),
                    "avg_launch_us_raw": round(raw_ms * 1e3 / n_launch, 2), "event_pair_overhead_us": round(ev_overhead_ms * 1e3, 2),
                    "measured": f"HIP events around every launch, {nsteps} instrumented single-stream steps after the timed region; each interval "
                                "minus the calibrated event-pair overhead (2 I1 - I2 around one / two minimal kernels), which the rocprofv3 kernel trace does not contain"}
            # `frac` divides by the 2.4-GHz sheet peak; the part runs this step well below that clock (power-limited), so the same rate is
            # also quoted against the MFMA peak AT THE SUSTAINED CLOCK: peak x clock / 2 400 MHz.  Clock = mean sclk of GPU 0 over ten
            # single-stream steps (the schedule the per-launch intervals come from); the timed region's own clock / power is in `power`.
            if clk1 is not None and clk1.get("clock_mhz"):
                roof["clock_mhz"] = clk1["clock_mhz"]
                roof["power_w"] = clk1.get("power_w")
                roof["peak_at_clock"] = round(PEAK_BF16_TFLOPS * clk1["clock_mhz"] / 2400.0, 1)
                roof["frac_at_clock"] = round(ach / (PEAK_BF16_TFLOPS * clk1["clock_mhz"] / 2400.0), 4)
                roof["clock_note"] = (f"mean of {clk1['samples']} samples ({clk1['source']}) over 10 un-instrumented single-stream steps; "
                                      "per kernel family: profiles/r06_power.txt")
            else:
                roof["clock_mhz"] = None
                roof["clock_note"] = f"no clock source on this box: {clk1}"
            # per-shape table of the same launches (where the family's time goes inside the step): M bucketed to 1 k rows
            by_shape = {}
            for (a, b, _), (M_, N_, K_, epi_, f32_) in zip(ev["gemm"], shape_log):
                key = (round(M_ / 1024) * 1024 if M_ >= 2048 else M_, N_, K_, epi_, f32_)
                t_, f_, n_ = by_shape.get(key, (0.0, 0.0, 0))
                by_shape[key] = (t_ + a.elapsed_time(b) - ev_overhead_ms, f_ + 2.0 * M_ * N_ * K_, n_ + 1)
            roof["shapes"] = [{"M~": k[0], "N": k[1], "K": k[2], "epi": k[3], "f32_out": k[4], "launches_per_step": v[2] // nsteps,
                               "ms_per_step": round(v[0] / nsteps, 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else None}
                              for k, v in sorted(by_shape.items(), key=lambda kv: -kv[1][0])[:16]]
            # HBM-side traffic of the same launches comes from separate rocprofv3 --pmc passes (they cannot run inside this
            # process); the committed summary is quoted only when it was taken on this exact workload.
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            pmc_path = next((os.path.join(pdir, f) for f in ("r06_pmc_nt_gemm.json", "r05_pmc_nt_gemm.json", "r04_pmc_nt_gemm.json", "r03_pmc_nt_gemm.json", "r02_pmc_nt_gemm.json") if os.path.exists(os.path.join(pdir, f))), "")
            if pmc_path:
                pmc = json.load(open(pmc_path))
                if pmc["workload"] == {"batch": B, "seq_len": Lt, "layers": nt, "queue": args.queue}:
                    roof["traffic"] = round(pmc["traffic_bytes_per_launch"])
                    roof["traffic_note"] = (f"bytes per launch, FETCH_SIZE x2 + WRITE_SIZE from profiles/{os.path.basename(pmc_path)} "
                                            "(L2<->fabric requests, Infinity-Cache hits included)")
        except Exception as e:          # noqa: BLE001
            if world > 1:
                raise
            import traceback
            ops.gemm_nt, ops.attn_fwd, model.engine._attn_block_fwd, model.engine.opt, model.engine.multi_stream, model.engine.wgrad_async = _saved_ops
            roof = {"error": "per-launch instrumentation failed: " + repr(e)[:300], "traceback": traceback.format_exc()[-1500:]}
            xattn = None

    flops = step_flops(B, Lt, n_text=nt, fusion=f, n_pv=npv, Q=args.queue)
    value = world * B / (dt / args.steps)
    out = {"metric": "pretrain molecules/sec", "value": round(value, 2), "unit": "molecules/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic", "step_ms": spread,
           "config": {"workload": f"SPMM pretrain step, text {nt} layers (fusion at {f}) + PV {npv} layers, H=768, 12 heads, queue {args.queue}, "
                                  f"train mode (dropout 0.1), fwd+bwd+clip+AdamW+EMA", "global_batch": world * B, "seq_len": Lt,
                      "parallelism": f"dp{world}", "schedule": "one hipGraph replay per step, dense text layout" if args.graph else
                      (("eager launches on one HIP stream" if getattr(model.engine, "_one_stream", False) else "eager launches on three HIP streams") + ", packed text rows" + ("" if sync is None else
                       "; per-layer gradient exchange overlapped with the backward, weight gradients on the backward's stream meanwhile"))},
           "step_tflop": round(flops / 1e12, 2),
           "executed_step_tflop": round((flops - shared_kv_saving(B, Lt, n_text=nt, fusion=f)
                                         - (padding_saving(B, Lt, n_valid, n_text=nt, fusion=f) if Lt <= ops.ATTN_MAXL else 0.0)
                                         - ((cls_top_saving(B, Lt, n_valid) + neg_pack_saving(B, Lt, n_valid, n_text=nt, fusion=f))
                                            if (Lt <= ops.ATTN_MAXL and opts.cls_only_top and opts.pack_text and not opts.resid_fp32) else 0.0)) / 1e12, 2),
           "valid_text_tokens_frac": round(n_valid / (B * Lt), 4),
           "model_tflops_per_gpu": round(flops / (dt / args.steps) / 1e12, 1),
           "mfma_frac_of_peak_step": round(flops / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4), "losses": final_losses, "hbm": hbm}
    if power is not None:
        out["power"] = dict(power, note="GPU 0 over the timed region (three-stream schedule): mean shader clock and socket power")
        if power.get("clock_mhz"):
            out["mfma_frac_of_peak_step_at_clock"] = round(flops / (dt / args.steps) / 1e12 / (PEAK_BF16_TFLOPS * power["clock_mhz"] / 2400.0), 4)
    if dp_diag is not None:
        out["data_parallel"] = dp_diag
    if n_check:
        out["schedule_check_steps"] = n_check
        out["schedule_check"] = model.schedule_decision()
    from spmm_amd import streams
    if rccl_ranks is not None:
        out["rccl_ranks"] = rccl_ranks
        out["replicas"] = replica_report
    if streams.log():
        out["stream_placement"] = streams.log()
    if rank == 0:
        if roof is not None:
            out["roofline"] = roof
        if xattn is not None:
            out["cross_attention"] = xattn
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(16, Lt)
            except Exception as e:      # noqa: BLE001  (a report beside the headline, never instead of it)
                out["cpu_baseline"] = {"error": repr(e)[:300]}
        if world == 1 and default_run and not args.no_other_configs:
            # BASELINE configs[3] and configs[4]'s per-GPU shape, driver-visible: short runs of this same file in child processes
            # (this process's model and cached blocks are released first; a child that fails is reported, never fatal)
            model.engine.tape = None
            del model, one_step
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["other_configs"] = {
                "configs[3] PV->SMILES k-beam decode (1000 PVs, k=5, 100 positions)": child_bench(["--decode", "--no-cpu-baseline"],
                    ("metric", "value", "unit", "ms_per_position", "finished_hypotheses", "last_chunk", "roofline", "position_breakdown_ms")),
                "configs[3] with a [SEP] logit bias of 0.5 (molecules finish at different positions: early exit, finished molecules leave the batch)":
                    child_bench(["--decode", "--no-cpu-baseline", "--sep-bias", "0.5"], ("value", "unit", "ms_per_step", "finished_hypotheses", "last_chunk")),
                "configs[4] per-GPU shape in bf16 (B=512, Lt=256)": child_bench(["--batch", "512", "--seq-len", "256", "--steps", "5", "--warmup", "4",
                    "--no-cpu-baseline", "--no-kernel-timing", "--no-other-configs"],
                    ("metric", "value", "unit", "ms_per_step", "step_ms", "model_tflops_per_gpu", "mfma_frac_of_peak_step", "hbm", "config")),
                "configs[2] code path on ONE GPU: the default workload through the N>1 step (one-rank RCCL group, SPMM_FORCE_DIST=1: per-layer gradient "
                "exchanges on RCCL's stream during the backward, the fused feature all-gather) -- ms_per_step against this line's own is what the "
                "data-parallel schedule costs before any link time": child_bench(["--steps", "20", "--warmup", "6", "--no-cpu-baseline", "--no-kernel-timing",
                    "--no-other-configs"], ("value", "unit", "ms_per_step", "step_ms", "schedule_check", "stream_placement"),
                    env={"SPMM_FORCE_DIST": "1", "MASTER_PORT": str(29533 + os.getpid() % 400)})}
        print(json.dumps(out), flush=True)
    if args.check_replicas and world > 1:
        check_replicas(args.warmup + args.steps, fatal=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
