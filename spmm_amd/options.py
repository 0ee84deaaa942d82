"""`EngineOptions`: every switch that steers the product path, in one place.

A field is set (in this order of precedence) by the `options=` argument of `spmm_amd.SPMM`, by its environment variable (read
ONCE, when `EngineOptions.from_env()` runs in the constructor), or by its default.  The defaults are the benchmark
configuration; nothing else in the package reads `os.environ` for behaviour (the debugging aids listed at the
bottom aside)."""
from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass

# field -> (environment variable, parser)
_ENV = {
    "pack_text": ("SPMM_PACK_TEXT", lambda s: s != "0"),
    "cls_only_top": ("SPMM_CLS_ONLY_TOP", lambda s: s != "0"),
    "multi_stream": ("SPMM_STREAMS", lambda s: s != "1"),
    "wgrad_stream": ("SPMM_WGRAD_STREAM", lambda s: s != "0"),
    "pv_wgrad_inline": ("SPMM_PV_WGRAD_INLINE", int),
    "ln_from_y": ("SPMM_LN_FROM_Y", lambda s: s != "0"),
    "resid_fp32": ("SPMM_RESID_FP32", lambda s: s == "1"),
    "fused_xattn": ("SPMM_FUSED_XATTN", lambda s: {"0": "off", "1": "all"}.get(s, s)),
    "grad_overlap": ("SPMM_GRAD_OVERLAP", lambda s: s != "0"),
    "grad_wire": ("SPMM_GRAD_WIRE", str),
    "nt_under_comm": ("SPMM_NT_UNDER_COMM", str),
    "force_dist": ("SPMM_FORCE_DIST", lambda s: s == "1"),
    "dp_four_streams": ("SPMM_DP_FOUR_STREAMS", lambda s: s != "0"),
    "probe_streams": ("SPMM_PROBE_STREAMS", lambda s: s != "0"),
    "schedule_check": ("SPMM_SCHEDULE_CHECK", lambda s: s != "0"),
}


@dataclass
class EngineOptions:
    # --- schedule of one step (spmm_amd/step.py, engine.py) ---
    pack_text: bool = True        # drop padding-token rows from the text passes whose losses read only position 0
    cls_only_top: bool = True     # last fusion layer of the ITM passes on position 0 only (what the ITM head reads, SPMM_models.py:199-201); their other
    #                               rows stay keys / values of its self-attention.  Packed path only; exact
    multi_stream: bool = True     # independent encoder chains on three HIP streams; False = everything on the caller's stream
    wgrad_stream: bool = True     # weight-gradient GEMMs on a stream of their own (single rank; rests while gradients are exchanged)
    pv_wgrad_inline: int = 4      # ... except those of the PV encoder's first `n` layers (the LAST its backward reaches), which stay on that chain's
    #                               own stream: it ends ~2 ms before the text encoder's on the side stream, and the one weight-gradient stream,
    #                               fed by both chains, is what the optimiser then waits for (tools/phase_times.py; EXPERIMENTS.md 3.10; round 6 re-swept 2 / 3 / 4 / 5 / 6
    #                               on the final tree: 4 is best by 0.15-0.2 ms in five of five pairs)
    ln_from_y: bool = True        # the residual LayerNorms keep no pre-norm sum for the backward: spmm_ln_bwd recovers the normalised values from the
    #                               OUTPUT, (y - beta) / gamma -- one 131-MB write per LayerNorm less at the benchmark shape (-0.5 ms per step), same
    #                               rounding budget (one bf16 tensor read either way; EXPERIMENTS.md 4.8).  False = the stored sum (rounds 1-5)
    fused_xattn: str = "nograd"   # cross-attention block forward as ONE row-panel launch (csrc/xattn.hip: core + output projection + dropout + residual +
    #                               LayerNorm): "nograd" = the passes that keep no tape (momentum fusion pass, inference facades) -- 52.88 vs 53.01 ms
    #                               per step in three alternating pairs, profiles/r06_xattn_modes.txt; "all" = the taped passes too (53.00: no gain
    #                               there, the tape's extra outputs cost what the fusion saves); "off" = the composite of launches everywhere
    # --- precision (NOT the headline configuration) ---
    resid_fp32: bool = False      # fp32 residual stream through the LayerNorms and fp32 inputs to the loss heads
    # --- data parallelism (spmm_amd/parallel.py) ---
    grad_overlap: bool = True     # per-layer gradient exchange issued during the backward; False = one bucketed all-reduce after it
    grad_wire: str = "fp32"       # "fp32": all-reduce on the arena; "bf16": cast + reduce-scatter + all-gather (half the link bytes)
    nt_under_comm: str = "auto"   # NT GEMM launch form while collectives hold CUs: "tiles" (one workgroup per tile: +3-8 % beside a kernel that holds CUs),
    #                               "persistent" (+30-55 % there, but 1.3 ms per step faster when the collectives are short: 1.026 x instead of 1.051 x of
    #                               the plain step with a one-rank group) or "auto": start on "tiles" and let the data-parallel schedule check time both
    #                               on the node it runs on (bit-identical results either way)
    force_dist: bool = False      # run the N>1 code path with a one-rank process group (tests on a one-GPU box)
    dp_four_streams: bool = True  # data-parallel schedule on FOUR streams -- caller, RCCL's, ONE side stream (text student and momentum chains share it),
    #                               weight-gradient -- so that no two of them share a hardware slot (streams take slots by first use, index mod 4): the
    #                               asynchronous weight-gradient stream and the off-path maintenance then run beside the exchange as on a single rank,
    #                               and a layer's slice is handed to RCCL from the weight-gradient stream (behind an event on the backward's stream)
    #                               instead of stalling the backward until its weight gradients are done.  False = round 5's schedule (five streams,
    #                               the weight-gradient stream idle during the exchange)
    schedule_check: bool = True   # data-parallel runs time their candidate schedules during their first 20 steps with an exchange (4 + 4 warm-up steps, then
    #                               twelve timed steps rotating between the candidates: SPMM._schedule_check_begin) and keep the single stream only if its
    #                               median is >= 13 % faster, the persistent NT launch (nt_under_comm = auto) only if >= 1.5 % faster: the stream-order
    #                               dependent 80-ms mode of EXPERIMENTS.md 1.4 cannot be ruled out on a node this package has never run on
    #                               (results are identical either way; bench.py runs these steps before its warm-up)
    probe_streams: bool = False   # diagnostic: at start-up of a data-parallel run, check that RCCL's stream and the compute streams sit on
    #                               different hardware queues and re-draw one that does not (spmm_amd/streams.py: the probe's own
    #                               streams change the queue order, and some orders cost 20 ms per step -- off by default)

    @classmethod
    def from_env(cls, **overrides) -> "EngineOptions":
        kw = {}
        for field, (var, parse) in _ENV.items():
            if var in os.environ:
                kw[field] = parse(os.environ[var])
        kw.update(overrides)
        o = cls(**kw)
        if isinstance(o.fused_xattn, bool):
            o.fused_xattn = "all" if o.fused_xattn else "off"
        if o.fused_xattn not in ("off", "nograd", "all"):
            raise ValueError(f"fused_xattn must be off, nograd or all, not {o.fused_xattn!r}")
        if o.grad_wire not in ("fp32", "bf16"):
            raise ValueError(f"grad_wire must be fp32 or bf16, not {o.grad_wire!r}")
        if o.nt_under_comm not in ("tiles", "persistent", "auto"):
            raise ValueError(f"nt_under_comm must be tiles, persistent or auto, not {o.nt_under_comm!r}")
        return o

    def replace(self, **kw) -> "EngineOptions":
        return dataclasses.replace(self, **kw)


# Debugging aids outside EngineOptions (they change no result): SPMM_DEBUG_SYNC=1 names every launch and drains the GPU after it
# (ops.py); SPMM_BENCH_WATCHDOG=<s> makes bench.py dump all Python stacks and exit non-zero after <s> seconds;
# SPMM_DIST_BACKEND=gloo lets the test harness put two ranks on one GPU (bench.py / pretrain.py); SPMM_DECODE_PER_BEAM=1 makes
# spmm_decode_attn take its one-wave-per-beam-row kernel instead of the one-wave-per-molecule one (csrc/decode.hip; same results up to the
# summation order).
