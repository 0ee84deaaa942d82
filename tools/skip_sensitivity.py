"""What the step time is SENSITIVE to: run bench.py's timed loop with one family of launches left out (the C entry returns before its
launch; results are garbage, only the clock is read) and print the step time beside the full step's.  The difference is what that
family costs the step AS SCHEDULED (overlap with the other streams included), which is not its kernel time.  A measuring tool only.

    python tools/skip_sensitivity.py                      # every family below, one after the other (child processes)
    python tools/skip_sensitivity.py spmm_attn_bwd,...    # one run with these C entries skipped"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = {
    "nothing": "",
    "attention forward": "spmm_attn_fwd",
    "attention backward": "spmm_attn_bwd",
    "LayerNorm forward": "spmm_ln_fwd",
    "LayerNorm backward": "spmm_ln_bwd",
    "weight gradients (TN GEMM)": "spmm_gemm_tn",
    "column sums": "spmm_colsum_bf16",
    "AdamW + clip + EMA": "spmm_adamw_step,spmm_ema_update,spmm_grad_sqnorm",
    "FFN GEMMs' second output / factor (plain epilogues instead)": "EPI_PLAIN",
    "all NT GEMMs": "spmm_gemm_nt",
}

if len(sys.argv) > 1 or os.environ.get("SKIP_CHILD"):
    skip = set(filter(None, (sys.argv[1] if len(sys.argv) > 1 else "").split(",")))
    sys.path.insert(0, ROOT)
    from spmm_amd import ops
    orig = ops._call

    def call(name, *args):
        if name in skip:
            return None
        if "EPI_PLAIN" in skip and name == "spmm_gemm_nt" and args[19] in (ops.EPI_GELU_DERIV, ops.EPI_MUL, ops.EPI_GELU):
            args = list(args)                                         # same GEMM with the plain bf16 epilogue: no second output, no factor, no GELU
            args[19] = ops.EPI_BF16
            args[13], args[14], args[17], args[18] = None, 0, None, 0
        return orig(name, *args)

    ops._call = call
    import runpy
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-kernel-timing", "--steps", "20", "--warmup", "5"]
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
else:
    base = None
    print(f"{'launches left out':34s} {'ms/step':>8s} {'saved':>7s}")
    for label, names in FAMILIES.items():
        r = subprocess.run([sys.executable, os.path.abspath(__file__), names or ","], capture_output=True, text=True, cwd=ROOT,
                           env=dict(os.environ, SKIP_CHILD="1"))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not line:
            print(f"{label:34s} failed: {r.stderr.strip().splitlines()[-1] if r.stderr.strip() else r.returncode}")
            continue
        ms = json.loads(line[-1])["ms_per_step"]
        base = ms if base is None else base
        print(f"{label:34s} {ms:8.2f} {base - ms:7.2f}", flush=True)
