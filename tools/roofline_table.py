"""Every kernel family of the step at its dominant (S6 / full-model) shape against the chip peaks (MI355X_MICROARCH.md:
8 000 GB/s HBM3E, 2 500 TFLOP/s dense bf16 MFMA): algorithmic bytes (each tensor read / written once) or FLOPs per launch
divided by the measured launch time (HIP events, 20 launches after 3 warm-ups).  Writes a markdown table."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from spmm_amd import ops
from smi_sampler import Sampler            # tools/smi_sampler.py (this directory is sys.path[0])
BF = torch.bfloat16
# ROOFLINE_POWER=1: every row additionally runs back to back for ~1.2 s under the clock / power sampler -> columns MHz, W, and the
# fraction of the MFMA peak AT THE SUSTAINED CLOCK (2 500 TF/s x MHz / 2 400); written to gpurun_out/power_table.txt
POWER = os.environ.get("ROOFLINE_POWER") == "1"
dev = "cuda"
PEAK_BW, PEAK_TF = 8000.0, 2500.0


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


rows = []


def add(name, shape, fn, nbytes=None, flops=None, n=20):
    us = timeit(fn, n)
    gbs = nbytes / us / 1e3 if nbytes else None
    tfs = flops / us / 1e6 if flops else None
    bound = "HBM" if (flops is None or (nbytes and flops / nbytes < 300)) else "MFMA"
    frac = gbs / PEAK_BW if bound == "HBM" else tfs / PEAK_TF
    pw = None
    if POWER:
        reps = max(n, int(1.2e6 / max(us, 1.0)))
        with Sampler(period_s=0.02) as smp:
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
        pw = smp.summary()
        pw["us_sustained"] = e0.elapsed_time(e1) / reps * 1e3
    rows.append((name, shape, us, gbs, tfs, bound, frac, pw))
    extra = ""
    if pw and pw.get("clock_mhz"):
        tf_s = flops / pw["us_sustained"] / 1e6 if flops else None
        extra = f"  | sustained {pw['us_sustained']:8.1f} us  {pw['clock_mhz']:6.0f} MHz  {pw['power_w'] or 0:6.0f} W"
        if tf_s and bound == "MFMA":
            at_clk = tf_s / (PEAK_TF * pw["clock_mhz"] / 2400.0)
            extra += f"  {tf_s:7.1f} TF/s = {at_clk:5.3f} of the peak at that clock"
    print(f"{name:34s} {shape:28s} {us:9.1f} us  {'' if gbs is None else f'{gbs:7.0f} GB/s':>12s}  {'' if tfs is None else f'{tfs:7.1f} TF/s':>12s}  {bound:4s} {frac:6.3f}{extra}", flush=True)


M, H, I = 84256, 768, 3072
x = torch.randn(M, H, device=dev).to(BF); r = torch.randn(M, H, device=dev).to(BF)
g = torch.rand(H, device=dev) + 0.5; b = torch.randn(H, device=dev)
y = torch.empty_like(x); z = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
seed = torch.full((1,), 1234, dtype=torch.int64, device=dev)
dz = torch.empty_like(x); dx = torch.empty_like(x); dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); dxs = torch.zeros(H, device=dev)
row = M * H * 2
add("ln_fwd (dropout+residual+LN)", f"{M}x{H}", lambda: ops.ln_fwd(x, r, g, b, y, zout=z, mean=mean, rstd=rstd, dropout_p=0.1, seed=seed, salt=7), 4 * row)
add("ln_bwd (+dropout, dgamma, dbias)", f"{M}x{H}", lambda: ops.ln_bwd(x, z, mean, rstd, g, dz, dx=dx, dgamma=dg, dbeta=db, dropout_p=0.1, seed=seed, salt=7, dxsum=dxs), 4 * row)

# attention at the S6 shapes: 512 sequences, 12 heads
for nm, nseq, Lq, Lkv, cross in (("self PV", 512, 54, 54, False), ("self text", 512, 128, 128, False), ("cross PV->text", 512, 54, 128, True), ("cross text->PV", 512, 128, 54, True),
                                 ("self text, Lt = 256 (configs[4])", 512, 256, 256, False), ("cross PV->text, Lt = 256", 512, 54, 256, True),
                                 ("cross text->PV, Lt = 256", 512, 256, 54, True)):
    nH = 12
    q = torch.randn(nseq * Lq, 3 * H, device=dev).to(BF); kv = torch.randn(nseq * Lkv, 2 * H, device=dev).to(BF)
    Q, K, V = (q[:, :H], kv[:, :H], kv[:, H:]) if cross else (q[:, :H], q[:, H:2 * H], q[:, 2 * H:])
    O = torch.empty(nseq * Lq, H, device=dev, dtype=BF); lse = torch.empty(nseq, nH, Lq, device=dev)
    dO = torch.randn(nseq * Lq, H, device=dev).to(BF); dQ = torch.empty_like(O); dKV = torch.empty(nseq * Lkv, 2 * H, device=dev, dtype=BF)
    kw = dict(nseq=nseq, nH=nH, Lq=Lq, Lkv=Lkv, is_cross=cross, dropout_p=0.1, seed=seed, salt=3)
    fl = 4.0 * nseq * nH * Lq * Lkv * 64
    by = (2 * nseq * Lq * H + 2 * nseq * Lkv * H) * 2
    add(f"attn_fwd {nm}", f"{nseq} seq x {Lq}x{Lkv}", lambda: ops.attn_fwd(Q, K, V, O, lse, **kw), by, fl)
    add(f"attn_bwd {nm}", f"{nseq} seq x {Lq}x{Lkv}", lambda: ops.attn_bwd(Q, K, V, O, lse, dO, dQ, dKV[:, :H], dKV[:, H:], **kw),
        (4 * nseq * Lq * H + 4 * nseq * Lkv * H) * 2, 2.5 * fl)

# GEMMs
def gemm(Mg, N, K, epi=ops.EPI_BF16, two=False):
    A = torch.randn(Mg, K, device=dev).to(BF); W = torch.randn(N, K, device=dev).to(BF); bias = torch.zeros(N, device=dev)
    C = torch.empty(Mg, N, device=dev, dtype=BF); C2 = torch.empty(Mg, N, device=dev, dtype=BF) if two else None
    G = torch.randn(Mg, N, device=dev).to(BF) if epi in (ops.EPI_GELU_GRAD, ops.EPI_MUL) else None
    by = (Mg * K + N * K + Mg * N * (2 if two else 1) + (Mg * N if G is not None else 0)) * 2
    nm = {ops.EPI_BF16: "plain", ops.EPI_GELU: "erf-GELU (+pre-activation)", ops.EPI_GELU_DERIV: "erf-GELU + gelu' (FFN forward)",
          ops.EPI_MUL: "x G (FFN backward)", ops.EPI_GELU_GRAD: "x gelu'(G) (LM-head transform backward)"}[epi]
    add(f"gemm_nt {nm}", f"{Mg}x{N}x{K}", lambda: ops.gemm_nt(A, W, C, bias=None if G is not None else bias, epi=epi, C2=C2, G=G), by, 2.0 * Mg * N * K)
gemm(M, H, H); gemm(M, 3 * H, H); gemm(M, I, H, ops.EPI_GELU_DERIV, True); gemm(M, H, I); gemm(M, I, H, ops.EPI_MUL); gemm(M, I, H, ops.EPI_GELU, False)
for (N, K) in ((H, H), (I, H), (H, I)):
    A = torch.randn(M, N, device=dev).to(BF); B = torch.randn(M, K, device=dev).to(BF); C = torch.zeros(N, K, device=dev)
    add("gemm_tn (weight gradient)", f"{M}: {N}x{K}", lambda: ops.gemm_tn(A, B, C), (M * N + M * K) * 2 + N * K * 4, 2.0 * M * N * K)

# optimiser over the 144 M-parameter arena
n = 144_374_064
p = torch.randn(n, device=dev); gr = torch.randn(n, device=dev) * 1e-3; m1 = torch.zeros(n, device=dev); v1 = torch.zeros(n, device=dev)
sh = torch.empty(n, device=dev, dtype=BF); pm = p.clone(); shm = torch.empty(n, device=dev, dtype=BF)
lr = torch.full((1,), 5e-5, device=dev); nsq = torch.zeros(1, device=dev); step = torch.zeros(1, dtype=torch.int32, device=dev)
scal = torch.zeros(ops.adam_scalars_bytes() // 4, device=dev); nan = torch.zeros(1, dtype=torch.int32, device=dev)
def opt():
    nsq.zero_(); ops.grad_sqnorm(gr, nsq)
    ops.adamw_step(p, gr, m1, v1, sh, lr=lr, normsq=nsq, step=step, nan_flag=nan, scalars=scal)
add("grad norm + clip + AdamW (+bf16 shadow)", "144.4 M params", opt, n * (4 + 4 * 4 + 3 * 4 + 2), n=5)
add("EMA of the momentum arena (+shadow)", "144.4 M params", lambda: ops.ema_update(pm, p, shm, 0.995), n * (8 + 4 + 2), n=5)

# decode attention: 5000 rows, 12 heads, 100 cached positions
R, Lmax, t = 5000, 103, 100
qd = torch.randn(R, 3 * H, device=dev).to(BF); kc = torch.randn(R, Lmax, H, device=dev).to(BF); vc = torch.randn(R, Lmax, H, device=dev).to(BF)
anc = torch.arange(R, dtype=torch.int32, device=dev)[:, None].repeat(1, Lmax).contiguous(); od = torch.empty(R, H, device=dev, dtype=BF)
add("decode_attn (K/V cache gather)", f"{R} rows x {t} keys", lambda: ops.decode_attn(qd[:, :H], kc, vc, od, nH=12, Lkv=t, seq_stride=Lmax * H, tok_stride=H, anc=anc, group=5),
    R * 12 * t * 256)

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "roofline_table.md")
rows.sort(key=lambda r: 0)
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    f.write("| kernel | shape | us / launch | GB/s | TFLOP/s | bound | fraction of peak |\n|---|---|---|---|---|---|---|\n")
    for name, shape, us, gbs, tfs, bound, frac, _pw in rows:
        f.write(f"| {name} | {shape} | {us:.1f} | {'' if gbs is None else f'{gbs:.0f}'} | {'' if tfs is None else f'{tfs:.1f}'} | {bound} | {frac:.3f} |\n")
print("wrote", out)
if POWER:
    out2 = os.path.join(os.path.dirname(out), "power_table.txt")
    with open(out2, "w") as f:
        f.write("kernel family at its dominant shape, back-to-back launches for ~1.2 s each (tools/roofline_table.py, ROOFLINE_POWER=1; clock / power: tools/smi_sampler.py)\n")
        f.write(f"{'kernel':42s} {'shape':28s} {'us first':>9s} {'us sust.':>9s} {'MHz':>6s} {'W':>6s} {'TF/s':>8s} {'of 2.5 PF':>9s} {'at clock':>9s}\n")
        for name, shape, us, gbs, tfs, bound, frac, pw in rows:
            if not pw or not pw.get("clock_mhz"):
                f.write(f"{name:42s} {shape:28s} {us:9.1f}  (no clock / power source: {pw})\n")
                continue
            fl = tfs * us * 1e6 if tfs else None
            tf_s = fl / pw["us_sustained"] / 1e6 if fl else None
            at_clk = None if tf_s is None else tf_s / (PEAK_TF * pw["clock_mhz"] / 2400.0)
            f.write(f"{name:42s} {shape:28s} {us:9.1f} {pw['us_sustained']:9.1f} {pw['clock_mhz']:6.0f} {pw['power_w'] or 0:6.0f} "
                    f"{'' if tf_s is None else f'{tf_s:8.1f}':>8s} {'' if tf_s is None else f'{tf_s / PEAK_TF:9.3f}':>9s} "
                    f"{'' if at_clk is None else f'{at_clk:9.3f}':>9s}\n")
    print("wrote", out2)
