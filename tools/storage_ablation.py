"""Which bf16 STORES carry the deviation of each loss from the fp32 reference?  

The oracle's bf16 storage model (oracle/spmm_oracle.py `bf16_storage`) rounds the oracle's tensors to bf16 exactly where the HIP product writes
bf16 to HBM.  Here it is switched on ONE storage class at a time (`only={cls}`), and with everything BUT one class, at the benchmark shape --
full 12+6-layer H=768 model, B=128, Lt=128, queue 36 864, the batch / draws of tests/test_step_gpu.py::test_full_benchmark_batch_... --
and the four losses are compared with the fp32 oracle's.  CPU only (test infrastructure; no GPU, no product code).

    python tools/storage_ablation.py [B] > profiles/r06_storage_ablation.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import spmm_oracle as O  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    Lt = 128
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    ocfg = O.full_cfg()
    sd = O.init_state_dict(ocfg, seed=13)
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(7))

    def run(**kw):
        t0 = time.time()
        with torch.no_grad():
            if kw.get("fp32"):
                out = O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)
            else:
                with O.bf16_storage(only=kw.get("only"), weights=kw.get("weights")):
                    out = O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)
        return np.array([float(x) for x in out]), time.time() - t0

    names = ("loss_mlm", "5*loss_mpm", "loss_ita", "loss_itm")
    ref, dt = run(fp32=True)
    print(f"bf16 storage ablation of the oracle, full 12+6-layer H=768 model, B={B}, Lt={Lt}, queue {ocfg.queue_size}, dropout off, alpha 0.4 "
          f"({torch.get_num_threads()} threads, {dt:.0f} s per forward)")
    print(f"fp32 oracle losses: " + "  ".join(f"{n} {v:.6f}" for n, v in zip(names, ref)))
    print(f"\n|loss - fp32 oracle| with bf16 rounding at ...   {'  '.join(f'{n:>11s}' for n in names)}")
    rows = [("every storage point (the product's model)", None)]
    rows += [(f"ONLY {c}", {c}) for c in O.STORAGE_CLASSES]
    rows += [(f"all BUT {c}", set(O.STORAGE_CLASSES) - {c}) for c in O.STORAGE_CLASSES]
    for label, only in rows:
        got, _ = run(only=only)
        print(f"{label:48s} " + "  ".join(f"{abs(g - r):11.2e}" for g, r in zip(got, ref)), flush=True)
    # the weight shadows alone, by weight group (regex on the Linear's parameter prefix)
    groups = {"property_encoder(_m)": r"^property_encoder", "text layers 0-5 (+_m)": r"^text_encoder(_m)?\.bert\.encoder\.layer\.[0-5]\.",
              "text fusion layers 6-11 (+_m)": r"^text_encoder(_m)?\.bert\.encoder\.layer\.(6|7|8|9|10|11)\.",
              "fusion: self-attention": r"layer\.(6|7|8|9|10|11)\.attention\.", "fusion: cross-attention": r"crossattention",
              "fusion: FFN": r"layer\.(6|7|8|9|10|11)\.(intermediate|output)\.dense", "heads (cls / mtr / proj / itm)": r"^(?!.*encoder\.layer)",
              "student only (no _m)": r"^(property_encoder|text_encoder)\.", "momentum only": r"_m\."}
    print("\nbf16 rounding of the WEIGHT shadows only, by weight group:")
    for label, rx in groups.items():
        got, _ = run(only={"weights"}, weights=rx)
        print(f"ONLY weights of {label:32s} " + "  ".join(f"{abs(g - r):11.2e}" for g, r in zip(got, ref)), flush=True)


if __name__ == "__main__":
    main()
