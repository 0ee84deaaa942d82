"""BASELINE.json configs[3] (next tier): k=5 beam decode of synthetic PVs, full-size random-init model, 1 GPU.
Reference semantics (whole-prefix recompute each step, one molecule at a time, d_pv2smiles_batched.py:18-59)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd import decode
from spmm_amd.config import SPMMConfig
from spmm_amd.model import SPMM
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(42)
m = SPMM(config=None, spmm_config=SPMMConfig(), no_train=True).eval()
props = torch.randn(n, 53)
decode.beam_search(m, props[0], k=5, max_steps=5)
torch.cuda.synchronize(); t0 = time.time(); steps = 0
for i in range(n):
    hyps = decode.beam_search(m, props[i], k=5, max_steps=100)
torch.cuda.synchronize(); dt = time.time() - t0
print(json.dumps({"metric": "PV->SMILES beam decode (k=5, <=100 steps, no KV cache)", "molecules": n, "seconds": round(dt, 2),
                  "molecules_per_s": round(n / dt, 3), "finished_hypotheses_last": len(hyps)}))
