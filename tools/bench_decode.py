"""BASELINE.json configs[3]: PV -> SMILES k=5 beam decode on synthetic PVs, full-size model, random-init weights.
Reports molecules/s for (a) the reference's cost model -- one molecule at a time, whole prefix re-run every step
(d_pv2smiles_batched.py:18-59) -- and (b) the batched K/V-cache decoder.  With random weights [SEP] rarely wins, so nearly
every molecule runs the full `--steps` positions: this is the worst case, not a typical SMILES length."""
import argparse, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spmm_amd.model import SPMM
from spmm_amd import decode
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import decode_oracle                      # the reference's sequential search: the baseline this tool times the batched decoder against

ap = argparse.ArgumentParser()
ap.add_argument("--molecules", type=int, default=200)
ap.add_argument("--chunk", type=int, default=200)
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--uncached", type=int, default=2, help="molecules to time on the per-molecule recompute path")
ap.add_argument("--graph", type=int, default=0, help="1: replay one captured decode position as a hipGraph")
a = ap.parse_args()
torch.manual_seed(0)
from spmm_amd.config import BertConfig, SPMMConfig
cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True),
                 prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1), embed_dim=256, queue_size=36864)
m = SPMM(spmm_config=cfg, no_train=True).eval()
m.store.refresh_shadows()
props = torch.randn(a.molecules, 53)
out = {"k": a.k, "max_steps": a.steps, "molecules": a.molecules, "chunk": a.chunk, "graph": a.graph}
decode.beam_search_batched(m, props[: min(8, a.molecules)], k=a.k, max_steps=4)        # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
nfin = 0
for i in range(0, a.molecules, a.chunk):
    res = decode.beam_search_batched(m, props[i:i + a.chunk], k=a.k, max_steps=a.steps, graph=bool(a.graph))
    nfin += sum(len(r) for r in res)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["cached_batched_molecules_per_s"] = round(a.molecules / dt, 2)
out["cached_batched_ms_per_position"] = round(dt / (a.steps + 1) / ((a.molecules + a.chunk - 1) // a.chunk) * 1e3, 3)
out["finished_hypotheses"] = nfin
if a.uncached > 0:
    decode_oracle.beam_search(m, props[0], k=a.k, max_steps=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(a.uncached):
        decode_oracle.beam_search(m, props[i], k=a.k, max_steps=a.steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["per_molecule_recompute_molecules_per_s"] = round(a.uncached / dt, 2)
print(json.dumps(out))
