"""TEST INFRASTRUCTURE (never imported by spmm_amd/, bench.py's timed region or any product path): the sequential, one-molecule,
whole-prefix-per-step PV -> SMILES beam search the reference runs -- the yard-stick the batched, K/V-cached decoder of
spmm_amd/decode.py is checked against hypothesis for hypothesis.

Restates `generate` (d_pv2smiles_single.py:26-44, deterministic top-k branch) and the beam bookkeeping of `evaluate`
(d_pv2smiles_batched.py:23-57): every step re-runs the 12-layer causal text encoder with cross-attention to the PV embeddings on
the WHOLE prefix of all k beams; a candidate that ends in [SEP] is recorded and struck out (-1e5), the search stops once k
hypotheses are finished, otherwise the k best of the k*k candidates survive.  Works with anything exposing the reference's
module API (`text_encoder(...)`), i.e. the CPU oracle model and the HIP model's facades alike."""
from __future__ import annotations

from typing import List, Tuple

import torch

CLS_ID, SEP_ID = 2, 3           # vocab_bpe_300.txt:3-4


@torch.no_grad()
def next_token_topk(model, prop_embeds: torch.Tensor, prefix: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """log-probabilities and ids of the k most probable next tokens after `prefix` ([beams, t] ids, 0 = PAD) given the PV
    embeddings ([1 or beams, 54, H]) -- one whole-prefix forward of the causal text encoder (d_pv2smiles_single.py:29-44)."""
    pad_mask = (prefix != 0).long()
    pv_mask = torch.ones(prop_embeds.shape[:-1], dtype=torch.long, device=prop_embeds.device)
    last = model.text_encoder(prefix, attention_mask=pad_mask, encoder_hidden_states=prop_embeds, encoder_attention_mask=pv_mask,
                              return_dict=True, is_decoder=True, return_logits=True)[:, -1, :]
    best = torch.topk(torch.softmax(last.float(), dim=-1), k=k, dim=-1)
    return torch.log(best.values), best.indices


@torch.no_grad()
def beam_search(model, prop: torch.Tensor, k: int = 5, max_steps: int = 100, encode=None) -> List[Tuple[float, List[int]]]:
    """One molecule (prop: [53]) -> up to k finished hypotheses (log-prob, token ids incl. CLS and SEP), best first.
    `encode` maps PV -> prop_embeds (default: spmm_amd.decode.encode_properties, which only calls the module API)."""
    if encode is None:
        from spmm_amd.decode import encode_properties as encode
    pv = encode(model, prop.reshape(1, -1))
    dev = pv.device
    start = torch.full((1, 1), CLS_ID, dtype=torch.long, device=dev)
    score, tok = next_token_topk(model, pv, start, k)
    beams = torch.cat([start.expand(k, 1), tok.reshape(k, 1)], dim=1)              # k prefixes [CLS, t1]
    beam_lp = score.reshape(k)
    done: List[Tuple[float, torch.Tensor]] = []
    for _ in range(max_steps):
        score, tok = next_token_topk(model, pv, beams, k)                          # [k, k] continuations of every beam
        cand_lp = beam_lp[:, None] + score
        cand = torch.cat([beams[:, None, :].expand(k, k, beams.shape[1]), tok[:, :, None]], dim=2)
        for b, j in (tok == SEP_ID).nonzero(as_tuple=False).tolist():            # row-major, as the reference walks them
            done.append((float(cand_lp[b, j]), cand[b, j].clone()))
            cand_lp[b, j] = -1e5
        if len(done) >= k:
            break
        beam_lp, pick = torch.topk(cand_lp.reshape(-1), k)
        beams = cand.reshape(k * k, -1)[pick]
    done.sort(key=lambda h: h[0], reverse=True)
    return [(lp, ids.tolist()) for lp, ids in done[:k]]
