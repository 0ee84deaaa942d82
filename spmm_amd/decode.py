"""PV -> SMILES k-beam decoding (BASELINE.json configs[3]) on the sub-module facades.

Restates `generate` (d_pv2smiles_single.py:26-51, deterministic top-k branch) and the beam bookkeeping of `evaluate`
(d_pv2smiles_batched.py:18-59) with the reference's semantics: every step re-runs the 12-layer causal text encoder with
cross-attention to the PV embeddings on the whole prefix of all k beams (no KV cache -- that, and batching several
molecules per launch, is the planned optimisation).  Works with anything exposing the reference's module API
(`property_embed`, `property_cls`, `property_encoder`, `text_encoder`), so tests drive it with the CPU oracle as well."""
from __future__ import annotations

from typing import List, Tuple

import torch

CLS_ID, SEP_ID = 2, 3           # vocab_bpe_300.txt:3-4


@torch.no_grad()
def next_token_topk(model, prop_embeds: torch.Tensor, text: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """d_pv2smiles_single.generate with stochastic=False, k given: log of the top-k next-token probabilities and their ids.
    text: [beams, t] ids (0 = PAD), prop_embeds: [1 or beams, 54, H]."""
    text_atts = torch.where(text == 0, 0, 1)
    prop_att_mask = torch.ones(prop_embeds.shape[:-1], dtype=torch.long, device=prop_embeds.device)
    logits = model.text_encoder(text, attention_mask=text_atts, encoder_hidden_states=prop_embeds,
                                encoder_attention_mask=prop_att_mask, return_dict=True, is_decoder=True, return_logits=True)[:, -1, :]
    p = torch.softmax(logits.float(), dim=-1)
    top = torch.topk(p, k=k, dim=-1)
    return torch.log(top.values), top.indices


@torch.no_grad()
def encode_properties(model, prop: torch.Tensor) -> torch.Tensor:
    """d_pv2smiles_batched.py:24-27: PV [B,53] -> prop_embeds [B,54,H] (no masking at inference)."""
    feat = model.property_embed(prop.unsqueeze(2))
    cls = model.property_cls
    properties = torch.cat([cls.expand(feat.size(0), -1, -1).to(feat.dtype).to(feat.device), feat], dim=1)
    return model.property_encoder(inputs_embeds=properties, return_dict=True).last_hidden_state


@torch.no_grad()
def beam_search(model, prop: torch.Tensor, k: int = 5, max_steps: int = 100) -> List[Tuple[float, List[int]]]:
    """One molecule (prop: [53]).  Returns up to k finished hypotheses (log-prob, token ids incl. CLS and SEP), best first."""
    prop_embeds = encode_properties(model, prop.reshape(1, -1))
    dev = prop_embeds.device
    product_input = torch.full((1, 1), CLS_ID, dtype=torch.long, device=dev)
    values, indices = next_token_topk(model, prop_embeds, product_input, k)
    product_input = torch.cat([torch.full((k, 1), CLS_ID, dtype=torch.long, device=dev), indices.squeeze(0).unsqueeze(-1)], dim=-1)
    current_p = values.squeeze(0)
    final: List[Tuple[float, torch.Tensor]] = []
    for _ in range(max_steps):
        values, indices = next_token_topk(model, prop_embeds, product_input, k)
        k2_p = current_p[:, None] + values
        product_input_k2 = torch.cat([product_input.unsqueeze(1).repeat(1, k, 1), indices.unsqueeze(-1)], dim=-1)
        ends = (indices == SEP_ID).nonzero(as_tuple=False)
        if ends.numel():
            for e in ends:
                final.append((float(k2_p[e[0], e[1]]), product_input_k2[e[0], e[1]].clone()))
                k2_p[e[0], e[1]] = -1e5
            if len(final) >= k:
                break
        current_p, flat = torch.topk(k2_p.flatten(), k)
        rows, cols = flat // k, flat % k
        product_input = product_input_k2[rows, cols]
    final = sorted(final, key=lambda x: x[0], reverse=True)[:k]
    return [(p, s.tolist()) for p, s in final]
