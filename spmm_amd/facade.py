"""Sub-module facades: the call signatures external callers of the reference use on an SPMM instance (SURVEY.md section 8b;
e.g. d_pv2smiles_batched.py:25-27, d_smiles2pv.py:15-25, d_pv2smiles_single.py:26-44):

    model.property_embed(x)                                   nn.Linear(1, H)            SPMM_models.py:36
    model.property_encoder(inputs_embeds=, is_decoder=, return_dict=True).last_hidden_state            :38, :90, :242
    model.text_encoder.bert(input_ids | encoder_embeds=, attention_mask=, encoder_hidden_states=,
                            encoder_attention_mask=, is_decoder=, mode=, return_dict=True).last_hidden_state   xbert.py:950-1091
    model.text_encoder(input_ids, attention_mask=, encoder_hidden_states=, encoder_attention_mask=,
                       is_decoder=True, return_logits=True)    -> logits                     xbert.py:1377-1428
    model.property_proj / text_proj / itm_head / property_mtr_head (callables)

They run the same HIP kernels as the training step (inference: no tape, dropout off unless the model is in train mode).
Sequences up to 128 tokens run in single attention launches (a whole K/V panel on chip); longer ones (up to the position
table) through the chunked path of `ops.attn_fwd_long`."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch

from . import ops
from .engine import BF, Group


def _i32(mask, nseq, L, dev):
    if mask is None:
        return None
    return mask.to(dev).to(torch.int32).reshape(nseq, L).contiguous()


class BertFacade:
    """`BertModel.forward` on the engine (xbert.py:950-1091)."""

    def __init__(self, model, pfx: str, cfg, has_cross: bool):
        self._m, self.pfx, self.config, self.has_cross = model, pfx, cfg, has_cross

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, inputs_embeds=None, encoder_embeds=None, encoder_hidden_states=None,
                 encoder_attention_mask=None, return_dict=True, is_decoder=False, mode="multi_modal", **unused):
        eng, c, dev = self._m.engine, self.config, self._m.device_
        eng.train_mode = self._m.training
        H = c.hidden_size
        if encoder_embeds is not None:
            nseq, L = encoder_embeds.shape[:2]
            x = encoder_embeds.to(dev).to(BF).reshape(nseq * L, H).contiguous()
        elif inputs_embeds is not None:
            nseq, L = inputs_embeds.shape[:2]
            x = eng.embed_generic(self.pfx, c, inputs_embeds.to(dev).to(torch.float32).reshape(nseq * L, H).contiguous(), nseq, L)
        elif input_ids is not None:
            nseq, L = input_ids.shape
            x, _ = eng.embed_text(self.pfx, c, input_ids.to(dev).to(torch.int32).contiguous(), nseq, L, False)
        else:
            raise ValueError("You have to specify either input_ids or inputs_embeds or encoder_embeds")
        if L > c.max_position_embeddings:
            raise ValueError(f"sequence length {L} exceeds the {c.max_position_embeddings} position embeddings")
        kv, Lkv, kvm = None, 0, None
        if encoder_hidden_states is not None:
            Lkv = encoder_hidden_states.shape[1]
            if encoder_hidden_states.shape[0] == 1 and nseq > 1:      # one PV sequence shared by k beams: the reference relies on
                encoder_hidden_states = encoder_hidden_states.expand(nseq, -1, -1)   # matmul broadcasting (d_pv2smiles_single.py:29-35)
                if encoder_attention_mask is not None:
                    encoder_attention_mask = encoder_attention_mask.expand(nseq, -1)
            kv = encoder_hidden_states.to(dev).to(BF).reshape(nseq * Lkv, H).contiguous()
            kvm = _i32(encoder_attention_mask, nseq, Lkv, dev)
        lo, hi = {"text": (0, c.fusion_layer), "fusion": (c.fusion_layer, c.num_hidden_layers),
                  "multi_modal": (0, c.num_hidden_layers)}[mode]
        g = [Group(0, nseq, L, _i32(attention_mask, nseq, L, dev), 0 if is_decoder else nseq, kv=kv, Lkv=Lkv, kv_mask=kvm)]
        if self.has_cross and hi > c.fusion_layer and kv is None:
            raise AssertionError("encoder_hidden_states must be given for cross-attention layers")     # xbert.py:495
        y, _ = eng.stack_fwd(self.pfx, c, range(lo, hi), self.has_cross, x, g, False)
        out = y.view(nseq, L, H).float()
        return SimpleNamespace(last_hidden_state=out) if return_dict else (out,)


class MaskedLMFacade:
    """`BertForMaskedLM` (xbert.py:1352-1449): `.bert` + the tied LM head; only the `return_logits=True` path is used."""

    def __init__(self, model, pfx: str, cfg):
        self._m, self.pfx, self.config = model, pfx, cfg
        self.bert = BertFacade(model, pfx + "bert.", cfg, True)

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None, return_dict=True,
                 is_decoder=False, mode="multi_modal", return_logits=False, encoder_embeds=None, **unused):
        h = self.bert(input_ids, attention_mask=attention_mask, encoder_embeds=encoder_embeds, encoder_hidden_states=encoder_hidden_states,
                      encoder_attention_mask=encoder_attention_mask, is_decoder=is_decoder, mode=mode).last_hidden_state
        nseq, L, H = h.shape
        logits, _ = self._m.engine.lm_head_fwd(self.pfx, self.config, h.to(BF).reshape(nseq * L, H).contiguous(), False)
        logits = logits.view(nseq, L, -1)
        if return_logits:
            return logits
        return SimpleNamespace(logits=logits)


class LinearFacade:
    """nn.Linear-like callable over parameters of the flat arena: the in-tree small-head kernel (csrc/losses.hip rows_linear,
    fp32 weights and accumulation) -- these heads see a handful of pooled rows."""

    def __init__(self, model, name: str):
        self._m, self.name = model, name

    @property
    def weight(self):
        return self._m._parameters[self.name + ".weight"]

    @property
    def bias(self):
        return self._m._parameters[self.name + ".bias"]

    @torch.no_grad()
    def __call__(self, x):
        W = self.weight.detach()
        N, K = W.shape
        x2 = x.to(W.device).float().reshape(-1, K).contiguous()
        out = torch.empty(x2.shape[0], N, dtype=torch.float32, device=W.device)
        ops.rows_linear(x2, W.contiguous(), self.bias.detach(), out)
        return out.view(*x.shape[:-1], N)


class MtrHeadFacade:
    """property_mtr_head = Sequential(Linear, GELU, LayerNorm, Linear(H,1)) SPMM_models.py:39-42 -- the kernels the training step
    runs for it (step.py: bf16 GEMM with the erf-GELU epilogue, the LayerNorm row kernel), then the fp32 Linear(H, 1)."""

    def __init__(self, model):
        self._m = model

    @torch.no_grad()
    def __call__(self, x):
        m = self._m
        P, S = m._parameters, m.store
        H = x.shape[-1]
        x2 = x.to(m.device_).to(BF).reshape(-1, H).contiguous()
        rows = x2.shape[0]
        h = torch.empty(rows, H, dtype=BF, device=x2.device)
        ops.gemm_nt(x2, S.wb("property_mtr_head.0.weight"), h, bias=S.w("property_mtr_head.0.bias"), epi=ops.EPI_GELU)
        y = torch.empty_like(h)
        ops.ln_fwd(h, None, S.w("property_mtr_head.2.weight"), S.w("property_mtr_head.2.bias"), y, eps=m.cfg.text.layer_norm_eps)
        out = torch.empty(rows, 1, dtype=torch.float32, device=x2.device)
        ops.rows_linear(y, P["property_mtr_head.3.weight"].detach().contiguous(), P["property_mtr_head.3.bias"].detach(), out)
        return out.view(*x.shape[:-1], 1)
