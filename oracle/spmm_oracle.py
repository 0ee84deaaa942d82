"""CPU oracle for the SPMM pretraining step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32, purely functional restatement of the reference's algorithm for
the hot path (SPMM.forward / training_step and the xbert.py classes they drive).  It
operates on a flat ``state_dict`` (name -> tensor) that uses the reference's key
names and [out,in] layouts, so the same dict can be loaded into the reference module,
into this oracle and into the HIP product.

Who may import this: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- as the checker / the timed CPU baseline only.  Nothing under
``spmm_amd/`` imports it; the product path fails loudly without its HIP library.

Parity status: PINNED.  ``oracle/make_golden.py`` runs the real reference (imported
from /root/reference through ``oracle/ref_shim.py``) in the dev container and commits
inputs / recorded random draws / outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file against those vectors (<=1e-5).
The reference itself ships no tests or golden vectors.

Besides the fp32 restatement there is a bf16 STORAGE MODEL (``with bf16_storage():``,
below): the same functions with their tensors rounded to bfloat16 where the HIP product
stores bfloat16.  It is a second yard-stick for tests -- it separates deviations that
any bf16-storing implementation shares from kernel error -- not a second oracle: the
golden vectors pin the fp32 path only.

Every function cites the reference file:line it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import json
import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------- config
@dataclass
class BertCfg:
    """Subset of config_bert.json / config_bert_property.json the path reads."""
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    fusion_layer: int = 6
    vocab_size: int = 300
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    encoder_width: int = 768
    initializer_range: float = 0.02

    @staticmethod
    def from_json(path: str) -> "BertCfg":
        d = json.load(open(path))
        keys = BertCfg.__dataclass_fields__.keys()
        return BertCfg(**{k: d[k] for k in keys if k in d})


@dataclass
class SPMMCfg:
    text: BertCfg = field(default_factory=BertCfg)
    prop: BertCfg = field(default_factory=lambda: BertCfg(num_hidden_layers=6, vocab_size=1))
    embed_dim: int = 256
    temp: float = 0.07
    queue_size: int = 36864
    momentum: float = 0.995
    alpha: float = 0.4
    n_props: int = 53


def tiny_cfg() -> SPMMCfg:
    """The plumbing configuration of BASELINE.json configs[0] (H=128, 2 text layers with
    fusion_layer=1, 1 PV layer, E=64, Q=16)."""
    t = BertCfg(hidden_size=128, num_attention_heads=2, intermediate_size=512,
                num_hidden_layers=2, fusion_layer=1, encoder_width=128)
    p = BertCfg(hidden_size=128, num_attention_heads=2, intermediate_size=512,
                num_hidden_layers=1, fusion_layer=1, encoder_width=128, vocab_size=1)
    return SPMMCfg(text=t, prop=p, embed_dim=64, queue_size=16)


def full_cfg() -> SPMMCfg:
    """config_bert.json + config_bert_property.json + SPMM_pretrain.py:51-65."""
    return SPMMCfg()


# ------------------------------------------------------------------- state_dict spec
def _bert_keys(prefix: str, c: BertCfg, with_cross: bool) -> List[Tuple[str, tuple, str]]:
    """(name, shape, kind) in the reference's state_dict order (xbert.py:173-534)."""
    H, I = c.hidden_size, c.intermediate_size
    out = [
        (prefix + "embeddings.position_ids", (1, c.max_position_embeddings), "posid"),
        (prefix + "embeddings.word_embeddings.weight", (c.vocab_size, H), "emb"),
        (prefix + "embeddings.position_embeddings.weight", (c.max_position_embeddings, H), "emb"),
        (prefix + "embeddings.token_type_embeddings.weight", (c.type_vocab_size, H), "emb"),
        (prefix + "embeddings.LayerNorm.weight", (H,), "ln_w"),
        (prefix + "embeddings.LayerNorm.bias", (H,), "ln_b"),
    ]
    for i in range(c.num_hidden_layers):
        lp = f"{prefix}encoder.layer.{i}."
        blocks = ["attention"]
        if with_cross and i >= c.fusion_layer:
            blocks.append("crossattention")
        for blk in blocks:
            kin = c.encoder_width if blk == "crossattention" else H
            out += [
                (lp + blk + ".self.query.weight", (H, H), "lin_w"),
                (lp + blk + ".self.query.bias", (H,), "lin_b"),
                (lp + blk + ".self.key.weight", (H, kin), "lin_w"),
                (lp + blk + ".self.key.bias", (H,), "lin_b"),
                (lp + blk + ".self.value.weight", (H, kin), "lin_w"),
                (lp + blk + ".self.value.bias", (H,), "lin_b"),
                (lp + blk + ".output.dense.weight", (H, H), "lin_w"),
                (lp + blk + ".output.dense.bias", (H,), "lin_b"),
                (lp + blk + ".output.LayerNorm.weight", (H,), "ln_w"),
                (lp + blk + ".output.LayerNorm.bias", (H,), "ln_b"),
            ]
        out += [
            (lp + "intermediate.dense.weight", (I, H), "lin_w"),
            (lp + "intermediate.dense.bias", (I,), "lin_b"),
            (lp + "output.dense.weight", (H, I), "lin_w"),
            (lp + "output.dense.bias", (H,), "lin_b"),
            (lp + "output.LayerNorm.weight", (H,), "ln_w"),
            (lp + "output.LayerNorm.bias", (H,), "ln_b"),
        ]
    return out


def _mlm_keys(prefix: str, c: BertCfg):
    H, V = c.hidden_size, c.vocab_size
    return [
        (prefix + "cls.predictions.bias", (V,), "lin_b"),
        (prefix + "cls.predictions.transform.dense.weight", (H, H), "lin_w"),
        (prefix + "cls.predictions.transform.dense.bias", (H,), "lin_b"),
        (prefix + "cls.predictions.transform.LayerNorm.weight", (H,), "ln_w"),
        (prefix + "cls.predictions.transform.LayerNorm.bias", (H,), "ln_b"),
        (prefix + "cls.predictions.decoder.weight", (V, H), "tied_w"),
        (prefix + "cls.predictions.decoder.bias", (V,), "tied_b"),
    ]


def state_spec(cfg: SPMMCfg) -> List[Tuple[str, tuple, str]]:
    """All state_dict entries in the reference's order (SPMM_models.py:17-77; 758
    entries at full size, 178 at tiny_cfg -- SURVEY.md section 5)."""
    H, E, Q = cfg.text.hidden_size, cfg.embed_dim, cfg.queue_size
    s: List[Tuple[str, tuple, str]] = [
        ("property_cls", (1, 1, H), "zero"), ("property_mask", (1, 1, H), "zero"),
        ("temp", (), "temp"), ("prop_queue", (E, Q), "queue"), ("text_queue", (E, Q), "queue"),
        ("queue_ptr", (1,), "ptr"),
    ]
    s += _bert_keys("text_encoder.bert.", cfg.text, True) + _mlm_keys("text_encoder.", cfg.text)
    s += [("property_proj.weight", (E, H), "lin_w"), ("property_proj.bias", (E,), "lin_b"),
          ("text_proj.weight", (E, H), "lin_w"), ("text_proj.bias", (E,), "lin_b"),
          ("itm_head.weight", (2, 2 * H), "lin_w"), ("itm_head.bias", (2,), "lin_b"),
          ("property_embed.weight", (H, 1), "lin_w"), ("property_embed.bias", (H,), "lin_b")]
    s += _bert_keys("property_encoder.", cfg.prop, False)
    s += [("property_mtr_head.0.weight", (H, H), "lin_w"), ("property_mtr_head.0.bias", (H,), "lin_b"),
          ("property_mtr_head.2.weight", (H,), "ln_w"), ("property_mtr_head.2.bias", (H,), "ln_b"),
          ("property_mtr_head.3.weight", (1, H), "lin_w"), ("property_mtr_head.3.bias", (1,), "lin_b")]
    s += _bert_keys("property_encoder_m.", cfg.prop, False)
    s += [("property_proj_m.weight", (E, H), "lin_w"), ("property_proj_m.bias", (E,), "lin_b")]
    s += _bert_keys("text_encoder_m.bert.", cfg.text, True) + _mlm_keys("text_encoder_m.", cfg.text)
    s += [("text_proj_m.weight", (E, H), "lin_w"), ("text_proj_m.bias", (E,), "lin_b")]
    return s


MOMENTUM_PAIRS = [("property_encoder.", "property_encoder_m."), ("property_proj.", "property_proj_m."),
                  ("text_encoder.", "text_encoder_m."), ("text_proj.", "text_proj_m.")]
"""SPMM_models.py:56-60 model_pairs."""


def momentum_twin(name: str) -> Optional[str]:
    for a, b in MOMENTUM_PAIRS:
        if name.startswith(a):
            return b + name[len(a):]
    return None


def is_buffer(name: str) -> bool:
    return name.endswith("position_ids") or name in ("prop_queue", "text_queue", "queue_ptr")


def is_momentum(name: str) -> bool:
    return any(name.startswith(b) for _, b in MOMENTUM_PAIRS)


def trainable_names(cfg: SPMMCfg) -> List[str]:
    """Names handed to AdamW (SPMM_models.py:340 `self.parameters()` minus frozen `_m`
    twins :51-54), tied tensors listed once (decoder.weight / decoder.bias are aliases)."""
    out = []
    for n, _, kind in state_spec(cfg):
        if is_buffer(n) or is_momentum(n) or kind in ("tied_w", "tied_b"):
            continue
        out.append(n)
    return out


def closed_form_state_dict(cfg: SPMMCfg, scale: float = 0.08) -> SD:
    """Deterministic weights with no RNG stream to match:
    entry k, flat element i ->  scale*sin(0.37*i + k) (LayerNorm weight 1 + that).
    Momentum twins start as copies (SPMM_models.py:259-263), tied tensors alias."""
    sd: SD = {}
    for k, (name, shape, kind) in enumerate(state_spec(cfg)):
        n = int(math.prod(shape)) if shape else 1
        i = torch.arange(n, dtype=torch.float64)
        w = (scale * torch.sin(0.37 * i + k)).to(torch.float32).reshape(shape)
        if kind == "posid":
            w = torch.arange(shape[1]).expand(1, -1).clone()
        elif kind == "ptr":
            w = torch.zeros(1, dtype=torch.long)
        elif kind == "temp":
            w = torch.tensor(cfg.temp)
        elif kind == "ln_w":
            w = 1.0 + w
        elif kind == "queue":
            w = F.normalize(torch.sin(0.61 * i + 3 * k).to(torch.float32).reshape(shape), dim=0)
        sd[name] = w
    _finish_aliases(sd, cfg)
    return sd


def init_state_dict(cfg: SPMMCfg, seed: int = 0) -> SD:
    """Random init with the reference's distributions: BERT Linear/Embedding weights
    N(0, initializer_range), LayerNorm 1/0, biases 0 (xbert.py:742-752); the SPMM-level
    nn.Linear layers keep torch's default init (SPMM_models.py:31-42); cls/mask tokens 0
    (:43-44); queues randn normalised over dim 0 (:72-77)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for name, shape, kind in state_spec(cfg):
        in_bert = "encoder." in name or ".bert." in name or ".cls." in name
        if kind == "posid":
            w = torch.arange(shape[1]).expand(1, -1).clone()
        elif kind == "ptr":
            w = torch.zeros(1, dtype=torch.long)
        elif kind == "temp":
            w = torch.tensor(cfg.temp)
        elif kind == "zero":
            w = torch.zeros(shape)
        elif kind == "queue":
            w = F.normalize(torch.randn(shape, generator=g), dim=0)
        elif kind == "ln_w":
            w = torch.ones(shape)
        elif kind == "ln_b":
            w = torch.zeros(shape)
        elif kind in ("emb",) or (kind == "lin_w" and in_bert):
            w = torch.randn(shape, generator=g) * cfg.text.initializer_range
        elif kind == "lin_w":
            bound = 1.0 / math.sqrt(shape[-1])
            w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "lin_b":
            if in_bert:
                w = torch.zeros(shape)
            else:
                fan_in = {"property_embed.bias": 1, "itm_head.bias": 2 * cfg.text.hidden_size}.get(
                    name, cfg.text.hidden_size)
                w = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        else:  # tied_* filled below
            w = torch.zeros(shape)
        sd[name] = w
    _finish_aliases(sd, cfg)
    return sd


def _finish_aliases(sd: SD, cfg: SPMMCfg) -> None:
    for name in list(sd):
        twin = momentum_twin(name)
        if twin is not None and not is_buffer(name):
            sd[twin] = sd[name].clone()
    for p in ("text_encoder.", "text_encoder_m."):  # xbert.py:691, :1362-1368 (tie_weights)
        sd[p + "cls.predictions.decoder.weight"] = sd[p + "bert.embeddings.word_embeddings.weight"]
        sd[p + "cls.predictions.decoder.bias"] = sd[p + "cls.predictions.bias"]


# ------------------------------------------------------------------------ bf16 storage model
# `with bf16_storage():` makes the oracle round its tensors to bfloat16 at exactly the points where the HIP product STORES bf16
# (spmm_amd/engine.py, default options): GEMM weights (the bf16 shadows) and every GEMM / LayerNorm / attention output written to
# HBM, the un-normalised softmax numerators that feed the second attention MFMA.  Everything else stays as in the fp32 oracle:
# accumulation, biases, LayerNorm statistics and affine parameters, softmax sums, the loss heads' arithmetic.  It is the yard-stick
# that separates "deviation caused by bf16 storage" (shared by this model and the product) from "kernel error" (what is left
# between them): tests/test_step_gpu.py::test_losses_match_the_bf16_storage_model_of_the_oracle.  Forward only, dropout off.
# Round 5: `bf16_storage(backward=True)` extends the model to the BACKWARD: the gradient that autograd sends back through one of those
# storage points is rounded to bf16 too -- the product materialises d(loss)/d(that tensor) as a bf16 tensor between two of its kernels
# (data-gradient GEMM / LayerNorm-backward / attention-backward outputs), while weight and bias gradients, the hub gradients of the
# shared key / value sources and all accumulation stay fp32 (engine.py, step.py).
# Round 6: `bf16_storage(only={...})` rounds ONE (or a few) storage classes and leaves the others in fp32 -- the ablation that says which
# stores carry a loss's deviation (tools/storage_ablation.py -> profiles/r06_storage_ablation.txt).  Classes: "weights" (bf16 GEMM shadows),
# "qkv" (query / key / value projections), "softmax_e" (the un-normalised numerators fed to the second attention MFMA), "attn_ctx" (attention
# context), "attn_proj" (attention output projection, before the residual LayerNorm), "ffn_up" (GELU output), "ffn_down" (FFN output
# projection, before the residual LayerNorm), "ln" (LayerNorm outputs: the residual stream), "heads" (LM-head transform, MPM head).
STORAGE_CLASSES = ("weights", "qkv", "softmax_e", "attn_ctx", "attn_proj", "ffn_up", "ffn_down", "ln", "heads")
_BF16_STORAGE = False
_BF16_STORAGE_BWD = False
_BF16_ONLY = None
_BF16_WEIGHTS = None          # regex on the Linear's parameter prefix: which weight shadows are rounded (None = all)


class bf16_storage:
    def __init__(self, backward: bool = False, only=None, weights=None):
        self.backward = backward
        self.weights = None if weights is None else re.compile(weights)
        self.only = None if only is None else frozenset(only)
        if self.only is not None and not self.only <= set(STORAGE_CLASSES):
            raise ValueError(f"unknown storage classes {sorted(self.only - set(STORAGE_CLASSES))}")

    def __enter__(self):
        global _BF16_STORAGE, _BF16_STORAGE_BWD, _BF16_ONLY, _BF16_WEIGHTS
        self._old, _BF16_STORAGE = (_BF16_STORAGE, _BF16_STORAGE_BWD, _BF16_ONLY, _BF16_WEIGHTS), True
        _BF16_STORAGE_BWD, _BF16_ONLY, _BF16_WEIGHTS = self.backward, self.only, self.weights

    def __exit__(self, *exc):
        global _BF16_STORAGE, _BF16_STORAGE_BWD, _BF16_ONLY, _BF16_WEIGHTS
        _BF16_STORAGE, _BF16_STORAGE_BWD, _BF16_ONLY, _BF16_WEIGHTS = self._old


class _RoundBothWays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _lin_class(p: str) -> str:
    """Storage class of a Linear's output from its parameter prefix."""
    if p.endswith((".self.query", ".self.key", ".self.value")):
        return "qkv"
    if p.endswith(("attention.output.dense",)):                      # (attention / crossattention)
        return "attn_proj"
    if p.endswith("intermediate.dense"):
        return "ffn_up"
    if p.endswith("output.dense"):
        return "ffn_down"
    return "heads"


def _st(x: Tensor, cls: str = "heads") -> Tensor:
    """A tensor as the product holds it in HBM."""
    if not _BF16_STORAGE or (_BF16_ONLY is not None and cls not in _BF16_ONLY):
        return x
    if _BF16_STORAGE_BWD and x.requires_grad:
        return _RoundBothWays.apply(x)
    return x.to(torch.bfloat16).to(torch.float32)


# ------------------------------------------------------------------------ xbert.py
def _lin(sd: SD, p: str, x: Tensor, act=None, f32_out: bool = False, f32_w: bool = False) -> Tensor:
    """nn.Linear (+ fused activation).  bf16 storage model: bf16 weight shadow unless `f32_w` (the small loss heads read the fp32
    master), fp32 accumulation + bias + activation, bf16 output unless `f32_out` (logits, feature projections, head outputs)."""
    w = sd[p + ".weight"]
    if _BF16_WEIGHTS is not None and not _BF16_WEIGHTS.search(p):
        f32_w = True
    y = F.linear(x, w if f32_w else _st(w, "weights"), sd[p + ".bias"])
    if act is not None:
        y = act(y)
    return y if f32_out else _st(y, _lin_class(p))


def _ln(sd: SD, p: str, x: Tensor, eps: float) -> Tensor:
    """(the product's LayerNorm kernel forms the pre-norm sum in fp32 registers: only the OUTPUT is a storage point)"""
    return _st(F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps), "ln")


def _drop(x: Tensor, p: float, train: bool) -> Tensor:
    return F.dropout(x, p, train) if (train and p > 0) else x


def embeddings(sd: SD, p: str, c: BertCfg, input_ids=None, inputs_embeds=None, train=False) -> Tensor:
    """BertEmbeddings.forward xbert.py:193-220 (token_type 0, absolute positions)."""
    if inputs_embeds is None:
        # nn.Embedding(..., padding_idx=config.pad_token_id=0) xbert.py:178: lookups of PAD get no gradient
        inputs_embeds = F.embedding(input_ids, sd[p + "embeddings.word_embeddings.weight"], padding_idx=0)
    L = inputs_embeds.shape[1]
    e = inputs_embeds + sd[p + "embeddings.token_type_embeddings.weight"][0]
    e = e + sd[p + "embeddings.position_embeddings.weight"][:L]
    e = _ln(sd, p + "embeddings.LayerNorm", e, c.layer_norm_eps)
    return _drop(e, c.hidden_dropout_prob, train)


def extended_self_mask(mask: Tensor, is_decoder: bool) -> Tensor:
    """get_extended_attention_mask xbert.py:889-948 -> additive 0 / -10000."""
    m = mask.to(torch.float32)
    if is_decoder:
        L = m.shape[1]
        ids = torch.arange(L)
        causal = (ids[None, None, :] <= ids[None, :, None]).to(torch.float32)
        ext = causal[:, None, :, :] * m[:, None, None, :]
    else:
        ext = m[:, None, None, :]
    return (1.0 - ext) * -10000.0


def inverted_enc_mask(mask: Tensor) -> Tensor:
    """transformers-4.30.1 invert_attention_mask (call site xbert.py:1038-1043):
    (1 - m) * finfo(float32).min, shape [B,1,1,Lkv]."""
    m = mask.to(torch.float32)[:, None, None, :]
    return (1.0 - m) * torch.finfo(torch.float32).min


def attention(sd: SD, p: str, c: BertCfg, hidden: Tensor, add_mask: Tensor,
              enc: Optional[Tensor] = None, train=False) -> Tensor:
    """BertAttention.forward xbert.py:401-422 = BertSelfAttention :270-359 + BertSelfOutput :369-373."""
    B, L, H = hidden.shape
    nh, d = c.num_attention_heads, H // c.num_attention_heads
    kv_src = hidden if enc is None else enc
    q = _lin(sd, p + ".self.query", hidden).view(B, L, nh, d).permute(0, 2, 1, 3)
    k = _lin(sd, p + ".self.key", kv_src).view(B, -1, nh, d).permute(0, 2, 1, 3)
    v = _lin(sd, p + ".self.value", kv_src).view(B, -1, nh, d).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d) + add_mask
    if _BF16_STORAGE:
        # csrc/attention.hip: e = exp(s - max) in fp32, row sum of the UNROUNDED e, e rounded to bf16 as the MFMA operand,
        # context = (e_bf16 . V) / sum rounded to bf16
        e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        ctx = _st(torch.matmul(_st(e, "softmax_e"), v) / e.sum(dim=-1, keepdim=True), "attn_ctx").permute(0, 2, 1, 3).reshape(B, L, H)
    else:
        pr = _drop(torch.softmax(s, dim=-1), c.attention_probs_dropout_prob, train)
        ctx = torch.matmul(pr, v).permute(0, 2, 1, 3).reshape(B, L, H)
    out = _drop(_lin(sd, p + ".output.dense", ctx), c.hidden_dropout_prob, train)
    return _ln(sd, p + ".output.LayerNorm", out + hidden, c.layer_norm_eps)


def bert_layer(sd: SD, p: str, c: BertCfg, i: int, has_cross: bool, hidden, self_mask,
               enc=None, enc_mask=None, train=False) -> Tensor:
    """BertLayer.forward xbert.py:469-534 (self -> cross if layer_num >= fusion_layer -> FFN)."""
    lp = f"{p}encoder.layer.{i}."
    a = attention(sd, lp + "attention", c, hidden, self_mask, None, train)
    if has_cross and i >= c.fusion_layer:
        assert enc is not None
        a = attention(sd, lp + "crossattention", c, a, enc_mask, enc, train)
    h = _lin(sd, lp + "intermediate.dense", a, act=F.gelu)                   # :434-437 erf GELU
    o = _drop(_lin(sd, lp + "output.dense", h), c.hidden_dropout_prob, train)  # :447-451
    return _ln(sd, lp + "output.LayerNorm", o + a, c.layer_norm_eps)


def bert_model(sd: SD, p: str, c: BertCfg, has_cross: bool, *, input_ids=None, inputs_embeds=None,
               encoder_embeds=None, attention_mask=None, enc=None, enc_mask=None,
               is_decoder=False, mode="multi_modal", train=False) -> Tensor:
    """BertModel.forward xbert.py:950-1091 + BertEncoder.forward :543-644 (layer range by mode)."""
    if encoder_embeds is not None:
        x = encoder_embeds
    else:
        x = embeddings(sd, p, c, input_ids, inputs_embeds, train)
    B, L = x.shape[:2]
    if attention_mask is None:
        attention_mask = torch.ones(B, L)
    self_mask = extended_self_mask(attention_mask, is_decoder)
    emask = None
    if enc is not None:
        if enc_mask is None:
            enc_mask = torch.ones(enc.shape[:2])
        emask = inverted_enc_mask(enc_mask)
    lo, hi = {"text": (0, c.fusion_layer), "fusion": (c.fusion_layer, c.num_hidden_layers),
              "multi_modal": (0, c.num_hidden_layers)}[mode]
    for i in range(lo, hi):
        x = bert_layer(sd, p, c, i, has_cross, x, self_mask, enc, emask, train)
    return x


def mlm_head(sd: SD, p: str, c: BertCfg, x: Tensor) -> Tensor:
    """BertOnlyMLMHead xbert.py:662-706: decoder(LN(gelu(dense(x)))) + bias, decoder tied."""
    h = _lin(sd, p + "cls.predictions.transform.dense", x, act=F.gelu)
    h = _ln(sd, p + "cls.predictions.transform.LayerNorm", h, c.layer_norm_eps)
    dw = sd[p + "cls.predictions.decoder.weight"]
    if _BF16_WEIGHTS is None or _BF16_WEIGHTS.search(p + "cls.predictions.decoder"):
        dw = _st(dw, "weights")
    return F.linear(h, dw, sd[p + "cls.predictions.bias"])


# ------------------------------------------------------------------ SPMM_models.py
@torch.no_grad()
def momentum_update(sd: SD, cfg: SPMMCfg) -> None:
    """_momentum_update SPMM_models.py:266-269 (tied tensors updated once, as parameters() yields them once)."""
    seen = set()
    for name in list(sd):
        twin = momentum_twin(name)
        if twin is None or is_buffer(name) or sd[twin].data_ptr() in seen:
            continue
        seen.add(sd[twin].data_ptr())
        sd[twin].mul_(cfg.momentum).add_(sd[name].detach() * (1.0 - cfg.momentum))


@torch.no_grad()
def dequeue_and_enqueue(sd: SD, cfg: SPMMCfg, prop_feat_all: Tensor, text_feat_all: Tensor) -> None:
    """_dequeue_and_enqueue SPMM_models.py:272-286 (inputs are the all-gathered features)."""
    bs = prop_feat_all.shape[0]
    ptr = int(sd["queue_ptr"])
    assert cfg.queue_size % bs == 0
    sd["prop_queue"][:, ptr:ptr + bs] = prop_feat_all.T
    sd["text_queue"][:, ptr:ptr + bs] = text_feat_all.T
    sd["queue_ptr"][0] = (ptr + bs) % cfg.queue_size


def sample_negatives(weights: Tensor, u: Tensor) -> Tensor:
    """Row-wise inverse-CDF multinomial (one draw per row) from uniform u[B] in [0,1):
    the on-device replacement for the 2B `torch.multinomial(...).item()` host syncs at
    SPMM_models.py:165-178.  Same distribution, different RNG stream (parity tests inject
    the reference's recorded indices instead)."""
    cdf = torch.cumsum(weights.double(), dim=1)
    tgt = u.double()[:, None] * cdf[:, -1:]
    idx = (cdf <= tgt).sum(dim=1)
    # never pick a zero-weight entry (the diagonal)
    idx = idx.clamp(max=weights.shape[1] - 1)
    bad = weights.gather(1, idx[:, None])[:, 0] <= 0
    if bad.any():
        idx = torch.where(bad, weights.argmax(dim=1), idx)
    return idx


def spmm_forward(sd: SD, cfg: SPMMCfg, property_original: Tensor, text_input_ids: Tensor,
                 text_attention_mask: Tensor, alpha: float = 0.0, *, mpm_mask: Optional[Tensor] = None,
                 neg_idx: Optional[Tuple[Tensor, Tensor]] = None, train: bool = False,
                 gather=None, aux: Optional[dict] = None):
    """SPMM.forward SPMM_models.py:79-256.  Mutates sd exactly as the reference mutates
    its module: temp clamp (:80-81), EMA of the `_m` entries (:99), queue + pointer (:208).

    mpm_mask : the bernoulli(0.5) draw of :85 ([B,53], 1 = masked); drawn here if None.
    neg_idx  : (prop_neg_idx[B], text_neg_idx[B]) = the multinomial draws of :166 / :174.
    gather   : callable(tensor)->tensor replacing concat_all_gather :390-399 (identity if None).
    aux      : optional dict that receives intermediates for parity checks.
    """
    tc, pc = cfg.text, cfg.prop
    B = property_original.shape[0]
    with torch.no_grad():
        sd["temp"].clamp_(0.01, 0.5)                                                     # :80-81
    temp = sd["temp"]
    # ---- PV embedding / masking :82-88
    feat = F.linear(property_original.unsqueeze(2), sd["property_embed.weight"], sd["property_embed.bias"])
    if mpm_mask is None:
        mpm_mask = torch.bernoulli(torch.ones_like(property_original) * 0.5)
    mm = mpm_mask.unsqueeze(2)
    masked = feat * (1 - mm) + sd["property_mask"].expand(B, feat.shape[1], -1) * mm
    properties = torch.cat([sd["property_cls"].expand(B, -1, -1), masked], dim=1)
    # ---- unimodal encoders :90-95
    prop_embeds = bert_model(sd, "property_encoder.", pc, False, inputs_embeds=properties, train=train)
    prop_atts = torch.ones(prop_embeds.shape[:2], dtype=torch.long)
    prop_feat = F.normalize(_lin(sd, "property_proj", prop_embeds[:, 0, :], f32_out=True), dim=-1)
    text_embeds = bert_model(sd, "text_encoder.bert.", tc, True, input_ids=text_input_ids,
                             attention_mask=text_attention_mask, mode="text", train=train)
    text_feat = F.normalize(_lin(sd, "text_proj", text_embeds[:, 0, :], f32_out=True), dim=-1)
    # ---- momentum branch :98-119
    with torch.no_grad():
        momentum_update(sd, cfg)
        prop_embeds_m = bert_model(sd, "property_encoder_m.", pc, False, inputs_embeds=properties, train=train)
        prop_feat_m = F.normalize(_lin(sd, "property_proj_m", prop_embeds_m[:, 0, :], f32_out=True), dim=-1)
        prop_feat_all = torch.cat([prop_feat_m.t(), sd["prop_queue"].clone()], dim=1)
        text_embeds_m = bert_model(sd, "text_encoder_m.bert.", tc, True, input_ids=text_input_ids,
                                   attention_mask=text_attention_mask, mode="text", train=train)
        text_feat_m = F.normalize(_lin(sd, "text_proj_m", text_embeds_m[:, 0, :], f32_out=True), dim=-1)
        text_feat_all = torch.cat([text_feat_m.t(), sd["text_queue"].clone()], dim=1)
        sim_i2t_m = prop_feat_m @ text_feat_all / temp
        sim_t2i_m = text_feat_m @ prop_feat_all / temp
        sim_i2i_m = prop_feat_m @ prop_feat_all / temp
        sim_t2t_m = text_feat_m @ text_feat_all / temp
        tgt = torch.zeros_like(sim_i2t_m)
        tgt.fill_diagonal_(1)
        t_i2t = alpha * F.softmax(sim_i2t_m, dim=1) + (1 - alpha) * tgt
        t_t2i = alpha * F.softmax(sim_t2i_m, dim=1) + (1 - alpha) * tgt
        t_i2i = alpha * F.softmax(sim_i2i_m, dim=1) + (1 - alpha) * tgt
        t_t2t = alpha * F.softmax(sim_t2t_m, dim=1) + (1 - alpha) * tgt
    # ---- ITA :121-133
    sim_i2t = prop_feat @ text_feat_all / temp
    sim_t2i = text_feat @ prop_feat_all / temp
    sim_i2i = prop_feat @ prop_feat_all / temp
    sim_t2t = text_feat @ text_feat_all / temp
    l_i2t = -torch.sum(F.log_softmax(sim_i2t, dim=1) * t_i2t, dim=1).mean()
    l_t2i = -torch.sum(F.log_softmax(sim_t2i, dim=1) * t_t2i, dim=1).mean()
    l_i2i = -torch.sum(F.log_softmax(sim_i2i, dim=1) * t_i2i, dim=1).mean()
    l_t2t = -torch.sum(F.log_softmax(sim_t2t, dim=1) * t_t2t, dim=1).mean()
    loss_ita = (l_i2t + l_t2i + l_i2i + l_t2t) / 2
    if torch.isnan(sim_i2t).any() or torch.isnan(sim_t2i).any() or torch.isnan(loss_ita):
        z = torch.tensor(0.)
        return z, z.clone(), z.clone(), z.clone()
    # ---- ITM :137-206
    fus = dict(has_cross=True, mode="fusion", train=train)
    pos_pos_prop = bert_model(sd, "text_encoder.bert.", tc, encoder_embeds=prop_embeds, attention_mask=prop_atts,
                              enc=text_embeds, enc_mask=text_attention_mask, **fus)[:, 0, :]
    pos_pos_text = bert_model(sd, "text_encoder.bert.", tc, encoder_embeds=text_embeds,
                              attention_mask=text_attention_mask, enc=prop_embeds, enc_mask=prop_atts, **fus)[:, 0, :]
    pos_pos = torch.cat([pos_pos_prop, pos_pos_text], dim=-1)
    with torch.no_grad():
        w_i2t = F.softmax(sim_i2t[:, :B], dim=1)
        w_t2i = F.softmax(sim_t2i[:, :B], dim=1)
        w_i2t.fill_diagonal_(0)
        w_t2i.fill_diagonal_(0)
    if neg_idx is None:
        prop_neg_idx = torch.multinomial(w_t2i, 1)[:, 0]      # :165-167 (one draw per row)
        text_neg_idx = torch.multinomial(w_i2t, 1)[:, 0]      # :173-176
    else:
        prop_neg_idx, text_neg_idx = neg_idx
    prop_embeds_neg = prop_embeds[prop_neg_idx]
    text_embeds_neg = text_embeds[text_neg_idx]
    text_atts_neg = text_attention_mask[text_neg_idx]
    text_embeds_all = torch.cat([text_embeds, text_embeds_neg], dim=0)
    text_atts_all = torch.cat([text_attention_mask, text_atts_neg], dim=0)
    prop_embeds_all = torch.cat([prop_embeds_neg, prop_embeds], dim=0)
    prop_atts_all = torch.cat([prop_atts, prop_atts], dim=0)
    pos_neg_prop = bert_model(sd, "text_encoder.bert.", tc, encoder_embeds=prop_embeds_all, attention_mask=prop_atts_all,
                              enc=text_embeds_all, enc_mask=text_atts_all, **fus)[:, 0, :]
    pos_neg_text = bert_model(sd, "text_encoder.bert.", tc, encoder_embeds=text_embeds_all, attention_mask=text_atts_all,
                              enc=prop_embeds_all, enc_mask=prop_atts_all, **fus)[:, 0, :]
    pos_neg = torch.cat([pos_neg_prop, pos_neg_text], dim=-1)
    vl = torch.cat([pos_pos, pos_neg], dim=0)
    vl_output = _lin(sd, "itm_head", vl, f32_out=True, f32_w=True)
    itm_labels = torch.cat([torch.ones(B, dtype=torch.long), torch.zeros(2 * B, dtype=torch.long)])
    loss_itm = F.cross_entropy(vl_output, itm_labels)
    # ---- queue :208
    g = gather if gather is not None else (lambda t: t)
    dequeue_and_enqueue(sd, cfg, g(prop_feat_m), g(text_feat_m))
    # ---- "MLM" = next-token LM with momentum distillation :211-238
    labels = text_input_ids[:, 1:]
    with torch.no_grad():
        hid_m = bert_model(sd, "text_encoder_m.bert.", tc, True, input_ids=text_input_ids,
                           attention_mask=text_attention_mask, enc=prop_embeds_m, enc_mask=prop_atts,
                           is_decoder=True, train=train)
        logits_m = mlm_head(sd, "text_encoder_m.", tc, hid_m)[:, :-1, :]
    hid = bert_model(sd, "text_encoder.bert.", tc, True, input_ids=text_input_ids,
                     attention_mask=text_attention_mask, enc=prop_embeds, enc_mask=prop_atts,
                     is_decoder=True, train=train)
    mlm_output = mlm_head(sd, "text_encoder.", tc, hid)[:, :-1, :]
    loss_mlm = F.cross_entropy(mlm_output.permute(0, 2, 1), labels, ignore_index=-100)
    distill = -torch.sum(F.log_softmax(mlm_output, dim=-1) * F.softmax(logits_m, dim=-1), dim=-1)
    distill = distill[labels != 0].mean()
    loss_mlm = (1 - alpha) * loss_mlm + alpha * distill
    # ---- MPM :241-254
    prop_embeds_causal = bert_model(sd, "property_encoder.", pc, False, inputs_embeds=properties,
                                    is_decoder=True, train=train)
    prop_output = bert_model(sd, "text_encoder.bert.", tc, encoder_embeds=prop_embeds_causal,
                             attention_mask=prop_atts, enc=text_embeds, enc_mask=text_attention_mask,
                             is_decoder=True, **fus)[:, :-1, :]
    h = _lin(sd, "property_mtr_head.0", prop_output, act=F.gelu)
    h = _st(F.layer_norm(h, (h.shape[-1],), sd["property_mtr_head.2.weight"], sd["property_mtr_head.2.bias"],
                         tc.layer_norm_eps))
    pred = _lin(sd, "property_mtr_head.3", h, f32_out=True, f32_w=True).squeeze(-1)
    keep = (1 - mpm_mask).to(torch.bool)
    loss_mpm = F.mse_loss(pred[keep], property_original[keep])
    if aux is not None:
        aux.update(dict(prop_embeds=prop_embeds, text_embeds=text_embeds, prop_feat=prop_feat, text_feat=text_feat,
                        prop_feat_m=prop_feat_m, text_feat_m=text_feat_m, sim_i2t=sim_i2t, sim_t2i=sim_t2i,
                        sim_i2t_m=sim_i2t_m, vl_output=vl_output, mlm_output=mlm_output, logits_m=logits_m,
                        pred=pred, mpm_mask=mpm_mask, prop_neg_idx=prop_neg_idx, text_neg_idx=text_neg_idx,
                        loss_ita_parts=torch.stack([l_i2t, l_t2i, l_i2i, l_t2t]),
                        pos_pos=pos_pos, pos_neg=pos_neg, prop_embeds_causal=prop_embeds_causal,
                        prop_embeds_m=prop_embeds_m, text_embeds_m=text_embeds_m, properties=properties))
    return loss_mlm, loss_mpm * 5, loss_ita, loss_itm


# -------------------------------------------------------------- training_step wrapper
def cosine_lr(t: int, sched: dict) -> float:
    """scheduler/cosine_lr.py:69-96 with the arguments scheduler_factory.py:27-42 passes for
    SPMM_pretrain.py:62-63 (t_mul 1, decay_rate 1, cycle_limit 1, warmup_prefix True)."""
    warm, base = sched["warmup_epochs"], sched["lr"]
    if t < warm:
        return sched["warmup_lr"] + t * (base - sched["warmup_lr"]) / warm
    t = t - warm
    t_i = sched["epochs"]
    i = t // t_i
    if i < 1:
        return sched["min_lr"] + 0.5 * (base - sched["min_lr"]) * (1 + math.cos(math.pi * (t - t_i * i) / t_i))
    return sched["min_lr"]


def alpha_at(cfg_alpha: float, epoch: int, batch_idx: int, loader_len: int) -> float:
    """SPMM_models.py:355."""
    return cfg_alpha if epoch > 0 else cfg_alpha * min(1., batch_idx / loader_len)


class OracleTrainer:
    """SPMM.training_step SPMM_models.py:348-380 + configure_optimizers :338-343 on a state_dict:
    zero_grad, forward, sum of 4 losses, backward, clip_grad_norm_(5.), AdamW(lr, wd on ALL params),
    scheduler cadence (every 100 batches during warm-up in epoch 0 / once per epoch afterwards)."""

    def __init__(self, sd: SD, cfg: SPMMCfg, sched: dict, opt: dict, loader_len: int):
        self.sd, self.cfg, self.sched, self.loader_len = sd, cfg, sched, loader_len
        self.names = trainable_names(cfg)
        for n in self.names:
            sd[n].requires_grad_(True)
        _finish_tied(sd)
        self.params = [sd[n] for n in self.names]
        self.opt = torch.optim.AdamW(self.params, lr=opt["lr"], weight_decay=opt["weight_decay"])
        self.lr0 = cosine_lr(0, sched)           # Scheduler.__init__ -> update_groups(warmup_lr_init)
        for gp in self.opt.param_groups:
            gp["lr"] = self.lr0
        self.grad_norm = None

    def step(self, prop, ids, mask, epoch: int, batch_idx: int, **fw):
        self.opt.zero_grad()
        alpha = alpha_at(self.cfg.alpha, epoch, batch_idx, self.loader_len)
        losses = spmm_forward(self.sd, self.cfg, prop, ids, mask, alpha, **fw)
        loss = sum(losses)
        if float(loss) != 0.0:
            loss.backward()
            self.grad_norm = torch.nn.utils.clip_grad_norm_(self.params, 5.)
            self.opt.step()
        step_size, warm = 100, self.sched["warmup_epochs"]
        if epoch > 0 and batch_idx == 0:
            self._set_lr(cosine_lr(epoch + warm, self.sched))
        elif epoch == 0 and batch_idx % step_size == 0 and batch_idx <= warm * step_size:
            self._set_lr(cosine_lr(batch_idx // step_size, self.sched))
        return [float(x) for x in losses]

    def _set_lr(self, lr):
        for gp in self.opt.param_groups:
            gp["lr"] = lr


def _finish_tied(sd: SD) -> None:
    sd["text_encoder.cls.predictions.decoder.weight"] = sd["text_encoder.bert.embeddings.word_embeddings.weight"]
    sd["text_encoder.cls.predictions.decoder.bias"] = sd["text_encoder.cls.predictions.bias"]


# ------------------------------------------------------------------- synthetic data
def synthetic_batch(B: int, Lt: int, seed: int = 42, n_props: int = 53, vocab: int = 300):
    """SURVEY.md section 8d recipe: PV ~ N(0,1); ids = [2] + U{4..V-1} + [3], zero padded,
    len ~ U{Lt/2..Lt} with row 0 full length; mask = ids != 0."""
    g = torch.Generator().manual_seed(seed)
    prop = torch.randn(B, n_props, generator=g)
    ids = torch.zeros(B, Lt, dtype=torch.long)
    lens = torch.randint(max(Lt // 2, 3), Lt + 1, (B,), generator=g)
    lens[0] = Lt
    for b in range(B):
        n = int(lens[b])
        ids[b, 0] = 2
        ids[b, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g)
        ids[b, n - 1] = 3
    mask = (ids != 0).long()
    return prop, ids, mask


# ------------------------------------------------------------------ module-API view (for the decode parity tests)
class OracleModule:
    """The reference's module API over the functional oracle, CPU fp32: lets code written against
    `model.property_embed / property_cls / property_encoder / text_encoder(..., return_logits=True)` (d_pv2smiles_*.py) run
    on the oracle."""

    def __init__(self, sd: SD, cfg: SPMMCfg):
        self.sd, self.cfg = sd, cfg
        self.property_cls = sd["property_cls"]
        self.property_mask = sd["property_mask"]

    def property_embed(self, x):
        return F.linear(x, self.sd["property_embed.weight"], self.sd["property_embed.bias"])

    def property_encoder(self, inputs_embeds=None, return_dict=True, is_decoder=False):
        from types import SimpleNamespace
        with torch.no_grad():
            return SimpleNamespace(last_hidden_state=bert_model(self.sd, "property_encoder.", self.cfg.prop, False,
                                                                inputs_embeds=inputs_embeds, is_decoder=is_decoder))

    @property
    def text_encoder(self):
        return _OracleMaskedLM(self.sd, self.cfg)

    def property_mtr_head(self, x):
        """SPMM_models.py:39-42: Linear -> GELU -> LayerNorm -> Linear(H, 1)."""
        h = F.gelu(_lin(self.sd, "property_mtr_head.0", x))
        h = _ln(self.sd, "property_mtr_head.2", h, self.cfg.text.layer_norm_eps)
        return _lin(self.sd, "property_mtr_head.3", h)


class _OracleMaskedLM:
    """`model.text_encoder(...)` (BertForMaskedLM with return_logits, xbert.py:1377-1428) and `model.text_encoder.bert(...)`."""

    def __init__(self, sd: SD, cfg: SPMMCfg):
        self.sd, self.cfg = sd, cfg

    def bert(self, input_ids=None, attention_mask=None, encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
             return_dict=True, is_decoder=False, mode="multi_modal"):
        from types import SimpleNamespace
        with torch.no_grad():
            enc, em = encoder_hidden_states, encoder_attention_mask
            n = (input_ids if input_ids is not None else encoder_embeds).shape[0]
            if enc is not None:
                enc = enc.expand(n, -1, -1)
                em = None if em is None else em.expand(n, -1)
            h = bert_model(self.sd, "text_encoder.bert.", self.cfg.text, True, input_ids=input_ids, encoder_embeds=encoder_embeds,
                           attention_mask=attention_mask, enc=enc, enc_mask=em, is_decoder=is_decoder, mode=mode)
            return SimpleNamespace(last_hidden_state=h)

    def __call__(self, input_ids, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                 return_dict=True, is_decoder=False, return_logits=False):
        h = self.bert(input_ids, attention_mask=attention_mask, encoder_hidden_states=encoder_hidden_states,
                      encoder_attention_mask=encoder_attention_mask, is_decoder=is_decoder).last_hidden_state
        with torch.no_grad():
            return mlm_head(self.sd, "text_encoder.", self.cfg.text, h)
