"""Architecture / training configuration of the SPMM pretraining step.

Mirrors what the reference reads from config_bert.json / config_bert_property.json (through
BertConfig.from_json_file, SPMM_models.py:26,37) and from the inline dict of SPMM_pretrain.py:51-65.
No HF dependency: the JSON is parsed here and the string "True" of config_bert.json:22 is accepted."""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class BertConfig:
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
    pad_token_id: int = 0
    add_cross_attention: bool = False

    @classmethod
    def from_json_file(cls, path: str) -> "BertConfig":
        raw = json.load(open(path))
        kw = {}
        for k in cls.__dataclass_fields__:
            if k in raw:
                v = raw[k]
                if k == "add_cross_attention" and isinstance(v, str):
                    v = v.strip().lower() == "true"
                kw[k] = v
        c = cls(**kw)
        c.validate()
        return c

    def validate(self):
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be a multiple of num_attention_heads (xbert.py:228-232)")
        if self.hidden_size // self.num_attention_heads != 64:
            raise ValueError("the gfx950 attention kernels are built for head_dim 64 (config_bert.json: 768/12)")
        if self.encoder_width != self.hidden_size:
            raise ValueError("encoder_width != hidden_size is not supported (reference uses 768/768)")
        if self.hidden_size % 64 or self.intermediate_size % 64:
            raise ValueError("hidden/intermediate sizes must be multiples of 64")


@dataclass
class SPMMConfig:
    text: BertConfig = field(default_factory=lambda: BertConfig(add_cross_attention=True))
    prop: BertConfig = field(default_factory=lambda: BertConfig(num_hidden_layers=6, vocab_size=1))
    embed_dim: int = 256
    temp: float = 0.07
    queue_size: int = 36864
    momentum: float = 0.995
    alpha: float = 0.4
    n_props: int = 53

    @classmethod
    def from_reference_dict(cls, config: dict) -> "SPMMConfig":
        """`config` as built in SPMM_pretrain.py:51-65 (keys embed_dim, bert_config_text, bert_config_property, and --
        when training -- temp, queue_size, momentum, alpha)."""
        t = BertConfig.from_json_file(config["bert_config_text"])
        p = BertConfig.from_json_file(config["bert_config_property"])
        return cls(text=t, prop=p, embed_dim=config["embed_dim"], temp=config.get("temp", 0.07),
                   queue_size=config.get("queue_size", 36864), momentum=config.get("momentum", 0.995),
                   alpha=config.get("alpha", 0.4))


def tiny_config() -> SPMMConfig:
    t = BertConfig(hidden_size=128, num_attention_heads=2, intermediate_size=512, num_hidden_layers=2, fusion_layer=1,
                   encoder_width=128, add_cross_attention=True)
    p = BertConfig(hidden_size=128, num_attention_heads=2, intermediate_size=512, num_hidden_layers=1, fusion_layer=1,
                   encoder_width=128, vocab_size=1)
    return SPMMConfig(text=t, prop=p, embed_dim=64, queue_size=16)


# ------------------------------------------------------------------------------------------------ state_dict spec
Spec = Tuple[str, tuple, str]   # (name, shape, kind)


def _bert_spec(prefix: str, c: BertConfig, with_cross: bool) -> List[Spec]:
    H, I = c.hidden_size, c.intermediate_size
    out = [(prefix + "embeddings.position_ids", (1, c.max_position_embeddings), "posid"),
           (prefix + "embeddings.word_embeddings.weight", (c.vocab_size, H), "emb"),
           (prefix + "embeddings.position_embeddings.weight", (c.max_position_embeddings, H), "emb"),
           (prefix + "embeddings.token_type_embeddings.weight", (c.type_vocab_size, H), "emb"),
           (prefix + "embeddings.LayerNorm.weight", (H,), "ln_w"), (prefix + "embeddings.LayerNorm.bias", (H,), "ln_b")]
    for i in range(c.num_hidden_layers):
        lp = f"{prefix}encoder.layer.{i}."
        blocks = ["attention"] + (["crossattention"] if with_cross and i >= c.fusion_layer else [])
        for blk in blocks:
            kin = c.encoder_width if blk == "crossattention" else H
            for nm, shp in (("query", (H, H)), ("key", (H, kin)), ("value", (H, kin))):
                out += [(f"{lp}{blk}.self.{nm}.weight", shp, "lin_w"), (f"{lp}{blk}.self.{nm}.bias", (H,), "lin_b")]
            out += [(f"{lp}{blk}.output.dense.weight", (H, H), "lin_w"), (f"{lp}{blk}.output.dense.bias", (H,), "lin_b"),
                    (f"{lp}{blk}.output.LayerNorm.weight", (H,), "ln_w"), (f"{lp}{blk}.output.LayerNorm.bias", (H,), "ln_b")]
        out += [(lp + "intermediate.dense.weight", (I, H), "lin_w"), (lp + "intermediate.dense.bias", (I,), "lin_b"),
                (lp + "output.dense.weight", (H, I), "lin_w"), (lp + "output.dense.bias", (H,), "lin_b"),
                (lp + "output.LayerNorm.weight", (H,), "ln_w"), (lp + "output.LayerNorm.bias", (H,), "ln_b")]
    return out


def _mlm_spec(prefix: str, c: BertConfig) -> List[Spec]:
    H, V = c.hidden_size, c.vocab_size
    return [(prefix + "cls.predictions.bias", (V,), "lin_b"),
            (prefix + "cls.predictions.transform.dense.weight", (H, H), "lin_w"),
            (prefix + "cls.predictions.transform.dense.bias", (H,), "lin_b"),
            (prefix + "cls.predictions.transform.LayerNorm.weight", (H,), "ln_w"),
            (prefix + "cls.predictions.transform.LayerNorm.bias", (H,), "ln_b"),
            (prefix + "cls.predictions.decoder.weight", (V, H), "tied_w"),
            (prefix + "cls.predictions.decoder.bias", (V,), "tied_b")]


def state_spec(cfg: SPMMConfig) -> List[Spec]:
    """Every state_dict entry of the reference SPMM module, in its order (SPMM_models.py:17-77).
    758 entries for the published configs."""
    H, E, Q = cfg.text.hidden_size, cfg.embed_dim, cfg.queue_size
    s: List[Spec] = [("property_cls", (1, 1, H), "zero"), ("property_mask", (1, 1, H), "zero"), ("temp", (), "temp"),
                     ("prop_queue", (E, Q), "queue"), ("text_queue", (E, Q), "queue"), ("queue_ptr", (1,), "ptr")]
    s += _bert_spec("text_encoder.bert.", cfg.text, True) + _mlm_spec("text_encoder.", cfg.text)
    s += [("property_proj.weight", (E, H), "lin_w"), ("property_proj.bias", (E,), "lin_b"),
          ("text_proj.weight", (E, H), "lin_w"), ("text_proj.bias", (E,), "lin_b"),
          ("itm_head.weight", (2, 2 * H), "lin_w"), ("itm_head.bias", (2,), "lin_b"),
          ("property_embed.weight", (H, 1), "lin_w"), ("property_embed.bias", (H,), "lin_b")]
    s += _bert_spec("property_encoder.", cfg.prop, False)
    s += [("property_mtr_head.0.weight", (H, H), "lin_w"), ("property_mtr_head.0.bias", (H,), "lin_b"),
          ("property_mtr_head.2.weight", (H,), "ln_w"), ("property_mtr_head.2.bias", (H,), "ln_b"),
          ("property_mtr_head.3.weight", (1, H), "lin_w"), ("property_mtr_head.3.bias", (1,), "lin_b")]
    s += _bert_spec("property_encoder_m.", cfg.prop, False)
    s += [("property_proj_m.weight", (E, H), "lin_w"), ("property_proj_m.bias", (E,), "lin_b")]
    s += _bert_spec("text_encoder_m.bert.", cfg.text, True) + _mlm_spec("text_encoder_m.", cfg.text)
    s += [("text_proj_m.weight", (E, H), "lin_w"), ("text_proj_m.bias", (E,), "lin_b")]
    return s


MOMENTUM_PAIRS = [("property_encoder.", "property_encoder_m."), ("property_proj.", "property_proj_m."),
                  ("text_encoder.", "text_encoder_m."), ("text_proj.", "text_proj_m.")]   # SPMM_models.py:56-60


def momentum_twin(name: str):
    for a, b in MOMENTUM_PAIRS:
        if name.startswith(a):
            return b + name[len(a):]
    return None


def student_of(name: str):
    for a, b in MOMENTUM_PAIRS:
        if name.startswith(b):
            return a + name[len(b):]
    return None


def is_buffer(name: str) -> bool:
    return name.endswith("position_ids") or name in ("prop_queue", "text_queue", "queue_ptr")
