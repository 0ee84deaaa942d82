"""DEV-ONLY: import the upstream reference (/root/reference) in THIS container.

TEST INFRASTRUCTURE -- never imported by the product (spmm_amd/).  This module only
exists so that `oracle/make_golden.py` can run the real reference once, here, and
write small golden vectors under tests/golden/.  /root/reference does not exist on
the GPU box and nothing under tests -m gpu / bench.py / smoke() imports this file.

The reference pins transformers==4.30.1 / torch==1.13.1 (requirements.txt:7-9); this
container has transformers 5.x, so five compatibility patches are applied (SURVEY.md
section 8c).  None of them changes arithmetic on the pretraining path.
"""
import json
import os
import sys
import tempfile
import types

import torch
import torch.nn as nn

REF = os.environ.get("SPMM_REFERENCE", "/root/reference")


def _install():
    if "SPMM_models" in sys.modules:
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree {REF} is not present (dev container only)")
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    # (1) symbols xbert.py:54-59 imports from transformers.modeling_utils
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(
            NotImplementedError("head pruning is off the pretraining path"))
    # (2) pytorch_lightning is not installed: LightningModule -> nn.Module
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):
            pass

        pl.LightningModule = LightningModule
        sys.modules["pytorch_lightning"] = pl
    sys.path.insert(0, REF)
    import xbert  # noqa: E402

    # (4) transformers-4.30.1 init_weights semantics (apply _init_weights, then tie
    #     cls.predictions.decoder.weight to bert.embeddings.word_embeddings.weight)
    def init_weights(self):
        self.apply(self._init_weights)
        if hasattr(self, "cls") and hasattr(self, "bert"):
            self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight

    xbert.BertPreTrainedModel.init_weights = init_weights
    # (5) get_head_mask is gone in transformers 5
    xbert.BertPreTrainedModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    if not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.distributed.init_process_group("gloo", rank=0, world_size=1)
    import SPMM_models  # noqa: F401,E402


def write_bert_config(path, **over):
    """(3) config_bert.json:22 has "add_cross_attention": "True" (a string) which
    transformers 5 rejects -- write a copy with booleans, optionally overriding sizes."""
    base = json.load(open(os.path.join(REF, "config_bert.json")))
    base["add_cross_attention"] = True
    base.update(over)
    json.dump(base, open(path, "w"))
    return path


def build_reference_spmm(text_over, prop_over, config):
    """Return the upstream SPMM module built from overridden BertConfig JSONs."""
    _install()
    import SPMM_models
    d = tempfile.mkdtemp()
    tcfg = write_bert_config(os.path.join(d, "t.json"), **text_over)
    pover = dict(prop_over)
    pover.setdefault("vocab_size", 1)
    pover.pop("add_cross_attention", None)
    pcfg = write_bert_config(os.path.join(d, "p.json"), **pover)
    cfg = dict(config)
    cfg["bert_config_text"] = tcfg
    cfg["bert_config_property"] = pcfg
    return SPMM_models.SPMM(config=cfg, tokenizer=None, loader_len=cfg.get("loader_len", 10))
