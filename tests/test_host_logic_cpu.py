"""Host-side logic on CPU: the engine's launch schedule is run in DRY-RUN mode (every C-ABI call is checked against the
prototype parsed from include/spmm_hip.h; nothing is launched), plus parameter-arena / state_dict / schedule checks."""
import math

import pytest
import torch

import spmm_oracle as O


@pytest.fixture()
def dry():
    from spmm_amd import ops
    ops._DRY_RUN = True
    ops._dry_log.clear()
    yield ops
    ops._DRY_RUN = False


def _tiny_model(train_cfg=None):
    from spmm_amd.config import tiny_config
    from spmm_amd.model import SPMM
    return SPMM(config=train_cfg, spmm_config=tiny_config(), loader_len=4)


def test_state_dict_keys_shapes_and_order_match_reference_layout(dry):
    m = _tiny_model()
    sd = m.state_dict()
    spec = O.state_spec(O.tiny_cfg())
    assert list(sd.keys()) == [n for n, _, _ in spec]
    for n, shape, _ in spec:
        assert tuple(sd[n].shape) == tuple(shape), n
    # tied tensors alias their sources (xbert.py:691, :1362-1368)
    assert sd["text_encoder.cls.predictions.decoder.weight"].data_ptr() == sd["text_encoder.bert.embeddings.word_embeddings.weight"].data_ptr()
    assert sd["text_encoder_m.cls.predictions.decoder.bias"].data_ptr() == sd["text_encoder_m.cls.predictions.bias"].data_ptr()
    # trainable set = the reference's (momentum twins frozen, buffers excluded)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    assert sorted(names) == sorted(O.trainable_names(O.tiny_cfg()))


def test_full_size_arena_counts(dry):
    from spmm_amd.config import SPMMConfig, state_spec
    from spmm_amd.params import _layout_order
    cfg = SPMMConfig()
    spec = state_spec(cfg)
    assert len(spec) == 758
    order = _layout_order(spec)
    shapes = {n: s for n, s, _ in spec}
    assert sum(math.prod(shapes[n]) if shapes[n] else 1 for n in order) == 144_374_064


def test_load_state_dict_roundtrip_and_fused_views(dry):
    m = _tiny_model()
    sd = O.closed_form_state_dict(O.tiny_cfg())
    m.load_state_dict(sd)
    out = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(out[k].cpu(), v), k
    P = m.store
    pfx = "text_encoder.bert.encoder.layer.1.crossattention"
    kv = P.fused(pfx + ".self.", ("key", "value"), "weight", what="w")
    assert torch.equal(kv, torch.cat([sd[pfx + ".self.key.weight"], sd[pfx + ".self.value.weight"]]))
    qkvb = P.fused("property_encoder.encoder.layer.0.attention.self.", ("query", "key", "value"), "bias", what="w")
    assert torch.equal(qkvb, torch.cat([sd[f"property_encoder.encoder.layer.0.attention.self.{n}.bias"] for n in ("query", "key", "value")]))
    with pytest.raises(KeyError):
        m.load_state_dict({"bogus": torch.zeros(1)})


def test_step_schedule_dry_run(dry):
    """forward + backward + optimiser launch sequence with argument validation against the header."""
    sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20,
             'cooldown_epochs': 0}
    tc = {'embed_dim': 64, 'temp': 0.07, 'queue_size': 16, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched,
          'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
    m = _tiny_model(tc).train()
    prop, ids, mask = O.synthetic_batch(4, 16, seed=7)
    losses = m.training_step((prop, (ids, mask)), batch_idx=0)
    assert losses.shape == (4,)
    log = list(dry._dry_log)
    for needed in ("spmm_gemm_nt", "spmm_attn_fwd", "spmm_attn_bwd", "spmm_ln_fwd", "spmm_ln_bwd", "spmm_embed_ln_fwd", "spmm_embed_bwd",
                   "spmm_ita_rows", "spmm_sample_neg", "spmm_lm_loss", "spmm_itm_head", "spmm_mpm_head", "spmm_enqueue",
                   "spmm_ema_update", "spmm_grad_sqnorm", "spmm_adamw_step", "spmm_clamp_scalar", "spmm_l2norm_fwd", "spmm_l2norm_bwd"):
        assert needed in log, needed
    # 2 text layers (1 fusion) + 1 PV layer, packed text passes: S1 one group; S2 two (packed P2 | dense causal P10a);
    # S6 = its top layer alone, five groups (position-0 rows of the PV / of the packed-text / of the text-negative ITM sequences | LM pass |
    # causal PV pass), each with a self- and a cross-attention launch
    assert log.count("spmm_attn_bwd") == 1 + 2 + 5 * 2
    assert log.count("spmm_fusion_plan") == 1 and log.count("spmm_pack_plan") == 1
    assert log.count("spmm_segment_sum_bf16") == 2              # one fold per shared key/value source (text, PV) per fusion layer
    # autograd-boundary path
    dry._dry_log.clear()
    out = m(prop, ids, mask, alpha=0.1)
    sum(out).backward()
    assert "spmm_attn_bwd" in dry._dry_log
    m.eval()
    with torch.no_grad():
        out = m(prop, ids, mask, alpha=0.1)
    assert len(out) == 4


def test_cosine_schedule_matches_oracle_table():
    from spmm_amd.model import _CosineSchedule
    for sc in ({'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'warmup_lr': 5e-5, 'warmup_epochs': 20},
               {'lr': 1e-3, 'epochs': 4, 'min_lr': 1e-5, 'warmup_lr': 1e-4, 'warmup_epochs': 2}):
        s = _CosineSchedule(sc)
        for t in range(60):
            assert abs(s.lr_at(t) - O.cosine_lr(t, sc)) < 1e-15


def test_bert_config_accepts_string_true(tmp_path):
    import json
    from spmm_amd.config import BertConfig
    p = tmp_path / "c.json"
    json.dump({"hidden_size": 768, "num_attention_heads": 12, "add_cross_attention": "True", "fusion_layer": 6}, open(p, "w"))
    assert BertConfig.from_json_file(str(p)).add_cross_attention is True
    json.dump({"hidden_size": 100, "num_attention_heads": 3}, open(p, "w"))
    with pytest.raises(ValueError):
        BertConfig.from_json_file(str(p))


def test_checkpoint_roundtrip_and_legacy_layouts(dry, tmp_path):
    """Lightning-style checkpoint dict ('state_dict'), the legacy 'model' key and the legacy `_unk` names
    (SPMM_pretrain.py:24-37, SPMM_models_rxn.py:19-21, d_regression.py:157-161)."""
    m = _tiny_model()
    sd = O.closed_form_state_dict(O.tiny_cfg())
    m.load_state_dict(sd)
    path = str(tmp_path / "checkpoint_epoch=0.ckpt")
    m.save_checkpoint(path)
    ck = torch.load(path)
    assert set(ck) >= {"state_dict", "epoch"} and len(ck["state_dict"]) == 178
    m2 = _tiny_model()
    m2.load_checkpoint(path)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    legacy = {k.replace("property_mask", "property_unk"): v for k, v in sd.items()}
    m3 = _tiny_model()
    res = m3.load_checkpoint({"model": legacy})
    assert not res.missing_keys and torch.equal(m3.state_dict()["property_mask"], sd["property_mask"])
    # consumers drop queue / PV word-embedding keys before loading (d_pv2smiles_batched.py:138-142): strict=False tolerates it
    partial = {k: v for k, v in sd.items() if "queue" not in k and "property_encoder.embeddings.word_embeddings" not in k}
    res = _tiny_model().load_checkpoint({"state_dict": partial})
    assert set(res.missing_keys) == {"prop_queue", "text_queue", "queue_ptr", "property_encoder.embeddings.word_embeddings.weight"}


class _HashLM:
    """A stand-in exposing the reference's module API whose next-token logits are a deterministic pseudo-random function of
    (molecule, prefix): beams diverge, [SEP] turns up at random depths, and rows do not depend on batch composition."""
    V = 40

    class _Enc:
        def __call__(self, inputs_embeds=None, return_dict=True, **kw):
            from types import SimpleNamespace
            return SimpleNamespace(last_hidden_state=inputs_embeds)

    def __init__(self):
        import torch
        self.property_cls = torch.zeros(1, 1, 1)
        self.property_encoder = self._Enc()

    def property_embed(self, x):
        return x                                               # [B,53,1]: the "embedding" is the PV itself

    def text_encoder(self, text, attention_mask=None, encoder_hidden_states=None, **kw):
        import torch, zlib
        out = torch.empty(text.shape[0], text.shape[1], self.V)
        for r in range(text.shape[0]):
            key = encoder_hidden_states[min(r, encoder_hidden_states.shape[0] - 1), 1:4, 0].numpy().tobytes() + bytes(text[r].tolist())
            g = torch.Generator().manual_seed(zlib.crc32(key))
            out[r, -1] = torch.randn(self.V, generator=g) * 2.0
            out[r, -1, 3] += 1.0
        return out


def test_batched_beam_bookkeeping_matches_sequential_search():
    """BeamBook (tensorised, N molecules at once) makes the decisions of the per-molecule loop that restates
    d_pv2smiles_batched.py:29-57 -- same hypotheses, same order, same scores -- including molecules that stop early, ones
    that never finish and ones with several [SEP] candidates in one step."""
    import torch
    from spmm_amd.decode import beam_search_batched
    from decode_oracle import beam_search                 # oracle/: the reference's sequential one-molecule search
    m = _HashLM()
    props = torch.randn(12, 53, generator=torch.Generator().manual_seed(1))
    for k, steps in ((3, 10), (5, 6), (2, 1)):
        got = beam_search_batched(m, props, k=k, max_steps=steps, cached=False, sync_every=1)
        n_nonempty = 0
        for n in range(props.shape[0]):
            want = beam_search(m, props[n], k=k, max_steps=steps)
            assert len(want) == len(got[n]), (k, n)
            for (pw, sw), (pg, sg) in zip(want, got[n]):
                assert sw == sg and abs(pw - pg) < 1e-5, (k, n, sw, sg)
            n_nonempty += bool(want)
        assert n_nonempty >= (6 if steps > 1 else 0), (k, n_nonempty)


def test_wordpiece_tokenizer_matches_reference_golden(golden_dir):
    """SmilesWordPiece vs ids produced by the WordpieceTokenizer the reference wires into its BertTokenizer, on the
    reference's 300-piece vocabulary (fixture written by oracle/make_tokenizer_golden.py): drug-like SMILES, charged and
    stereo atoms, truncation at max_length=100, an out-of-vocabulary character, the empty string, whitespace."""
    import os
    import numpy as np
    import torch
    from spmm_amd.tokenizer import SmilesWordPiece
    g = np.load(os.path.join(golden_dir, "tokenizer_vocab300.npz"))
    tok = SmilesWordPiece([str(t) for t in g["vocab"]])
    texts = ["[CLS]" + str(s) for s in g["smiles"]]
    out = tok(texts, padding="longest", truncation=True, max_length=100, return_tensors="pt")
    assert torch.equal(out.input_ids, torch.from_numpy(g["input_ids"]))
    assert torch.equal(out.attention_mask, torch.from_numpy(g["attention_mask"]))
    assert out.input_ids.shape[1] == 100 and (out.input_ids[:, 0] == 2).all()          # the long row was truncated to 100
    # what the model consumes (SPMM_models.py:357) and the way back to text
    ids = out.input_ids[:, 1:]
    assert ids[0, 0].item() == tok.cls_token_id
    assert tok.decode(ids[0].tolist()) == str(g["smiles"][0])
    assert tok.decode(ids[4].tolist()) == "CCO"


def test_engine_options_from_env_and_overrides(monkeypatch):
    """EngineOptions.from_env: environment variables are read once, explicit overrides win, and every enumerated field rejects what it does not
    know -- incl. the legacy 0 / 1 and boolean spellings of `fused_xattn` (rounds 3-5) and the fields removed in round 6."""
    from spmm_amd.options import EngineOptions, _ENV
    for var, _ in _ENV.values():
        monkeypatch.delenv(var, raising=False)
    o = EngineOptions.from_env()
    assert o.fused_xattn == "nograd" and o.grad_wire == "fp32" and o.nt_under_comm == "auto" and o.pack_text and not o.resid_fp32
    assert not hasattr(o, "fp8") and not hasattr(o, "fuse_drop_res")
    for spelled, want in (("0", "off"), ("1", "all"), ("off", "off"), ("nograd", "nograd"), ("all", "all")):
        monkeypatch.setenv("SPMM_FUSED_XATTN", spelled)
        assert EngineOptions.from_env().fused_xattn == want
    assert EngineOptions.from_env(fused_xattn=True).fused_xattn == "all" and EngineOptions.from_env(fused_xattn=False).fused_xattn == "off"
    monkeypatch.setenv("SPMM_FUSED_XATTN", "sometimes")
    with pytest.raises(ValueError):
        EngineOptions.from_env()
    monkeypatch.delenv("SPMM_FUSED_XATTN")
    monkeypatch.setenv("SPMM_GRAD_WIRE", "fp16")
    with pytest.raises(ValueError):
        EngineOptions.from_env()
    monkeypatch.setenv("SPMM_GRAD_WIRE", "bf16")
    monkeypatch.setenv("SPMM_STREAMS", "1")
    o = EngineOptions.from_env(grad_overlap=False)
    assert o.grad_wire == "bf16" and not o.multi_stream and not o.grad_overlap
    assert o.replace(grad_wire="fp32").grad_wire == "fp32" and o.grad_wire == "bf16"
