"""End-to-end parity of the HIP pretraining step (through spmm_amd.SPMM -> C ABI) against the CPU oracle and the
golden vectors produced by the real reference.  bf16 activations/weights with fp32 accumulation; tolerances are
stated next to every comparison."""
import os

import numpy as np
import pytest
import torch

import decode_oracle                      # oracle/decode_oracle.py (tests/conftest.py puts oracle/ on sys.path)

pytestmark = pytest.mark.gpu


# Per-loss absolute bounds |hip - reference| in the order (loss_mlm, 5*loss_mpm, loss_ita, loss_itm).  north_star asks for 1e-3:
# that is asserted wherever the bf16 pipeline meets it; elsewhere the bound is 1.5x the deviation measured on MI355X (DESIGN.md
# section 5 has the table and the reason: activations are stored in bf16, 8 mantissa bits, so a hidden state of magnitude 2-4
# carries 4e-3 .. 1.6e-2 of rounding per store whatever the accumulation precision).
LOSS_ATOL = {
    "tiny_golden":   [1e-3, 2.6e-3, 1.5e-3, 1.5e-3],    # measured 1.7e-4 / 1.7e-3 / 9.4e-4 / 9.7e-4  (vs the REAL reference)
    "tiny_golden_2": [1e-3, 5.4e-3, 4.2e-3, 1.5e-3],    # the second forward, from the state the first one left (queue, EMA): 1.7e-4 / 3.6e-3 / 2.8e-3 / 6.9e-4
    "h768_2layer":   [1e-3, 1.7e-2, 1.4e-3, 1.7e-3],    # measured 2.1e-4 / 1.1e-2 / 9.2e-4 / 1.1e-3
    "full_depth_b8": [1e-3, 4.9e-3, 6.2e-3, 3.8e-3],    # measured 6.4e-5 / 3.2e-3 / 4.1e-3 / 2.5e-3
    "bench_shape":   [1e-3, 1.4e-2, 5.5e-3, 2e-3],      # measured 6.4e-4 / 9.2e-3 / 3.6e-3 / 1.3e-3  (B=32, Lt=128, Q=36864, 12+6 layers)
    "bench_shape_r32": [1e-3, 1.4e-2, 1e-3, 1.7e-3],    # the same (packed rows) with the fp32 residual stream (EngineOptions.resid_fp32): measured 5.2e-4 / 9.1e-3 /
                                                        # 9.3e-4 / 1.13e-3 (bf16 stream, same run: 5.5e-4 / 1.0e-2 / 3.8e-3 / 1.8e-3): MLM and ITA meet north_star's 1e-3
    "edge_shapes":   [1e-3, 3e-3, 3.1e-3, 2.3e-3],      # worst over the five cases: 4.4e-4 / 1.9e-3 / 2.1e-3 / 1.5e-3
    "lt160":         [1e-3, 1.3e-3, 4.4e-3, 1e-3],      # measured 2.0e-4 / 8.6e-4 / 2.9e-3 / 1.2e-4
    "lt300":         [1e-3, 1.3e-3, 4.4e-3, 1e-3],      # (chunked attention path, > 256 tokens)
    "lt256_golden":  [1e-3, 2.8e-3, 2.9e-3, 1e-3],      # vs the REAL reference at Lt = 256, toy widths, closed-form weights: measured 1.9e-6 / 1.83e-3 / 1.91e-3 / 4.6e-4
    "lt256_h768":    [1e-3, 1e-3, 1e-3, 1e-3],          # measured 5.2e-5 / 3.8e-4 / 6.5e-4 / 6.2e-4: BASELINE configs[4]'s sequence length at the published widths, 2+2 layers, packed rows
    "wide_golden":   [1e-3, 4.4e-3, 2e-2, 7.7e-3],      # measured 5.4e-4 / 2.9e-3 / 1.33e-2 / 5.1e-3 (closed-form weights ~0.08: sims up to 40)
    "grad_tiny":     [1e-3, 2.9e-3, 1.3e-3, 1e-3],      # measured 2.5e-4 / 1.9e-3 / 8.2e-4 / 4.0e-4
}

# product vs `oracle.bf16_storage(backward=True)`: (whole-gradient relative L2, worst family relative L2, worst ranked tensor) -- 1.5 x the measured
# 2.1e-3 / 2.9e-3 / 6.3e-3 (h768_2layer) and 2.2e-3 / 2.6e-3 / 9.5e-3 (bench shape; profiles/r05_gradient_parity.txt); against the fp32 oracle the
# same gradients sit at 5.3e-3 / 6.1e-3.  See test_gradients_match_the_bf16_storage_model_of_the_oracle.
GRAD_VS_STORAGE_MODEL = {"h768_2layer": (3.5e-3, 4.5e-3, 1e-2), "bench_shape": (3.5e-3, 4.5e-3, 1.5e-2)}
# test_training_trace_vs_reference: per step (absolute loss tolerance, relative gradient-norm tolerance) for the bf16 / fp32 residual stream.
# Measured worst loss deviation per step 9.2e-4 / 3.5e-3 / 1.3e-2 (bf16) and 5.4e-4 / 5.8e-3 / 1.25e-2 (fp32 stream), gradient norm
# 2e-4 / 8e-4 / 2.2e-2 and 1e-4 / 9e-4 / 2.0e-2: the third step of this toy run (closed-form weights, lr 1e-3) amplifies whatever the first two left.
TRACE_TOL = {False: ([1.5e-3, 6e-3, 2e-2], [1e-3, 2e-3, 3.3e-2]), True: ([1.5e-3, 9e-3, 2e-2], [1e-3, 2e-3, 3.3e-2])}


def assert_losses(got, ref, key, what=""):
    got, ref, tol = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64), np.asarray(LOSS_ATOL[key])
    d = np.abs(got - ref)
    assert (d <= tol).all(), f"{what or key}: |hip - reference| = {d} exceeds {tol} (hip {got}, reference {ref})"


def _mk(SPMM, cfg, sd, train_cfg=None):
    m = SPMM(config=train_cfg, spmm_config=cfg)
    m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
    return m


from helpers_gpu import _cuda, _tiny_train_model      # noqa: E402  (tests/helpers_gpu.py)


def test_forward_matches_reference_golden(env, golden_dir):
    O, SPMM, tiny_config, *_ = env
    for name, B, Lt, seed in (("fwd_tiny_b4_l16.npz", 4, 16, 7), ("fwd_tiny_b8_l24.npz", 8, 24, 11)):
        g = np.load(os.path.join(golden_dir, name))
        sd = O.closed_form_state_dict(O.tiny_cfg())
        m = _mk(SPMM, tiny_config(), sd).eval()
        prop, ids, mask = O.synthetic_batch(B, Lt, seed=seed)
        aux = {}
        with torch.no_grad():
            losses = m(prop, ids, mask, alpha=float(g["alpha"]), mpm_mask=torch.from_numpy(g["mpm_mask"]).cuda(),
                       neg_idx=tuple(_cuda(torch.from_numpy(g["prop_neg_idx"]), torch.from_numpy(g["text_neg_idx"]))), aux=aux)
        got = np.array([float(x) for x in losses])
        print(name, "hip", got, "reference", g["losses"], "diff", np.abs(got - g["losses"]))
        assert_losses(got, g["losses"], "tiny_golden", name)
        # state mutations: queue block, pointer, EMA'd momentum weights, clamped temp
        sdo = m.state_dict()
        np.testing.assert_allclose(sdo["prop_queue"].cpu().numpy(), g["prop_queue"], atol=2e-2)
        np.testing.assert_allclose(sdo["text_queue"].cpu().numpy(), g["text_queue"], atol=2e-2)
        assert int(sdo["queue_ptr"]) == int(g["queue_ptr"][0])
        for k in g.files:
            if k.startswith("chk::"):
                t = sdo[k[5:]].double()
                np.testing.assert_allclose([t.sum().item(), t.abs().sum().item()], g[k], rtol=1e-5, atol=1e-5)
        # second forward from the mutated state
        prop2, ids2, mask2 = O.synthetic_batch(B, Lt, seed=seed + 1)
        with torch.no_grad():
            l2 = m(prop2, ids2, mask2, alpha=0.0, mpm_mask=torch.from_numpy(g["mpm_mask2"]).cuda(),
                   neg_idx=tuple(_cuda(torch.from_numpy(g["prop_neg_idx2"]), torch.from_numpy(g["text_neg_idx2"]))))
        assert_losses([float(x) for x in l2], g["losses2"], "tiny_golden_2", name + " second forward")
        assert int(m.queue_ptr) == int(g["queue_ptr2"][0])


def test_submodule_facades_match_reference_blocks(env, golden_dir):
    """The sub-module call signatures external callers use against activations captured from the REAL
    reference's sub-modules (oracle/make_golden.py blk_*): PV encoder (plain and causal), text encoder mode='text',
    fusion mode on encoder_embeds, and text_encoder(..., is_decoder=True, return_logits=True)."""
    O, SPMM, tiny_config, *_ = env
    g = np.load(os.path.join(golden_dir, "fwd_tiny_b4_l16.npz"))
    m = _mk(SPMM, tiny_config(), O.closed_form_state_dict(O.tiny_cfg())).eval()
    B, Lt = 4, 16
    ids, mask = torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])
    x = torch.from_numpy(g["blk_prop_in"])
    pe = m.property_encoder(inputs_embeds=x, return_dict=True).last_hidden_state
    pec = m.property_encoder(inputs_embeds=x, is_decoder=True, return_dict=True).last_hidden_state
    te = m.text_encoder.bert(ids, attention_mask=mask, return_dict=True, mode='text').last_hidden_state
    ones = torch.ones(B, 54, dtype=torch.long)
    fu = m.text_encoder.bert(encoder_embeds=pe, attention_mask=ones, encoder_hidden_states=te, encoder_attention_mask=mask,
                             return_dict=True, mode='fusion').last_hidden_state
    lg = m.text_encoder(ids, attention_mask=mask, encoder_hidden_states=pe, encoder_attention_mask=ones, return_dict=True,
                        is_decoder=True, return_logits=True)
    # stated tolerance (bf16 activations): max |diff| < 6e-2 and mean |diff| < 1.2e-2 on O(1) activations.  The fp32 oracle
    # with weights/activations rounded to bf16 at the same points shows the same figures against these vectors
    # (text block: max 0.042 / mean 0.0087, PV block: 0.018 / 0.0029), i.e. this is the bf16 floor, not kernel error.
    for got, key in ((pe, "blk_prop_enc"), (pec, "blk_prop_enc_causal"), (te, "blk_text_enc"), (fu, "blk_fusion"), (lg, "blk_logits")):
        ref = torch.from_numpy(g[key])
        d = (got.cpu() - ref).abs()
        print(f"  facade {key}: max|diff| {d.max().item():.4g} mean|diff| {d.mean().item():.4g} (ref max {ref.abs().max().item():.3g})")
        assert got.shape == ref.shape and d.max().item() < 6e-2 and d.mean().item() < 1.2e-2, key
    # small heads as callables (fp32): same arithmetic as nn.Linear / nn.Sequential
    sd = O.closed_form_state_dict(O.tiny_cfg())
    cls = pe[:, 0, :]
    ref = torch.nn.functional.linear(cls.cpu(), sd["property_proj.weight"], sd["property_proj.bias"])
    assert torch.allclose(m.property_proj(cls).cpu(), ref, atol=1e-5)
    F = torch.nn.functional
    pin = torch.from_numpy(g["prop"]).unsqueeze(2)
    pf = m.property_embed(pin)
    assert pf.shape == (B, 53, 128)
    assert torch.allclose(pf.cpu(), F.linear(pin, sd["property_embed.weight"], sd["property_embed.bias"]), atol=1e-6)
    pair = torch.cat([cls, te[:, 0, :]], dim=-1)                             # itm_head = Linear(2H, 2) on the two CLS states (SPMM_models.py:43, :201)
    assert torch.allclose(m.itm_head(pair).cpu(), F.linear(pair.cpu(), sd["itm_head.weight"], sd["itm_head.bias"]), atol=1e-5)
    mt = m.property_mtr_head(pe[:, :-1, :])
    assert mt.shape == (B, 53, 1)
    h = F.gelu(F.linear(pe[:, :-1, :].cpu(), sd["property_mtr_head.0.weight"], sd["property_mtr_head.0.bias"]))
    h = F.layer_norm(h, (128,), sd["property_mtr_head.2.weight"], sd["property_mtr_head.2.bias"], 1e-12)
    ref = F.linear(h, sd["property_mtr_head.3.weight"], sd["property_mtr_head.3.bias"])
    assert (mt.cpu() - ref).abs().max().item() < 3e-2, (mt.cpu() - ref).abs().max().item()       # bf16 GEMM + LayerNorm rows, as in the training step
    assert m.property_cls.shape == (1, 1, 128) and m.device.type == "cuda"


def _mid_cfg(env, layers=(2, 1, 2), Q=64):
    O, SPMM, tiny_config, SPMMConfig, BertConfig = env
    nt, f, npv = layers
    t = BertConfig(num_hidden_layers=nt, fusion_layer=f, add_cross_attention=True)
    p = BertConfig(num_hidden_layers=npv, fusion_layer=f, vocab_size=1)
    cfg = SPMMConfig(text=t, prop=p, embed_dim=256, queue_size=Q)
    ocfg = O.full_cfg()
    ocfg.text.num_hidden_layers, ocfg.text.fusion_layer = nt, f
    ocfg.prop.num_hidden_layers, ocfg.prop.fusion_layer = npv, f
    ocfg.queue_size = Q
    return cfg, ocfg


def test_forward_matches_oracle_h768(env):
    """H=768 / 12 heads (the real widths), 2 text layers (1 fusion) + 2 PV layers, random-init weights, B=8, Lt=40."""
    O, SPMM, *_ = env
    cfg, ocfg = _mid_cfg(env)
    sd = O.init_state_dict(ocfg, seed=3)
    m = _mk(SPMM, cfg, sd).eval()
    B, Lt = 8, 40
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=5)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(1))
    neg = (torch.arange(B).roll(3), torch.arange(B).roll(5))
    aux, oaux = {}, {}
    with torch.no_grad():
        losses = m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)), aux=aux)
        ref = O.spmm_forward(sd, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, aux=oaux)
    got, ref = np.array([float(x) for x in losses]), np.array([float(x) for x in ref])
    print("hip", got, "oracle", ref, "diff", np.abs(got - ref))
    assert_losses(got, ref, "h768_2layer")
    Lp = 54
    for key, shape, tol in (("prop_embeds", (B, Lp, 768), 6e-2), ("text_embeds", (B, Lt, 768), 6e-2),
                            ("prop_feat", (B, 256), 4e-3), ("text_feat", (B, 256), 4e-3), ("prop_feat_m", (B, 256), 4e-3),
                            ("sim_i2t", None, 8e-2), ("sim_t2i", None, 8e-2), ("vl_output", (3 * B, 2), 2e-2),
                            ("mlm_output", None, 5e-2), ("logits_m", None, 5e-2), ("pred", (B, 53), 3e-2)):
        a = aux[key].float().cpu().reshape(oaux[key].shape)
        err = (a - oaux[key]).abs().max().item()
        print(f"  {key}: max|diff| {err:.4g} (ref max {oaux[key].abs().max().item():.3g})")
        assert err < tol, key


def test_fused_cross_attention_self_check_falls_back_to_the_composite(env, monkeypatch):
    """The one-launch cross-attention block is in the DEFAULT path (the passes that keep no tape) and is checked once per process against
    the composite of launches on the first block it serves.  A kernel that disagrees must not end the run: the engine warns, stops using it
    for the process and the forward's results are the composite's.  Here the kernel is made to disagree (its output zeroed)."""
    O, SPMM, *_ = env
    from spmm_amd import ops
    from spmm_amd.options import EngineOptions
    cfg, ocfg = _mid_cfg(env)
    sd = O.init_state_dict(ocfg, seed=3)
    B, Lt = 8, 48
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=21)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(1))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
    out, calls = {}, []
    orig = ops.xattn_fwd

    def broken(Q, K, V, WoF, bo, R, gamma, beta, Y, **kw):
        calls.append(1)
        r = orig(Q, K, V, WoF, bo, R, gamma, beta, Y, **kw)
        Y.zero_()
        return r

    for mode in ("off", "broken"):
        m = SPMM(config=None, spmm_config=cfg, options=EngineOptions.from_env(fused_xattn="off" if mode == "off" else "nograd"))
        m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
        m.eval()
        if mode == "broken":
            monkeypatch.setattr(ops, "xattn_fwd", broken)
        with torch.no_grad():
            if mode == "broken":
                with pytest.warns(UserWarning, match="falling back to the composite"):
                    out[mode] = np.array([float(x) for x in m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))])
                n_first = len(calls)
                m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))          # a second forward: the kernel stays unused
                assert len(calls) == n_first == 1 and m.engine._xattn_off
            else:
                out[mode] = np.array([float(x) for x in m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))])
    print("composite", out["off"], "after the fallback", out["broken"])
    np.testing.assert_allclose(out["broken"], out["off"], rtol=1e-5, atol=1e-6)       # (eval mode: no dropout, the composite is deterministic but for atomic sums)


def test_fused_cross_attention_inside_the_step(env):
    """EngineOptions.fused_xattn: the cross-attention blocks of every fusion layer as ONE launch each (csrc/xattn.hip) inside the real
    step -- H=768, 2+2 layers, packed text rows, shared K/V sources, train mode with dropout.  Both forms draw the same dropout
    masks, so losses and the whole gradient must agree with the composite's to rounding (the fused kernel keeps the dense output
    in fp32 where the composite rounds it to bf16 before the LayerNorm), and the losses match the oracle within the usual budget
    with dropout off."""
    O, SPMM, *_ = env
    from spmm_amd.options import EngineOptions
    cfg, ocfg = _mid_cfg(env)
    sd = O.init_state_dict(ocfg, seed=3)
    B, Lt = 8, 48
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=21)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(1))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
    sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20, 'cooldown_epochs': 0}
    tc = {'embed_dim': 256, 'temp': 0.07, 'queue_size': 64, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched,
          'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
    res = {}
    for fused in (False, True):
        m = SPMM(config=tc, spmm_config=cfg, options=EngineOptions.from_env(fused_xattn=fused))
        m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
        m.train()
        m.engine.seed.fill_(777)
        l = [float(x) for x in m.fused_step(*_cuda(prop, ids, mask), 0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))]
        res[fused] = (np.array(l), m.store.grad.clone())
        if fused:               # (state BEFORE the eval forward: the forward itself moves the queue and the momentum weights)
            sd2 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            m.eval()
            with torch.no_grad():
                got = np.array([float(x) for x in m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))])
    print("composite", res[False][0], "fused", res[True][0])
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=2e-3, atol=2e-3)
    g0, g1 = res[False][1], res[True][1]
    rel = ((g1 - g0).norm() / g0.norm()).item()
    print("relative L2 difference of the whole gradient", rel)
    assert rel < 2e-2
    # (the eval forward above ran after one AdamW step at lr 5e-5: compare with the oracle on the UPDATED weights)
    with torch.no_grad():
        ref = np.array([float(x) for x in O.spmm_forward(sd2, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
    print("fused eval", got, "oracle", ref)
    assert_losses(got, ref, "h768_2layer")


def test_weight_gradients_on_the_backward_chains_own_stream_change_nothing(env):
    """EngineOptions.pv_wgrad_inline: the PV encoder's first n layers keep their weight-gradient GEMMs on the stream of their own backward
    chain instead of the shared weight-gradient stream (a scheduling choice: the one launch per weight and step is the same launch).  Same
    seed, train mode with dropout: for n = 3 and all layers the losses and the WHOLE gradient arena sit as close to n = 0 as a second run of
    n = 0 does (the step is not bit-reproducible: fp32 atomics in the loss / column-sum / hub reductions)."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd.options import EngineOptions
    cfg, ocfg = _mid_cfg(env)
    sd = O.init_state_dict(ocfg, seed=5)
    B, Lt = 16, 64
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=31)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(2))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
    res = {}
    for n in (0, 3, 99, -1):                      # (-1: n = 0 again, the run-to-run noise)
        m = SPMM(config=None, spmm_config=cfg, options=EngineOptions.from_env(pv_wgrad_inline=max(n, 0)))
        m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
        m.train()
        m.engine.seed.fill_(777)
        losses = m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))
        sum(losses).backward()
        torch.cuda.synchronize()
        res[n] = (torch.stack([x.detach() for x in losses]).clone(), m.store.grad.detach().clone())
    gn = res[0][1].norm()
    noise = float((res[-1][1] - res[0][1]).norm() / gn)
    assert float(gn) > 0 and noise < 1e-4
    for n in (3, 99):
        rel = float((res[n][1] - res[0][1]).norm() / gn)
        print(f"pv_wgrad_inline={n}: whole-gradient relative L2 difference {rel:.3g} (two runs of n = 0: {noise:.3g})")
        assert (res[n][0] - res[0][0]).abs().max().item() < 1e-4, n
        assert rel <= max(3.0 * noise, 1e-6), n


def test_full_depth_forward_matches_oracle(env):
    """The published architecture (12 text layers with 6 fusion + 6 PV layers, H=768), random-init weights, B=8, Lt=32,
    queue 1024: bf16 pipeline vs the fp32 CPU oracle.  This is the depth at which bf16 rounding has accumulated most."""
    O, SPMM, *_ = env
    cfg, ocfg = _mid_cfg(env, layers=(12, 6, 6), Q=1024)
    sd = O.init_state_dict(ocfg, seed=11)
    m = _mk(SPMM, cfg, sd).eval()
    B, Lt = 8, 32
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(4))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(3))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    aux, oaux = {}, {}
    with torch.no_grad():
        losses = m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)), aux=aux)
        ref = O.spmm_forward(sd, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, aux=oaux)
    got, ref = np.array([float(x) for x in losses]), np.array([float(x) for x in ref])
    print("full depth: hip", got, "oracle", ref, "diff", np.abs(got - ref), "rel", np.abs(got - ref) / np.abs(ref))
    for key in ("prop_embeds", "text_embeds", "prop_feat", "text_feat", "sim_i2t", "vl_output", "mlm_output", "pred"):
        a = aux[key].float().cpu().reshape(oaux[key].shape)
        print(f"  {key}: max|diff| {(a - oaux[key]).abs().max().item():.4g} (ref max {oaux[key].abs().max().item():.3g})")
    assert_losses(got, ref, "full_depth_b8")


def test_benchmark_shape_forward_matches_oracle(env):
    """BASELINE configs[1]'s real shape -- 12+6 layers, H=768, Lt=128, queue 36 864 (the 36 992-wide softmax at temperature
    0.07 that nothing smaller exercises) -- at B=32, forward only, recorded draws, dropout off, vs the fp32 CPU oracle."""
    O, SPMM, *_ = env
    cfg, ocfg = _mid_cfg(env, layers=(12, 6, 6), Q=36864)
    sd = O.init_state_dict(ocfg, seed=13)
    m = _mk(SPMM, cfg, sd).eval()
    B, Lt = 32, 128
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(7))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    aux, oaux = {}, {}
    with torch.no_grad():
        losses = m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)), aux=aux)
        ref = O.spmm_forward(sd, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, aux=oaux)
    got, ref = np.array([float(x) for x in losses]), np.array([float(x) for x in ref])
    print("benchmark shape: hip", got, "oracle", ref, "diff", np.abs(got - ref), "rel", np.abs(got - ref) / np.abs(ref))
    for key in ("prop_feat", "text_feat", "sim_i2t", "sim_t2i", "vl_output", "mlm_output", "pred"):
        a = aux[key].float().cpu().reshape(oaux[key].shape)
        print(f"  {key}: max|diff| {(a - oaux[key]).abs().max().item():.4g} (ref max {oaux[key].abs().max().item():.3g})")
    assert_losses(got, ref, "bench_shape")
    assert int(m.queue_ptr) == B


@pytest.mark.parametrize("shape", ["h768_2layer", "bench_shape"])
def test_losses_match_the_bf16_storage_model_of_the_oracle(env, shape):
    """north_star asks for losses within 1e-3 of the CPU reference.  Against the fp32 oracle the bf16 pipeline misses that on three of
    the four losses at the benchmark shape (LOSS_ATOL) -- and so does ANY implementation that stores activations in bf16: the oracle
    with its tensors rounded to bf16 at the product's storage points (`oracle.bf16_storage`: GEMM / LayerNorm / attention outputs
    and weight shadows; fp32 accumulation, statistics and loss math untouched) deviates from its own fp32 self by the same amounts.
    What is left between the PRODUCT and that storage model is kernel error plus the bf16 roundings that flip when two fp32
    accumulation orders differ in the last bits; it is asserted at 1e-3 on the LM, property and matching losses and at 1.6e-3 on the
    contrastive loss (whose logits carry the 1 / temp = 14x gain; measured at the benchmark shape: 9e-5 / 5.9e-4 / 1.08e-3 / 5.3e-4
    where the storage model itself sits 6.4e-4 / 9.4e-3 / 4.8e-3 / 1.2e-3 from the fp32 oracle)."""
    O, SPMM, *_ = env
    if shape == "bench_shape":
        cfg, ocfg = _mid_cfg(env, layers=(12, 6, 6), Q=36864)
        sd, (B, Lt), neg_roll = O.init_state_dict(ocfg, seed=13), (32, 128), (1, 7)
    else:
        cfg, ocfg = _mid_cfg(env)
        sd, (B, Lt), neg_roll = O.init_state_dict(ocfg, seed=3), (8, 40), (3, 5)
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
    neg = (torch.arange(B).roll(neg_roll[0]), torch.arange(B).roll(neg_roll[1]))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    m = _mk(SPMM, cfg, sd).eval()
    with torch.no_grad():
        got = np.array([float(x) for x in m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))])
        ref32 = np.array([float(x) for x in O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
        with O.bf16_storage():
            ref16 = np.array([float(x) for x in O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
    print(f"{shape}: hip {got}\n  fp32 oracle {ref32}  |hip - fp32| {np.abs(got - ref32)}\n  bf16-storage oracle {ref16}  |hip - bf16 model| {np.abs(got - ref16)}"
          f"  |bf16 model - fp32| {np.abs(ref16 - ref32)}")
    assert (np.abs(got - ref16) <= np.array([1e-3, 1e-3, 1.6e-3, 1e-3])).all(), (got, ref16)


GRAD_FAMILIES = (("embeddings", "embeddings."), ("self-attention q/k/v", ".attention.self."), ("self-attention output", ".attention.output."),
                 ("cross-attention q/k/v", ".crossattention.self."), ("cross-attention output", ".crossattention.output."),
                 ("FFN up", ".intermediate."), ("FFN down + LayerNorm", ".output."), ("heads and projections", ""))


@pytest.mark.parametrize("shape", ["h768_2layer", "bench_shape"])
def test_gradients_match_the_bf16_storage_model_of_the_oracle(env, shape):
    """The backward against the storage model (round 5: `oracle.bf16_storage(backward=True)` also rounds the gradient flowing back through
    every bf16 storage point, as the product's bf16 dX tensors do; weight / bias gradients and their accumulation stay fp32 on both sides).
    Asserted: relative L2 error of the WHOLE gradient, and per parameter family (embeddings, q/k/v, attention outputs, FFN, heads) the
    family's own relative L2 error and its worst tensor.  Dropout off, recorded draws, the published widths at 2+2 layers and the
    benchmark shape (12+6 layers, B = 32, Lt = 128, queue 36 864).  The same numbers against the fp32 oracle are printed beside them."""
    O, SPMM, *_ = env
    if shape == "bench_shape":
        cfg, ocfg = _mid_cfg(env, layers=(12, 6, 6), Q=36864)
        seed, (B, Lt), neg_roll = 13, (32, 128), (1, 7)
    else:
        cfg, ocfg = _mid_cfg(env)
        seed, (B, Lt), neg_roll = 3, (8, 40), (3, 5)
    for c in (ocfg.text, ocfg.prop, cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
    neg = (torch.arange(B).roll(neg_roll[0]), torch.arange(B).roll(neg_roll[1]))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    sd0 = O.init_state_dict(ocfg, seed=seed)
    m = _mk(SPMM, cfg, sd0).train()
    sum(m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))).backward()
    names = O.trainable_names(ocfg)
    hip = {n: m.store.g(n).detach().float().cpu() for n in names}
    del m
    torch.cuda.empty_cache()
    res = {}
    for model in ("bf16 storage model", "fp32 oracle"):
        sd = {k: v.clone() for k, v in sd0.items()}
        for n in names:
            sd[n].requires_grad_(True)
        O._finish_tied(sd)
        if model == "fp32 oracle":
            sum(O.spmm_forward(sd, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, train=True)).backward()
        else:
            with O.bf16_storage(backward=True):
                sum(O.spmm_forward(sd, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, train=True)).backward()
        tot = sum(float((sd[n].grad.double() ** 2).sum()) for n in names if sd[n].grad is not None) ** 0.5
        fam = {f: [0.0, 0.0, 0.0, ""] for f, _ in GRAD_FAMILIES}           # err^2, ref^2, worst per-tensor relative error, its name
        err2 = 0.0
        for n in names:
            rg = sd[n].grad
            if rg is None:
                assert float(hip[n].abs().max()) == 0, n
                continue
            e, r = float((hip[n].reshape(rg.shape) - rg).norm()), float(rg.norm())
            err2 += e * e
            f = next(f for f, key in GRAD_FAMILIES if key in n)
            fam[f][0] += e * e; fam[f][1] += r * r
            if r > 5e-4 * tot and e / r > fam[f][2]:                       # (tensors below the summation-noise floor are not ranked)
                fam[f][2], fam[f][3] = e / r, n
        res[model] = (err2 ** 0.5 / tot, {f: ((v[0] / max(v[1], 1e-30)) ** 0.5, v[2], v[3]) for f, v in fam.items()})
        print(f"{shape}: product vs {model}: whole gradient |g| = {tot:.4f}, relative L2 error {res[model][0]:.5f}")
        for f, (rel, worst, wn) in res[model][1].items():
            print(f"    {f:26s} family rel L2 {rel:.5f}   worst tensor {worst:.4f}  {wn}")
        del sd
    glob, fams = res["bf16 storage model"]
    # stated bounds against the storage model (1.5 x the measured values, profiles/r05_gradient_parity.txt): the product's remaining
    # distance is fp32 accumulation order and the roundings that flip with it
    assert glob < GRAD_VS_STORAGE_MODEL[shape][0], glob
    for f, (rel, worst, wn) in fams.items():
        assert rel < GRAD_VS_STORAGE_MODEL[shape][1] and worst < GRAD_VS_STORAGE_MODEL[shape][2], (f, rel, worst, wn)
    assert glob <= 1.05 * res["fp32 oracle"][0] + 1e-4          # and it is the better predictor of the product than the fp32 oracle


def test_full_benchmark_batch_against_the_oracle_and_its_storage_model(env):
    """The FULL benchmark batch (B = 128, Lt = 128, 12+6 layers, queue 36 864; round 4's one-off tools/parity_b128.py as a test): the four
    losses against the fp32 oracle (the bf16 pipeline's stated deviation, LOSS_ATOL['bench_shape']) and against the oracle's bf16 storage
    model (1e-3 on three losses, 2.1e-3 on the property loss; measured 2.4e-5 / 1.39e-3 / 2.2e-4 / 1.4e-4, profiles/r05_gradient_parity.txt).  ~2 minutes of host time."""
    O, SPMM, *_ = env
    cfg, ocfg = _mid_cfg(env, layers=(12, 6, 6), Q=36864)
    sd = O.init_state_dict(ocfg, seed=13)
    B, Lt = 128, 128
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(7))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    m = _mk(SPMM, cfg, sd).eval()
    with torch.no_grad():
        got = np.array([float(x) for x in m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))])
        ref32 = np.array([float(x) for x in O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
        with O.bf16_storage():
            ref16 = np.array([float(x) for x in O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
    print(f"B = 128: hip {got}\n  |hip - fp32 oracle| {np.abs(got - ref32)}\n  |hip - bf16 storage model| {np.abs(got - ref16)}\n  |storage model - fp32| {np.abs(ref16 - ref32)}")
    assert_losses(got, ref32, "bench_shape")
    assert (np.abs(got - ref16) <= np.array([1e-3, 2.1e-3, 1e-3, 1e-3])).all(), (got, ref16)


def test_fp32_residual_stream_at_the_benchmark_shape(env):
    """EngineOptions.resid_fp32: the residual stream in fp32 through every LayerNorm (fp32 twin of each hidden state),
    fp32 inputs to the ITM / MPM heads and the feature projections -- measured at the benchmark shape (12+6 layers, H=768, B=32,
    Lt=128, queue 36 864; packed text rows as in the real step) against the fp32 CPU oracle, next to the default bf16 stream.
    Asserted: the fp32 stream is not worse than the bf16 one on any loss, and meets north_star's 1e-3 where the table in
    DESIGN.md says it does."""
    O, SPMM, *_ = env
    from spmm_amd.options import EngineOptions
    cfg, ocfg = _mid_cfg(env, layers=(12, 6, 6), Q=36864)
    sd = O.init_state_dict(ocfg, seed=13)
    B, Lt = 32, 128
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=42)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(7))
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    with torch.no_grad():       # (on a copy: the forward moves the queues and the momentum weights in place)
        ref = np.array([float(x) for x in O.spmm_forward({k: v.clone() for k, v in sd.items()}, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg)])
    dev = {}
    for r32 in (False, True):
        m = SPMM(config=None, spmm_config=cfg, options=EngineOptions.from_env(resid_fp32=r32))
        m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
        m.eval()
        with torch.no_grad():
            got = np.array([float(x) for x in m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))])
        dev[r32] = np.abs(got - ref)
        print(f"resid_fp32={r32}: hip {got} oracle {ref} |diff| {dev[r32]}")
        del m
    assert np.all(dev[True] <= np.maximum(1.25 * dev[False], 5e-4)), (dev[True], dev[False])
    assert_losses(ref + dev[True], ref, "bench_shape_r32")


def test_gradients_match_oracle(env):
    """All parameter gradients of sum(losses) vs oracle autograd, dropout off (tiny config, golden draws)."""
    O, SPMM, tiny_config, *_ = env
    ocfg = O.tiny_cfg()
    for c in (ocfg.text, ocfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    cfg = tiny_config()
    for c in (cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sd = O.closed_form_state_dict(ocfg)
    m = _mk(SPMM, cfg, sd).train()
    B, Lt = 8, 24
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=13)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(2))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
    losses = m(prop, ids, mask, alpha=0.3, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))
    sum(losses).backward()
    names = O.trainable_names(ocfg)
    for n in names:
        sd[n].requires_grad_(True)
    O._finish_tied(sd)
    ref_losses = O.spmm_forward(sd, ocfg, prop, ids, mask, 0.3, mpm_mask=mpm, neg_idx=neg, train=True)
    sum(ref_losses).backward()
    print("losses hip", [float(x) for x in losses], "oracle", [float(x) for x in ref_losses])
    assert_losses([float(x) for x in losses], [float(x) for x in ref_losses], "grad_tiny")
    total_r = torch.sqrt(sum((sd[n].grad.double() ** 2).sum() for n in names if sd[n].grad is not None)).item()
    worst, err2 = [], 0.0
    for n in names:
        rg = sd[n].grad
        hg = m.store.g(n).detach().cpu()
        if rg is None:
            assert hg.abs().max().item() == 0, n
            continue
        err = (hg.reshape(rg.shape) - rg).norm().item()
        err2 += err * err
        worst.append((err / max(rg.norm().item(), 1e-12), err, rg.norm().item(), n))
    worst.sort(reverse=True)
    for rel, err, nrm, n in worst[:10]:
        print(f"  rel err {rel:.4f}  abs err {err:.4g}  |g|={nrm:.4g}  {n}")
    glob = err2 ** 0.5 / total_r
    print(f"  whole gradient: |g|={total_r:.4f}  relative L2 error {glob:.5f}")
    # stated tolerances (bf16 activations and gradients, fp32 accumulation):
    #   whole flat gradient: relative L2 error < 1.5e-2
    #   every tensor: error <= 6 % of its own norm, or below the bf16 summation-noise floor of 5e-4 x the total norm
    #   (tensors such as key biases have a true gradient of ~0 -- softmax is shift invariant)
    assert glob < 1.5e-2
    for rel, err, nrm, n in worst:
        assert err <= max(6e-2 * nrm, 5e-4 * total_r), (n, rel, err, nrm)


@pytest.mark.parametrize("resid_fp32", [False, True])
def test_training_trace_vs_reference(env, golden_dir, resid_fp32):
    """First three steps of the reference's recorded AdamW/clip/scheduler trace (later steps of that toy run are chaotic: closed-form
    weights, lr 1e-3, Adam's sign-like first updates).  Run in the default bf16 residual stream and with EngineOptions.resid_fp32 (fp32
    residual stream through the LayerNorms; the backward is the same bf16 one); TRACE_TOL holds the bounds of both."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd.options import EngineOptions
    g = np.load(os.path.join(golden_dir, "train_tiny_b4_l16.npz"))
    cfg = tiny_config()
    for c in (cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sched = {'sched': 'cosine', 'lr': 1e-3, 'epochs': 4, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 1e-4,
             'warmup_epochs': 2, 'cooldown_epochs': 0}
    tc = {'embed_dim': 64, 'temp': 0.07, 'queue_size': 16, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched,
          'optimizer': {'opt': 'adamW', 'lr': 1e-3, 'weight_decay': 0.02}}
    m = SPMM(config=tc, spmm_config=cfg, loader_len=int(g["loader_len"]), options=EngineOptions.from_env(resid_fp32=resid_fp32))
    m.load_state_dict(O.closed_form_state_dict(O.tiny_cfg()))
    m.train()
    opt = m.optimizers()
    loss_tol, gn_tol = TRACE_TOL[resid_fp32]
    B, Lt, seed = int(g["B"]), int(g["Lt"]), int(g["seed"])
    for s, (epoch, bidx) in enumerate(g["plan"][:3]):
        prop, ids, mask = O.synthetic_batch(B, Lt, seed=seed + s)
        m.current_epoch = int(epoch)
        lr_used = opt.param_groups[0]["lr"]
        alpha = tc["alpha"] if epoch > 0 else tc["alpha"] * min(1., int(bidx) / m.loader_len)
        assert abs(alpha - g["alpha"][s]) < 1e-12
        losses = m.fused_step(prop, ids, mask, alpha, mpm_mask=torch.from_numpy(g["mpm_mask"][s]).cuda(),
                              neg_idx=tuple(_cuda(torch.from_numpy(g["prop_neg_idx"][s]), torch.from_numpy(g["text_neg_idx"][s]))))
        # scheduler cadence of training_step
        step_size, warm = 100, m.warmup_steps
        if epoch > 0 and bidx == 0:
            opt.param_groups[0]["lr"] = m.lr_schedulers().lr_at(int(epoch) + warm)
        elif epoch == 0 and bidx % step_size == 0 and bidx <= warm * step_size:
            opt.param_groups[0]["lr"] = m.lr_schedulers().lr_at(int(bidx) // step_size)
        got = losses.cpu().numpy()
        gn = float(opt.grad_norm)
        print(f"resid_fp32={resid_fp32} step {s}: hip {got} ref {g['losses'][s]} |diff| {np.abs(got - g['losses'][s])}  grad-norm hip {gn:.3f} ref {g['grad_norm'][s]:.3f} "
              f"({abs(gn / g['grad_norm'][s] - 1):.4f})")
        assert abs(lr_used - g["lr_used"][s]) < 1e-12 and abs(opt.param_groups[0]["lr"] - g["lr_next"][s]) < 1e-12
        np.testing.assert_allclose(got, g["losses"][s], rtol=0, atol=loss_tol[s])
        np.testing.assert_allclose(gn, g["grad_norm"][s], rtol=gn_tol[s])
        assert int(m.queue_ptr) == int(g["ptr"][s])
        np.testing.assert_allclose(float(m.temp), g["temp"][s], atol=2e-4 * (s + 1))


def test_train_mode_dropout_runs_and_is_seeded(env):
    O, SPMM, tiny_config, *_ = env
    sd = O.closed_form_state_dict(O.tiny_cfg())
    m = _mk(SPMM, tiny_config(), sd).train()
    B, Lt = 4, 16
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=7)
    mpm = torch.zeros(B, 53)
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
    outs = []
    for seed in (1, 1, 2):
        m.load_state_dict(sd)
        m.engine.seed.fill_(seed)
        with torch.no_grad():
            l = m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))
        outs.append(torch.stack(l).cpu())
        assert torch.isfinite(outs[-1]).all()
    # same seed -> same masks (loss sums use atomicAdd, so equality holds to fp32 summation-order noise only)
    assert torch.allclose(outs[0], outs[1], rtol=0, atol=2e-5)
    assert (outs[0] - outs[2]).abs().max().item() > 1e-4          # different seed, different masks


def test_forward_api_draws_new_masks_every_call(env):
    """The reference-style `model(...)` API in train mode: every call must use fresh dropout masks and negative draws
    (the seed is a device counter advanced by the engine's forward, not only by fused_step)."""
    O = env[0]
    m = _tiny_train_model(env)
    prop, ids, mask = O.synthetic_batch(4, 16, seed=7)
    mpm = torch.zeros(4, 53).cuda()
    neg = tuple(_cuda(torch.arange(4).roll(1), torch.arange(4).roll(2)))
    s0 = int(m.engine.seed)
    with torch.no_grad():
        a = torch.stack(m(prop, ids, mask, alpha=0.4, mpm_mask=mpm, neg_idx=neg)).cpu()
        m.load_state_dict(O.closed_form_state_dict(O.tiny_cfg()))
        b = torch.stack(m(prop, ids, mask, alpha=0.4, mpm_mask=mpm, neg_idx=neg)).cpu()
    assert int(m.engine.seed) == s0 + 2
    assert (a - b).abs().max().item() > 1e-4            # same weights, same batch, different masks
    m.eval()
    with torch.no_grad():
        m(prop, ids, mask, alpha=0.4, mpm_mask=mpm, neg_idx=neg)
    assert int(m.engine.seed) == s0 + 2                  # eval forwards draw nothing


def test_nan_step_leaves_queue_and_pointer_untouched(env):
    """SPMM_models.py:132-134 returns before _dequeue_and_enqueue (:208): a non-finite momentum feature must not enter the
    queue (it would poison every similarity until its slot is overwritten) and the optimiser step is skipped."""
    O = env[0]
    m = _tiny_train_model(env, dropout=False)
    prop, ids, mask = O.synthetic_batch(4, 16, seed=7)
    m.fused_step(prop, ids, mask, 0.4)                   # a healthy step first
    q0, t0, p0 = m.store.buffers["prop_queue"].clone(), m.store.buffers["text_queue"].clone(), int(m.queue_ptr)
    w0 = m.store.flat.clone()
    m.store.w("text_proj_m.bias")[0] = float("nan")      # momentum text feature -> NaN similarities
    losses = m.fused_step(prop, ids, mask, 0.4).cpu()
    assert int(m.engine.nan_flag) == 1
    assert torch.equal(m.store.buffers["prop_queue"], q0) and torch.equal(m.store.buffers["text_queue"], t0)
    assert int(m.queue_ptr) == p0
    assert torch.equal(m.store.flat, w0)                 # AdamW skipped on the device
    assert torch.isfinite(m.store.buffers["prop_queue"]).all()
    del losses


def test_checkpoint_resume_continues_the_run(env, tmp_path):
    """save_checkpoint / load_checkpoint carry the Adam moments, step count, lr and the dropout seed: a resumed run takes the
    same next step as the uninterrupted one (dropout on)."""
    O = env[0]
    batches = [O.synthetic_batch(4, 16, seed=20 + i) for i in range(3)]
    a = _tiny_train_model(env)
    for i in range(2):
        a.fused_step(*batches[i], 0.4)
    path = str(tmp_path / "ck.pt")
    a.current_epoch = 1
    a.save_checkpoint(path)
    mpm = (torch.arange(4 * 53).view(4, 53) % 3 == 0).float().cuda()   # the bernoulli draw comes from torch's global generator: pin it
    la = a.fused_step(*batches[2], 0.4, mpm_mask=mpm).cpu()
    b = _tiny_train_model(env)
    b.load_checkpoint(path)
    assert b.current_epoch == 1 and int(b.optimizers().step_count) == 2 and int(b.engine.seed) == int(a.engine.seed) - 1
    lb = b.fused_step(*batches[2], 0.4, mpm_mask=mpm).cpu()
    assert torch.allclose(la, lb, rtol=0, atol=5e-5), (la, lb)          # atomics-order noise only
    da = (a.store.flat - b.store.flat).abs().max().item()
    assert da < 5e-6, da
    ck = torch.load(path, map_location="cpu")
    assert set(ck["state_dict"].keys()) == set(a.state_dict().keys())   # the reference's consumers still find their layout


def test_trainer_runs_the_reference_schedule_and_resumes(env, golden_dir, tmp_path):
    """The driver's pieces on the tiny config, batch 4.  (a) `training_step` -- the hook a trainer calls -- replays the first
    three steps of the REAL reference's recorded run (tests/golden/train_tiny_b4_l16.npz: losses, lr cadence of
    SPMM_models.py:372-378, queue pointer).  (b) spmm_amd.trainer.Trainer (what pretrain.py builds) runs 64 steps on synthetic
    batches with finite losses, writes Lightning-layout checkpoints and resumes from one."""
    O = env[0]
    from spmm_amd.trainer import Trainer
    g = np.load(os.path.join(golden_dir, "train_tiny_b4_l16.npz"))
    B, Lt, seed = int(g["B"]), int(g["Lt"]), int(g["seed"])
    m = _tiny_train_model(env, dropout=False)
    m.loader_len = int(g["loader_len"])
    for s, (epoch, bidx) in enumerate(g["plan"][:3]):
        prop, ids, mask = O.synthetic_batch(B, Lt, seed=seed + s)
        m.current_epoch = int(epoch)
        draws = dict(mpm_mask=torch.from_numpy(g["mpm_mask"][s]).cuda(),
                     neg_idx=tuple(_cuda(torch.from_numpy(g["prop_neg_idx"][s]), torch.from_numpy(g["text_neg_idx"][s]))))
        out = m.training_step((prop.cuda(), (ids.cuda(), mask.cuda()), draws), int(bidx))
        np.testing.assert_allclose(out.cpu().numpy(), g["losses"][s], rtol=0, atol=[2e-2, 6e-2, 0.15][s])
        assert abs(m.optimizers().param_groups[0]["lr"] - g["lr_next"][s]) < 1e-12
        assert int(m.queue_ptr) == int(g["ptr"][s])

    class Loader:
        def __len__(self):
            return 64

        def __iter__(self):
            for s in range(64):
                prop, ids, mask = O.synthetic_batch(B, Lt, seed=1000 + s)
                yield prop.cuda(), (ids.cuda(), mask.cuda())

    m = _tiny_train_model(env, dropout=True)
    tr = Trainer(max_epochs=1, output_dir=str(tmp_path), every_n_train_steps=32, log_every_n_steps=8, quiet=True)
    tr.fit(m, Loader())
    assert tr.global_step == 64 and len(tr.history) == 8
    assert all(np.isfinite(h[2]).all() for h in tr.history) and tr.history[-1][3] > 0
    ck = torch.load(tmp_path / "checkpoint_epoch=0.ckpt", map_location="cpu")
    assert ck["global_step"] == 64 and ck["optimizer_states"][0]["step_count"] == 64 and len(ck["state_dict"]) == len(m.state_dict())
    m2 = _tiny_train_model(env, dropout=True)
    tr2 = Trainer(max_epochs=2, output_dir=str(tmp_path / "r"), every_n_train_steps=0, log_every_n_steps=64, max_steps=66, quiet=True)
    tr2.fit(m2, Loader(), ckpt_path=str(tmp_path / "checkpoint_epoch=0.ckpt"))      # epoch 0 is complete: continues with epoch 1
    assert tr2.global_step == 66 and int(m2.optimizers().step_count) == 66 and m2.current_epoch == 1


def test_on_device_negative_sampling_and_bernoulli(env):
    O, SPMM, tiny_config, *_ = env
    sd = O.closed_form_state_dict(O.tiny_cfg())
    m = _mk(SPMM, tiny_config(), sd).eval()
    prop, ids, mask = O.synthetic_batch(8, 16, seed=3)
    aux = {}
    with torch.no_grad():
        l = m(prop, ids, mask, alpha=0.4, aux=aux)
    assert torch.isfinite(torch.stack(l)).all()
    pn, tn = aux["prop_neg_idx"].cpu(), aux["text_neg_idx"].cpu()
    assert (pn != torch.arange(8)).all() and (tn != torch.arange(8)).all()
    assert set(aux["mpm_mask"].unique().tolist()) <= {0.0, 1.0}


def test_smoke_entry(env):
    import __graft_entry__ as g
    g.smoke()


def test_two_rank_data_parallel_step_on_one_gpu(env):
    """bench.py under torch.distributed.run with 2 ranks sharing this GPU (gloo transports CUDA tensors through the host):
    exercises the real feature all-gather + bucketed gradient averaging around the HIP step and asserts that parameters,
    momentum parameters and queues stay replica-identical (there is no per-step buffer broadcast).  RCCL itself needs one GPU
    per rank and is exercised by the driver's multi-GPU run."""
    import subprocess, sys, json, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # two processes on one GPU: the HIP default of 4 hardware queues each (bench.py does not raise it for gloo; more oversubscribe the
    # hardware scheduler and the run crawls); the watchdog turns a hang into Python stacks after 4 minutes instead of a silent timeout
    envv = dict(os.environ, SPMM_DIST_BACKEND="gloo", SPMM_BENCH_WATCHDOG="240")
    envv.pop("GPU_MAX_HW_QUEUES", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
           "--seq-len", "32", "--layers", "2,1,1", "--queue", "64", "--no-cpu-baseline", "--check-replicas"]
    # no --no-kernel-timing: the instrumented steps after the timed region contain collectives, so every rank must run them
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=envv, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "replicas identical after" in out.stdout and "queue_ptr = " in out.stdout
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    js = json.loads(line)
    assert js["n_gpus"] == 2 and js["config"]["global_batch"] == 16 and all(np.isfinite(js["losses"]))


def test_beam_decode_matches_oracle_teacher_forced(env):
    """PV -> SMILES k-beam decode (d_pv2smiles_batched.py:18-59 on the facades).  The oracle runs the same algorithm in fp32
    on CPU; at every step the HIP path is fed the ORACLE's beam prefixes (teacher forcing, so a near-tie flipped by bf16 cannot
    derail the comparison) and must return the same top-k log-probabilities within 3e-2 and the same ids wherever the margin to
    the next candidate exceeds that tolerance."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    sd = O.closed_form_state_dict(O.tiny_cfg())
    # make the LM head peaky enough that beams differ and SEP shows up: scale the decoder bias a little
    om = O.OracleModule(sd, O.tiny_cfg())
    m = _mk(SPMM, tiny_config(), sd).eval()
    k = 3
    prop = torch.randn(53, generator=torch.Generator().manual_seed(9))
    pe_o = decode.encode_properties(om, prop.reshape(1, -1))
    pe_h = decode.encode_properties(m, prop.reshape(1, -1))
    assert (pe_h.cpu() - pe_o).abs().max().item() < 6e-2
    text = torch.full((1, 1), decode.CLS_ID, dtype=torch.long)
    vo, io = decode_oracle.next_token_topk(om, pe_o, text, k)
    text = torch.cat([torch.full((k, 1), decode.CLS_ID, dtype=torch.long), io.squeeze(0).unsqueeze(-1)], dim=-1)
    cur = vo.squeeze(0)
    for step in range(6):
        vo, io = decode_oracle.next_token_topk(om, pe_o, text, k)
        vh, ih = decode_oracle.next_token_topk(m, pe_h, text.cuda(), k)
        assert (vh.cpu() - vo).abs().max().item() < 3e-2, step
        gap = (vo[:, :-1] - vo[:, 1:]).min(dim=1).values              # margin between consecutive candidates
        for b in range(k):
            if gap[b] > 6e-2:
                assert torch.equal(ih[b].cpu(), io[b]), (step, b)
        k2 = cur[:, None] + vo
        cur, flat = torch.topk(k2.flatten(), k)
        text = torch.cat([text.unsqueeze(1).repeat(1, k, 1), io.unsqueeze(-1)], dim=-1)[flat // k, flat % k]
    # and the free-running search returns well-formed hypotheses
    hyps = decode_oracle.beam_search(m, prop, k=k, max_steps=12)
    for p, seq in hyps:
        assert seq[0] == decode.CLS_ID and seq[-1] == decode.SEP_ID and p <= 0.0


def _peaky_lm(sd, seed=5, sep_gap=1.5):
    """Make next-token distributions well separated and [SEP] a frequent runner-up, so beams diverge and finish."""
    g = torch.Generator().manual_seed(seed)
    b = torch.randn(sd["text_encoder.cls.predictions.bias"].shape, generator=g) * 1.5
    b[3] = b.max() - sep_gap
    sd = dict(sd)
    sd["text_encoder.cls.predictions.bias"] = b
    sd["text_encoder.cls.predictions.decoder.bias"] = b          # tied alias (xbert.py:695-701): both keys are in the state_dict
    return sd


def test_cached_decoder_step_matches_full_prefix_forward(env):
    """One-token-per-step decoding against the K/V cache gives the logits the whole-prefix forward (the reference's cost
    model, d_pv2smiles_single.py:26-44) gives for the last position -- through beam reorders that only touch the ancestry
    table.  Tolerance 3e-2 on log-probabilities (bf16 activations; the two paths use different attention kernels)."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    sd = _peaky_lm(O.closed_form_state_dict(O.tiny_cfg()))
    m = _mk(SPMM, tiny_config(), sd).eval()
    N, k, T = 3, 4, 9
    g = torch.Generator().manual_seed(2)
    props = torch.randn(N, 53, generator=g)
    pe = decode.encode_properties(m, props)
    cached = decode.CachedDecoder(m, pe, k, T + 3)
    full = decode.RecomputeDecoder(m, pe, k, T + 3)
    ids = torch.full((N * k,), decode.CLS_ID, dtype=torch.long, device="cuda")
    for t in range(T):
        lc = torch.log_softmax(cached.step(ids, t).float(), -1)
        lf = torch.log_softmax(full.step(ids, t).float(), -1)
        assert (lc - lf).abs().max().item() < 3e-2, t
        parent = torch.randint(0, k, (N, k), generator=g).cuda()
        cached.reorder(parent, t + 1)
        full.reorder(parent, t + 1)
        ids = torch.randint(4, 300, (N * k,), generator=g).cuda()


def test_batched_cached_beam_search(env):
    """beam_search_batched on the K/V-cache path: hypotheses are well formed, sorted, and their scores are the sums of the
    teacher-forced log-probabilities of the uncached forward (3e-2 per token); against the uncached search of the same
    model the best score of every molecule agrees within 0.1 (a bf16 near-tie may legitimately pick another beam)."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    for sep_gap in (0.4, 0.7):          # 0.4: every molecule collects k finals within a few steps; 0.7: runs to max_steps
        _check_cached_beam_search(O, SPMM, tiny_config, decode, sep_gap)
    # other beam counts: 8 (the one-wave-per-row attention kernel; the 64 candidates of the beam kernel fill the wave), eager and replayed
    for k, graph, sep_gap in ((8, False, 0.4), (8, True, 0.4)):
        _check_cached_beam_search(O, SPMM, tiny_config, decode, sep_gap, k=k, graph=graph)


def test_batched_decode_of_many_molecules_against_the_oracle_search(env):
    """BASELINE configs[3] end to end at a size the CPU yard-stick finishes in seconds: 48 molecules x 5 beams decoded TOGETHER by the
    K/V-cache decoder (graph replay: the default below 2 500 beam rows) against the fp32 oracle model.  (1) Every hypothesis the
    decoder returns is scored by the ORACLE, teacher-forced on the same tokens: the log-probability sums agree within 3e-2 per token.
    (2) Against the reference's sequential one-molecule whole-prefix search on the oracle model (oracle/decode_oracle.py), free-running:
    a bf16 near-tie at the k-th candidate may legitimately drop or swap a beam, so the best hypothesis has to be token-for-token equal
    for most molecules, not all (bound 90 %; measured 48 / 48, worst per-token score difference 3.8e-3)."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    # (LM-head bias seed 10: its k-th and (k+1)-th largest entries are 0.67 apart -- with the default seed they are 0.06 apart, inside the
    # bf16 noise of the logits, and the two searches keep different fifth beams for most molecules)
    sd = _peaky_lm(O.closed_form_state_dict(O.tiny_cfg()), seed=10, sep_gap=0.5)
    om = O.OracleModule(sd, O.tiny_cfg())
    m = _mk(SPMM, tiny_config(), sd).eval()
    N, k, T = 48, 5, 14
    props = torch.randn(N, 53, generator=torch.Generator().manual_seed(12)) * 2
    got = decode.beam_search_batched(m, props, k=k, max_steps=T)
    pe_o = decode.encode_properties(om, props)
    same = n_hyp = 0
    worst = 0.0
    for n in range(N):
        for p_, seq in got[n]:
            assert seq[0] == decode.CLS_ID and seq[-1] == decode.SEP_ID and decode.SEP_ID not in seq[1:-1]
            text = torch.tensor([seq])
            logits = om.text_encoder(text, attention_mask=torch.ones_like(text), encoder_hidden_states=pe_o[n:n + 1],
                                     encoder_attention_mask=torch.ones(1, pe_o.shape[1], dtype=torch.long), return_dict=True, is_decoder=True,
                                     return_logits=True)
            lp = torch.log_softmax(logits[0, :-1].float(), -1).gather(1, text[0, 1:, None]).sum().item()
            worst = max(worst, abs(p_ - lp) / (len(seq) - 1))
            n_hyp += 1
        ref = decode_oracle.beam_search(om, props[n], k=k, max_steps=T)
        same += int(bool(ref) and bool(got[n]) and got[n][0][1] == ref[0][1])
    print(f"batched decode vs oracle: {n_hyp} hypotheses, worst |score - oracle teacher-forced score| per token {worst:.4f}; "
          f"best hypothesis identical to the sequential oracle search for {same} / {N} molecules")
    assert n_hyp > N and worst < 3e-2
    assert same >= 0.9 * N


def test_batched_decode_against_the_reference_search(env, golden_dir):
    """The HIP decoder (K/V cache, batched) against the REAL reference's search (tests/golden/decode_tiny_k5.npz by
    oracle/make_golden_decode.py: d_pv2smiles_batched.py:18-59 run in the dev container): 18 molecules with their own LM-head bias, k = 5, the
    reference's 100 steps.  The best hypothesis must be the reference's token for token -- including the 23-token one -- and the three
    molecules for which the reference finishes nothing must finish nothing here.  bf16 activations may flip a near-tie: at most 2 of the 18
    may differ (measured: 1 -- the 32-token hypothesis, a cycle of three tokens whose [SEP] runner-up wins at position 19 in bf16)."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    g = np.load(os.path.join(golden_dir, "decode_tiny_k5.npz"), allow_pickle=False)
    props, k = torch.from_numpy(g["props"]), int(g["k"])
    same, diff = 0, []
    for n in range(props.shape[0]):
        sd = _peaky_lm(O.closed_form_state_dict(O.tiny_cfg()), seed=int(g["bias_seed"][n]), sep_gap=float(g["sep_gap"][n]))
        m = _mk(SPMM, tiny_config(), sd).eval()
        got = decode.beam_search_batched(m, props[n:n + 1], k=k, max_steps=100)[0]
        want = g["best_ids"][n, :int(g["best_len"][n])].tolist()
        ok = (got == []) if not want else (bool(got) and got[0][1][:-1] == want and got[0][1][-1] == decode.SEP_ID)
        same += int(ok)
        if not ok:
            diff.append((n, want, got[0][1] if got else None))
    print(f"batched HIP decode vs the reference's search: {same} / {props.shape[0]} best hypotheses identical; differing: {diff}")
    assert same >= props.shape[0] - 2, diff


@pytest.mark.parametrize("seed,gap,scale", [(5, 0.9, 10.0), (7, 1.1, 5.0), (7, 1.1, 20.0)])
def test_batched_decode_drops_finished_molecules(env, seed, gap, scale):
    """Molecules that hold their k finals leave the decoded batch (decode.beam_search_batched(compact=True): activations, ancestry rows and
    cross-attention keys / values are gathered for the live ones, the K/V caches stay in place behind a row map): the hypotheses are those
    of the run that keeps every molecule to the end -- token for token, scores within 1e-4 -- on a model whose molecules finish at different
    positions (some never collect five finals), and the batch really shrank.  (The tiny closed-form model hardly reacts to the properties:
    which molecules finish when is decided by near-ties, so the scenarios are picked per attention kernel -- these three shrink the batch
    once or twice with the round-5 kernel.)"""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    sd = _peaky_lm(O.closed_form_state_dict(O.tiny_cfg()), seed=seed, sep_gap=gap)
    m = _mk(SPMM, tiny_config(), sd).eval()
    props = torch.randn(32, 53, generator=torch.Generator().manual_seed(12)) * scale
    whole = decode.beam_search_batched(m, props, k=5, max_steps=40, compact=False, graph=False)      # (eager: a replayed graph has a fixed batch)
    assert decode.last_run["compactions"] == 0
    small = decode.beam_search_batched(m, props, k=5, max_steps=40, compact=True, graph=False)
    run = dict(decode.last_run)
    print(f"decode with compaction: {run}")
    assert run["compactions"] >= 1 and run["final_batch"] < 32
    assert [[h[1] for h in mol] for mol in small] == [[h[1] for h in mol] for mol in whole]
    for a, b in zip(small, whole):
        for (pa, _), (pb, _) in zip(a, b):
            assert abs(pa - pb) < 1e-4


def _check_cached_beam_search(O, SPMM, tiny_config, decode, sep_gap, k=5, graph=None):
    sd = _peaky_lm(O.closed_form_state_dict(O.tiny_cfg()), sep_gap=sep_gap)
    m = _mk(SPMM, tiny_config(), sd).eval()
    N = 6
    props = torch.randn(N, 53, generator=torch.Generator().manual_seed(4)) * 2
    got = decode.beam_search_batched(m, props, k=k, max_steps=14, graph=graph)
    ref = decode.beam_search_batched(m, props, k=k, max_steps=14, cached=False)
    pe = decode.encode_properties(m, props)
    n_hyp = 0
    for n in range(N):
        ps = [p for p, _ in got[n]]
        assert ps == sorted(ps, reverse=True) and len(got[n]) <= k
        for p, seq in got[n]:
            n_hyp += 1
            # (a beam may START with [SEP] -- the reference only looks for [SEP] from the second generated token on,
            #  d_pv2smiles_batched.py:29-41 -- which the 8-beam runs of this bias do; nowhere else may it sit inside)
            assert seq[0] == decode.CLS_ID and seq[-1] == decode.SEP_ID and decode.SEP_ID not in seq[2:-1]
            text = torch.tensor([seq], device="cuda")
            logits = m.text_encoder(text, attention_mask=torch.ones_like(text), encoder_hidden_states=pe[n:n + 1],
                                    encoder_attention_mask=torch.ones(1, pe.shape[1], dtype=torch.long, device="cuda"),
                                    return_dict=True, is_decoder=True, return_logits=True)
            lp = torch.log_softmax(logits.float(), -1)[0, :-1].gather(1, text[0, 1:, None]).sum().item()
            assert abs(lp - p) < 3e-2 * (len(seq) - 1), (n, seq, lp, p)
        if got[n] and ref[n]:
            assert abs(got[n][0][0] - ref[n][0][0]) < 0.1, (n, got[n][0], ref[n][0])
        assert bool(got[n]) == bool(ref[n])
    print(f"cached beam search k={k} graph={graph} sep_gap={sep_gap}: {n_hyp} hypotheses")
    assert n_hyp >= (N if k >= 5 else 1)


def test_decode_at_published_widths(env):
    """BASELINE configs[3] at the published widths (H=768, 12 heads; 2 text layers with 1 fusion layer + 2 PV layers to keep the CPU
    oracle quick): teacher-forced next-token top-k against the fp32 oracle (log-probabilities within 2e-2, same ids where the
    margin exceeds twice that), and the batched K/V-cache search against the uncached search of the same model."""
    O, SPMM, *_ = env
    from spmm_amd import decode
    cfg, ocfg = _mid_cfg(env)
    sd = _peaky_lm(O.init_state_dict(ocfg, seed=21), sep_gap=0.6)
    om = O.OracleModule(sd, ocfg)
    m = _mk(SPMM, cfg, sd).eval()
    k = 3
    prop = torch.randn(53, generator=torch.Generator().manual_seed(19))
    pe_o = decode.encode_properties(om, prop.reshape(1, -1))
    pe_h = decode.encode_properties(m, prop.reshape(1, -1))
    assert (pe_h.cpu() - pe_o).abs().max().item() < 8e-2
    text = torch.full((1, 1), decode.CLS_ID, dtype=torch.long)
    vo, io = decode_oracle.next_token_topk(om, pe_o, text, k)
    text = torch.cat([torch.full((k, 1), decode.CLS_ID, dtype=torch.long), io.squeeze(0).unsqueeze(-1)], dim=-1)
    cur = vo.squeeze(0)
    worst = 0.0
    for step in range(5):
        vo, io = decode_oracle.next_token_topk(om, pe_o, text, k)
        vh, ih = decode_oracle.next_token_topk(m, pe_h, text.cuda(), k)
        worst = max(worst, (vh.cpu() - vo).abs().max().item())
        gap = (vo[:, :-1] - vo[:, 1:]).min(dim=1).values
        for b in range(k):
            if gap[b] > 1e-1:
                assert torch.equal(ih[b].cpu(), io[b]), (step, b)
        k2 = cur[:, None] + vo
        cur, flat = torch.topk(k2.flatten(), k)
        text = torch.cat([text.unsqueeze(1).repeat(1, k, 1), io.unsqueeze(-1)], dim=-1)[flat // k, flat % k]
    print("decode H=768: worst |d log p| vs oracle", worst)
    assert worst < 2e-2                                              # measured 6.8e-3
    N = 5
    props = torch.randn(N, 53, generator=torch.Generator().manual_seed(6)) * 2
    got = decode.beam_search_batched(m, props, k=5, max_steps=12)
    ref = decode.beam_search_batched(m, props, k=5, max_steps=12, cached=False)
    for n in range(N):
        assert bool(got[n]) == bool(ref[n])
        if got[n]:
            assert abs(got[n][0][0] - ref[n][0][0]) < 0.15, (n, got[n][0], ref[n][0])
            for p, seq in got[n]:
                assert seq[0] == decode.CLS_ID and seq[-1] == decode.SEP_ID and p <= 0.0


def test_batched_decode_against_the_reference_search_at_the_published_size(env, golden_dir):
    """BASELINE configs[3] at its published size -- H = 768, 12 text layers (fusion at 6), 6 PV layers -- against the REAL reference's search
    (tests/golden/decode_wide768_k5.npz by oracle/make_golden_decode_wide.py: d_pv2smiles_batched.py:18-59 / d_pv2smiles_single.py:26-44 run in
    the dev container on `init_state_dict(full_cfg(), seed=0)`, one LM-head bias per molecule, k = 5, the reference's 100 steps; hypotheses of
    2 ... 35 tokens and one search that finishes nothing).  The model is built once; only the LM-head bias changes between molecules.  The best
    hypothesis must be the reference's token for token (bf16 activations may flip a near-tie: at most one of the eight may differ), and a
    search the reference leaves empty must stay empty."""
    O, SPMM, tiny_config, SPMMConfig, BertConfig = env
    from spmm_amd import decode
    g = np.load(os.path.join(golden_dir, "decode_wide768_k5.npz"), allow_pickle=False)
    props, k = torch.from_numpy(g["props"]), int(g["k"])
    ocfg = O.full_cfg()
    cfg = SPMMConfig(text=BertConfig(num_hidden_layers=12, fusion_layer=6, add_cross_attention=True),
                     prop=BertConfig(num_hidden_layers=6, fusion_layer=6, vocab_size=1), embed_dim=256, queue_size=ocfg.queue_size)
    m = SPMM(config=None, spmm_config=cfg, no_train=True)
    m.load_state_dict({kk: v.detach().clone() for kk, v in O.init_state_dict(ocfg, seed=int(g["init_seed"])).items()})
    m.eval()
    same, diff = 0, []
    for n in range(props.shape[0]):
        gen = torch.Generator().manual_seed(int(g["bias_seed"][n]))
        b = torch.randn(300, generator=gen) * 1.5
        b[3] = b.max() - float(g["sep_gap"][n])
        m.store.w("text_encoder.cls.predictions.bias").copy_(b.to(m.device))
        m.store.refresh_shadows()
        got = decode.beam_search_batched(m, props[n:n + 1], k=k, max_steps=100)[0]
        want = g["best_ids"][n, :int(g["best_len"][n])].tolist()
        ok = (got == []) if not want else (bool(got) and got[0][1][:-1] == want and got[0][1][-1] == decode.SEP_ID)
        same += int(ok)
        if not ok:
            diff.append((n, want, got[0][1] if got else None))
    print(f"batched HIP decode vs the reference's search at H=768 / 12+6 layers: {same} / {props.shape[0]} best hypotheses identical; differing: {diff}")
    assert same >= props.shape[0] - 1, diff
    # The one allowed mismatch must be a NEAR-TIE for the fp32 oracle, not a wrong search.  The reference's search is replayed on the oracle
    # model (decode_oracle.next_token_topk, the beam bookkeeping of d_pv2smiles_batched.py:23-57) and the HIP path's best hypothesis is followed
    # through it: at the step where the oracle's search PRUNES its prefix, that candidate's cumulative log-probability must be within 3e-2
    # of the k-th candidate the oracle kept (bf16 log-probabilities deviate up to 2e-2 per position -- test above -- so a smaller margin
    # cannot be resolved); if the prefix survives to the end, the two finished hypotheses must be within 3e-2 of each other.
    import decode_oracle
    for n, want, got_ids in diff:
        assert got_ids is not None, "the reference finished a hypothesis, the HIP search none"
        sd = {kk: v.detach().clone() for kk, v in O.init_state_dict(ocfg, seed=int(g["init_seed"])).items()}
        gen = torch.Generator().manual_seed(int(g["bias_seed"][n]))
        b = torch.randn(300, generator=gen) * 1.5
        b[3] = b.max() - float(g["sep_gap"][n])
        sd["text_encoder.cls.predictions.bias"] = b
        om = O.OracleModule(sd, ocfg)
        pv = decode.encode_properties(om, props[n:n + 1])
        start = torch.full((1, 1), decode.CLS_ID, dtype=torch.long)
        score, tok = decode_oracle.next_token_topk(om, pv, start, k)
        beams, beam_lp = torch.cat([start.expand(k, 1), tok.reshape(k, 1)], dim=1), score.reshape(k)
        assert any(bm.tolist() == got_ids[:2] for bm in beams), "the HIP hypothesis' first token is not among the oracle's k best"
        margin = None
        for _ in range(len(got_ids)):
            score, tok = decode_oracle.next_token_topk(om, pv, beams, k)
            cand_lp = beam_lp[:, None] + score
            cand = torch.cat([beams[:, None, :].expand(k, k, beams.shape[1]), tok[:, :, None]], dim=2).reshape(k * k, -1)
            L = cand.shape[1]
            mine = [i for i in range(k * k) if cand[i].tolist() == got_ids[:L]]
            if L == len(got_ids):                                # the finished hypothesis itself: compare with the reference's best
                assert mine, "the HIP hypothesis is not a candidate of the oracle's search"
                ref_lp = float(g["best_lp"][n]) if "best_lp" in g.files else None
                margin = ("final", float(cand_lp.reshape(-1)[mine[0]]), ref_lp)
                break
            flat = cand_lp.reshape(-1).clone()
            flat[(tok == decode.SEP_ID).reshape(-1)] = -1e5      # finished candidates are recorded and struck out
            beam_lp, pick = torch.topk(flat, k)
            if not mine or mine[0] not in pick.tolist():
                assert mine, f"molecule {n}: the HIP hypothesis leaves the oracle's candidate set at length {L} (not a pruning tie)"
                margin = ("pruned", L, float(beam_lp[-1] - flat[mine[0]]))
                break
            beams = cand[pick]
        print(f"molecule {n}: the oracle's search and the HIP path's best hypothesis part ways: {margin}")
        assert margin is not None
        if margin[0] == "pruned":
            assert 0.0 <= margin[2] < 3e-2, margin                # measured 1.7e-2 (molecule 2, length 11)
        elif margin[2] is not None:
            assert abs(margin[1] - margin[2]) < 3e-2, margin


def test_smiles_to_pv_matches_oracle(env):
    """SMILES -> PV autoregressive regression (d_smiles2pv.py:14-52) on the facades vs the fp32 oracle running the same
    loop.  The predictions feed back into the prefix, so bf16 error compounds over the steps: tolerance 5e-2 absolute on
    normalised property values (O(0.3) here) for 12 free-running steps, plus a teacher-forced step that isolates one pass."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    sd = O.closed_form_state_dict(O.tiny_cfg())
    om = O.OracleModule(sd, O.tiny_cfg())
    m = _mk(SPMM, tiny_config(), sd).eval()
    _, ids, mask = O.synthetic_batch(5, 20, seed=3)
    want = decode.smiles_to_pv(om, ids, mask, n_props=12)
    got = decode.smiles_to_pv(m, ids.cuda(), mask.cuda(), n_props=12).cpu()
    assert got.shape == want.shape == (5, 12)
    assert (got - want).abs().max().item() < 5e-2, (got - want).abs().max().item()
    assert want.std().item() > 1e-2                       # not a constant predictor: the comparison is meaningful


def test_packed_text_passes_equal_dense_layout(env):
    """Dropping the padding-token rows from the passes that only read position 0 (P2, P4, P6, P8; spmm_amd/step.py) changes
    nothing: same four losses and the same gradient as the dense layout on a batch with ragged lengths, dropout off.
    (Not bit-exact: the weight-gradient GEMMs sum over a different number of rows, i.e. in a different split order.)"""
    O, SPMM, tiny_config, *_ = env
    cfg = tiny_config()
    for c in (cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sd = O.closed_form_state_dict(O.tiny_cfg())
    B, Lt = 8, 24
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=21)
    assert int(mask.sum()) < B * Lt and int(mask.sum(1).min()) < Lt           # ragged: there is something to drop
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(3))
    neg = (torch.arange(B).roll(2), torch.arange(B).roll(3))
    out = {}
    for pack in (True, False):
        m = _mk(SPMM, cfg, sd).train()
        m.engine.pack_text = pack
        losses = m(prop, ids, mask, alpha=0.3, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))
        sum(losses).backward()
        out[pack] = (torch.stack([l.detach() for l in losses]).cpu(), m.store.grad.detach().clone().cpu(),
                     m.store.buffers["text_queue"].clone().cpu())
    lp, gp, qp = out[True]
    ld, gd, qd = out[False]
    assert torch.allclose(lp, ld, rtol=0, atol=2e-5), (lp, ld)        # the loss sums use fp32 atomicAdd: a few ulps of 8.8 run to run
    assert torch.equal(qp, qd)
    rel = ((gp - gd).norm() / gd.norm()).item()
    print("packed vs dense: losses", lp.tolist(), "gradient rel L2", rel)
    assert rel < 2e-4
    # a mask that is not a prefix falls back to the dense layout (the packed index would not be the position)
    holes = mask.clone()
    holes[0, 2] = 0
    m = _mk(SPMM, cfg, sd).train()
    assert m.engine._pack_plan(holes.to(torch.int32).cuda(), B, Lt) is None
    assert m.engine._pack_plan(mask.to(torch.int32).cuda(), B, Lt)["M"] == int(mask.sum())


def test_full_depth_training_steps_match_oracle(env):
    """Three optimiser steps of the published architecture (12 text layers / 6 fusion + 6 PV layers, H=768) against the
    fp32 CPU oracle's trainer: forward, backward, clip, AdamW (reference lr 5e-5), EMA, queue.  Dropout off, recorded
    bernoulli / negative draws.  Step 0 is a pure forward/backward comparison (losses 1e-3, gradient norm 1e-3 relative).
    Later steps see parameters that both sides updated: Adam's first steps move every weight by +-lr whatever the gradient's
    size, so sign flips of near-zero bf16 gradients are amplified -- the unbounded regression loss (5*MPM, which jumps from 6.6
    to 23 at this random init) is the sensitive one.  Stated tolerances after an update: MLM / ITA / ITM 1 %, 5*MPM 10 %,
    gradient norm 12 %."""
    O, SPMM, *_ = env
    cfg, ocfg = _mid_cfg(env, layers=(12, 6, 6), Q=64)
    for c in (cfg.text, cfg.prop, ocfg.text, ocfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sched = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5, 'warmup_epochs': 20,
             'cooldown_epochs': 0}
    tc = {'embed_dim': 256, 'temp': 0.07, 'queue_size': 64, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched,
          'optimizer': {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}}
    sd = O.init_state_dict(ocfg, seed=17)
    m = SPMM(config=tc, spmm_config=cfg, loader_len=100)
    m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
    m.train()
    ocfg.alpha = 0.4
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    tr = O.OracleTrainer(sd, ocfg, sched, tc['optimizer'], loader_len=100)
    B, Lt = 8, 32
    opt = m.optimizers()
    for s in range(3):
        prop, ids, mask = O.synthetic_batch(B, Lt, seed=100 + s)
        mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(s))
        neg = (torch.arange(B).roll(1 + s), torch.arange(B).roll(2 + s))
        bidx = 50 + s                                              # alpha ramp mid-way, no scheduler event
        ref = tr.step(prop, ids, mask, 0, bidx, mpm_mask=mpm, neg_idx=neg, train=True)
        alpha = tc["alpha"] * min(1., bidx / 100)
        got = m.fused_step(prop, ids, mask, alpha, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg))).cpu().numpy()
        gn = float(opt.grad_norm)
        print(f"step {s}: hip {got} oracle {np.array(ref)} rel {np.abs(got - ref) / np.abs(ref)} grad-norm hip {gn:.4f} oracle {float(tr.grad_norm):.4f}")
        ref = np.array(ref)
        if s == 0:
            np.testing.assert_allclose(got, ref, rtol=2e-3, atol=0)
            np.testing.assert_allclose(gn, float(tr.grad_norm), rtol=2e-3)
        else:
            np.testing.assert_allclose(got[[0, 2, 3]], ref[[0, 2, 3]], rtol=1e-2, atol=0)
            np.testing.assert_allclose(got[1], ref[1], rtol=0.1)
            np.testing.assert_allclose(gn, float(tr.grad_norm), rtol=0.12)
    assert int(m.queue_ptr) == (3 * B) % 64


def test_token_count_hint_equals_device_read(env):
    """fused_step(..., n_tokens=) (the data pipeline's host-side mask sum) must give the step the device read-back gives."""
    O = env[0]
    prop, ids, mask = O.synthetic_batch(8, 24, seed=31)
    mpm = (torch.rand(8, 53, generator=torch.Generator().manual_seed(3)) < 0.5).float()
    neg = (torch.arange(8).roll(1), torch.arange(8).roll(2))
    out = []
    for hint in (None, int(mask.sum())):
        m = _tiny_train_model(env, dropout=False)
        l = m.fused_step(*_cuda(prop, ids, mask), 0.3, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)), n_tokens=hint)
        out.append(([float(x) for x in l], m.store.flat.clone()))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-5)
    assert (out[0][1] - out[1][1]).abs().max().item() < 2.5e-3


@pytest.mark.parametrize("delta", [-7, 9, -40])
def test_wrong_token_count_hint_is_caught_on_the_device(env, delta):
    """A caller's `n_tokens` that contradicts the attention mask (too small / too large) sizes the packed launches wrongly.  Nothing reads the
    mask back, so the device must catch it: spmm_pack_plan raises the hint flag, the step's non-finite flag follows, AdamW / enqueue are
    skipped exactly as for a non-finite loss (SPMM_models.py:132-134: weights, Adam state, queues, queue pointer untouched), every index the
    plan hands to a gather / attention launch stays inside the rows the hint sized, and the NEXT step with the right count is a normal step."""
    from spmm_amd import ops
    O = env[0]
    prop, ids, mask = O.synthetic_batch(8, 24, seed=31)
    true = int(mask.sum())
    mpm = (torch.rand(8, 53, generator=torch.Generator().manual_seed(3)) < 0.5).float()
    neg = (torch.arange(8).roll(1), torch.arange(8).roll(2))
    B, Lt, M = 8, 24, true + delta
    # (1) the plan itself: all indices within bounds whatever the mask says
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    pk = ops.pack_plan(mask.to(torch.int32).cuda().contiguous(), M, bad)
    torch.cuda.synchronize()
    assert int(bad) == 1
    r0, ln = pk["row0"].long(), pk["len"].long()
    assert bool((r0 >= 0).all()) and bool((ln >= 1).all()) and bool((r0 + ln <= M).all())
    assert bool(((pk["rows"] >= 0) & (pk["rows"] < B * Lt)).all())
    assert bool(((pk["gidx2"] >= 0) & (pk["gidx2"] < 2 * B * Lt)).all()) and bool(((pk["gidx4"] >= 0) & (pk["gidx4"] < 2 * B * Lt)).all())
    assert bool(((pk["inv"] >= -1) & (pk["inv"] < M + B * Lt)).all()) and bool(((pk["idx_m"] >= 0) & (pk["idx_m"] < 2 * M)).all())
    # (2) the step: skipped, state untouched; canaries around the arenas the step writes stay intact
    m = _tiny_train_model(env, dropout=False)
    m.fused_step(*_cuda(prop, ids, mask), 0.3, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)), n_tokens=true)      # a normal step first
    torch.cuda.synchronize()
    flat0, am0, av0 = m.store.flat.clone(), m.store.adam_m.clone(), m.store.adam_v.clone()
    q0, ptr0, cnt0 = m.store.buffers["text_queue"].clone(), int(m.queue_ptr), int(m.optimizers().step_count)
    m.fused_step(*_cuda(prop, ids, mask), 0.3, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)), n_tokens=M)
    torch.cuda.synchronize()
    assert int(m.engine.nan_flag) != 0 and int(m.engine.hint_bad) == 1
    assert torch.equal(m.store.flat, flat0) and torch.equal(m.store.adam_m, am0) and torch.equal(m.store.adam_v, av0)
    assert torch.equal(m.store.buffers["text_queue"], q0) and int(m.queue_ptr) == ptr0 and int(m.optimizers().step_count) == cnt0
    # (3) the next step with the right count runs normally (nothing sticky)
    l = m.fused_step(*_cuda(prop, ids, mask), 0.3, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)), n_tokens=true)
    torch.cuda.synchronize()
    assert int(m.engine.nan_flag) == 0 and all(np.isfinite(float(x)) for x in l)
    assert not torch.equal(m.store.flat, flat0) and int(m.optimizers().step_count) == cnt0 + 1


def test_training_step_as_one_hipgraph(env):
    """SPMM.fused_step_graphed: the whole step (zero_grad, forward, backward, clip, AdamW, EMA, enqueue) captured once and replayed.
    Against the eager run of the same dense-layout step on a twin model: identical batches and draws, six steps -- the losses
    of every step agree to the run-to-run noise of the fp32 atomic sums as six training steps amplify it (rtol 5e-3; 7e-4 typical) and so
    do the weights at the end."""
    import time
    O = env[0]
    batches = []
    for i in range(6):
        prop, ids, mask = O.synthetic_batch(8, 24, seed=100 + i)
        mpm = (torch.rand(8, 53, generator=torch.Generator().manual_seed(i)) < 0.5).float()
        neg = (torch.arange(8).roll(1 + i % 3), torch.arange(8).roll(2 + i % 3))
        batches.append(_cuda(prop, ids, mask, mpm, *neg))
    eager, graphed = _tiny_train_model(env, dropout=False), _tiny_train_model(env, dropout=False)
    eager.engine.pack_text = False
    le, lg, host = [], [], []
    for i, (prop, ids, mask, mpm, n0, n1) in enumerate(batches):
        le.append([float(x) for x in eager.fused_step(prop, ids, mask, 0.1 * i, mpm_mask=mpm, neg_idx=(n0, n1))])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = graphed.fused_step_graphed(prop, ids, mask, 0.1 * i, mpm_mask=mpm, neg_idx=(n0, n1))
        host.append(time.perf_counter() - t0)
        lg.append([float(x) for x in out])
    assert isinstance(graphed._graphs[next(iter(graphed._graphs))], tuple)          # captured (step 2) and replayed (steps 3-6)
    print("eager", le[-1], "graph", lg[-1], "host ms per replay", [round(h * 1e3, 3) for h in host])
    np.testing.assert_allclose(np.array(lg), np.array(le), rtol=5e-3)
    assert (graphed.store.flat - eager.store.flat).abs().max().item() < 2.5e-3      # six AdamW steps at lr <= 1e-3, sign noise on ~0 gradients
    assert int(graphed.queue_ptr) == int(eager.queue_ptr) and int(graphed.engine.seed) == int(eager.engine.seed)
    # (the host-side cost of a replay is a wall-clock property: tests/test_zz_timing_gpu.py)


def test_eager_then_graphed_steps_on_one_model(env):
    """An eager fused_step() leaves the off-path shadow rebuild pending on the weight-gradient stream (Engine._pre_bwd); the warm-up and
    the capture of fused_step_graphed() on the SAME model must drain it first (a wait on an event recorded outside the capture is a
    StreamCaptureIsolation error) -- and the mixed sequence eager, graphed x3, eager, load_state_dict must track a purely eager twin."""
    O = env[0]
    batches = []
    for i in range(5):
        prop, ids, mask = O.synthetic_batch(8, 24, seed=300 + i)
        mpm = (torch.rand(8, 53, generator=torch.Generator().manual_seed(40 + i)) < 0.5).float()
        neg = (torch.arange(8).roll(1 + i % 3), torch.arange(8).roll(2 + i % 3))
        batches.append(_cuda(prop, ids, mask, mpm, *neg))
    ref, mixed = _tiny_train_model(env, dropout=False), _tiny_train_model(env, dropout=False)
    ref.engine.pack_text = mixed.engine.pack_text = False
    lr, lm = [], []
    for i, (prop, ids, mask, mpm, n0, n1) in enumerate(batches):
        lr.append([float(x) for x in ref.fused_step(prop, ids, mask, 0.1 * i, mpm_mask=mpm, neg_idx=(n0, n1))])
        step = mixed.fused_step if i in (0, 4) else mixed.fused_step_graphed
        lm.append([float(x) for x in step(prop, ids, mask, 0.1 * i, mpm_mask=mpm, neg_idx=(n0, n1))])
    assert isinstance(mixed._graphs[next(iter(mixed._graphs))], tuple)
    np.testing.assert_allclose(np.array(lm), np.array(lr), rtol=5e-3)
    assert (mixed.store.flat - ref.store.flat).abs().max().item() < 2.5e-3
    # the last eager step's shadow rebuild is still in flight on the weight-gradient stream: a state_dict loaded now must win
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    mixed.load_state_dict(sd)
    torch.cuda.synchronize()
    name = "text_encoder.bert.encoder.layer.0.output.dense"
    wT = mixed.store._wT.get(name)
    if wT is not None:
        want = ref.store.w(name + ".weight").t().to(torch.bfloat16)
        assert torch.equal(wT[:want.shape[0], :want.shape[1]], want)


def test_rccl_code_path_single_rank(env):
    """The collective code path on real RCCL with a one-rank group (this box has one GPU): per-layer asynchronous all-reduces
    issued from the backward streams, the final sweep, the feature all-gather -- the step must produce the losses of the
    plain single-process run (same seed, same batches) and finite values throughout."""
    import subprocess, sys, json, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8", "--seq-len", "32", "--layers", "2,1,1",
           "--queue", "64", "--no-cpu-baseline", "--no-kernel-timing", "--eval-mode"]
    outs = []
    for force, wire in (("1", "fp32"), ("0", "fp32"), ("1", "bf16")):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        envv = dict(os.environ, SPMM_FORCE_DIST=force, SPMM_GRAD_WIRE=wire, MASTER_PORT=str(port), SPMM_SCHEDULE_CHECK="0")   # (same step count in all three)
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=envv, cwd=root)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        outs.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["losses"])
    assert all(np.isfinite(outs[0]))
    # two separate 4-step runs: fp32 atomic accumulation order (bias / LayerNorm gradients) makes them agree to ~5e-4 only
    np.testing.assert_allclose(outs[0], outs[1], rtol=3e-3, atol=0)
    # bf16 reduce-scatter + all-gather of the gradients (RCCL, one rank): gradients rounded to bf16 before AdamW
    np.testing.assert_allclose(outs[2], outs[1], rtol=2e-2, atol=0)


def test_data_parallel_schedule_check_times_both_schedules_and_decides(env):
    """EngineOptions.schedule_check (data-parallel runs): the first 20 steps try the three-stream schedule (NT GEMMs one workgroup per tile
    under the exchange), the one-stream schedule and -- nt_under_comm = "auto", the default -- the three-stream schedule with the persistent
    NT launch (8 warm-up steps, then 12 timed steps rotating between the three; bench.py runs them before its warm-up); the decision and its
    medians are reported in `stream_placement` and `schedule_check`; the losses of the run equal those of a run without the check at the same
    step count (all candidates give the same results).  One-rank RCCL group."""
    import subprocess, sys, json, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--batch", "8", "--seq-len", "32", "--layers", "2,1,1",
            "--queue", "64", "--no-cpu-baseline", "--no-kernel-timing", "--eval-mode"]
    res = {}
    for chk in ("1", "0", "tiles"):
        cmd = base + ["--warmup", "2" if chk != "0" else "22"]          # 20 check steps + 2 = 22 untimed steps either way (bench.py cycles four batches)
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        extra = dict(SPMM_SCHEDULE_CHECK="0" if chk == "0" else "1", **({"SPMM_NT_UNDER_COMM": "tiles"} if chk == "tiles" else {}))
        envv = {k: v for k, v in os.environ.items() if k != "SPMM_NT_UNDER_COMM"}
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root,
                             env=dict(envv, SPMM_FORCE_DIST="1", MASTER_PORT=str(port), **extra))
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        res[chk] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    notes = " | ".join(res["1"].get("stream_placement", []))
    print(notes)
    assert "schedule check:" in notes and "kept" in notes and res["1"]["schedule_check_steps"] == 20
    dec = res["1"]["schedule_check"]
    assert dec["samples_each"] == 4 and dec["kept"] in ("three streams", "one stream", "three streams, persistent NT")
    assert dec["three_streams_ms"] > 0 and dec["one_stream_ms"] > 0 and dec["three_streams_persistent_nt_ms"] > 0
    dec_t = res["tiles"]["schedule_check"]                               # an explicit launch form is not second-guessed: two candidates only
    assert dec_t["three_streams_persistent_nt_ms"] is None and dec_t["kept"] in ("three streams", "one stream")
    assert "schedule check:" not in " | ".join(res["0"].get("stream_placement", []))
    np.testing.assert_allclose(res["1"]["losses"], res["0"]["losses"], rtol=3e-3, atol=0)
    np.testing.assert_allclose(res["tiles"]["losses"], res["0"]["losses"], rtol=3e-3, atol=0)


@pytest.mark.parametrize("case", ["min_len", "full_len_128", "odd_33", "mask_holes", "one_long_rest_short"])
def test_edge_shapes_match_oracle(env, case):
    """Ragged and extreme batches, losses vs the fp32 oracle (tiny config, dropout off, fixed draws): the shortest sequences
    the tokenizer can produce ([CLS] x [SEP]), the longest the kernels take (128, no padding at all: dense path), an odd
    length, an attention mask with holes (not a prefix: the packed layout must fall back to dense), and one full-length row
    among minimal ones (maximum padding).  Tolerance 2e-2 absolute as for the golden forwards."""
    O, SPMM, tiny_config, *_ = env
    ocfg, cfg = O.tiny_cfg(), tiny_config()
    for c in (ocfg.text, ocfg.prop, cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sd = O.closed_form_state_dict(ocfg)
    g = torch.Generator().manual_seed(5)
    B, Lt = {"min_len": (4, 3), "full_len_128": (4, 128), "odd_33": (8, 33), "mask_holes": (4, 16), "one_long_rest_short": (8, 64)}[case]
    prop = torch.randn(B, 53, generator=g)
    ids = torch.randint(4, 300, (B, Lt), generator=g)
    ids[:, 0] = 2
    lens = torch.full((B,), Lt)
    if case == "odd_33":
        lens = torch.randint(3, Lt + 1, (B,), generator=g); lens[0] = Lt
    if case == "one_long_rest_short":
        lens = torch.full((B,), 3); lens[0] = Lt
    for i in range(B):
        ids[i, lens[i] - 1] = 3
        ids[i, lens[i]:] = 0
    mask = (ids != 0).long()
    if case == "mask_holes":
        mask[1, 5] = 0
        mask[2, 3:6] = 0
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=g)
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(2))
    m = _mk(SPMM, cfg, sd).train()
    losses = m(prop, ids, mask, alpha=0.25, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))
    sum(losses).backward()
    got = np.array([float(x) for x in losses])
    with torch.no_grad():
        ref = np.array([float(x) for x in O.spmm_forward(sd, ocfg, prop, ids, mask, 0.25, mpm_mask=mpm, neg_idx=neg, train=True)])
    print(case, "hip", got, "oracle", ref)
    assert np.isfinite(got).all() and torch.isfinite(m.store.grad).all()
    assert_losses(got, ref, "edge_shapes", case)


@pytest.mark.parametrize("Lt", [160, 300])
def test_long_sequences_forward_backward_match_oracle(env, Lt):
    """Lt = 160: the attention kernels' 128 < L <= 256 forms (packed rows); Lt = 300 > 256: the dense step with chunked attention
    (forward merge by log-sum-exp, two-pass D in backward) vs the oracle's losses, and the gradient norm vs the oracle's autograd."""
    O, SPMM, tiny_config, *_ = env
    ocfg, cfg = O.tiny_cfg(), tiny_config()
    for c in (ocfg.text, ocfg.prop, cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sd = O.closed_form_state_dict(ocfg)
    B = 4
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=31)
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(6))
    neg = (torch.arange(B).roll(1), torch.arange(B).roll(3))
    m = _mk(SPMM, cfg, sd).train()
    losses = m(prop, ids, mask, alpha=0.3, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))
    sum(losses).backward()
    got = np.array([float(x) for x in losses])
    names = O.trainable_names(ocfg)
    for n in names:
        sd[n].requires_grad_(True)
    O._finish_tied(sd)
    ref_l = O.spmm_forward(sd, ocfg, prop, ids, mask, 0.3, mpm_mask=mpm, neg_idx=neg, train=True)
    sum(ref_l).backward()
    ref = np.array([float(x) for x in ref_l])
    gn_ref = torch.sqrt(sum((sd[n].grad.double() ** 2).sum() for n in names if sd[n].grad is not None)).item()
    gn = m.store.grad.double().norm().item()
    print(f"Lt={Lt} hip", got, "oracle", ref, "diff", np.abs(got - ref), "grad norm", gn, gn_ref)
    assert_losses(got, ref, f"lt{Lt}")
    assert abs(gn - gn_ref) / gn_ref < 2e-3                 # measured 4e-4


def test_seq_len_256_step_matches_oracle_at_published_widths(env):
    """BASELINE configs[4]'s sequence length (Lt = 256) at H = 768 / 12 heads, 2 text layers (1 fusion) + 2 PV layers, B = 8, ragged
    lengths, packed rows (the default schedule), dropout off: four losses and the whole gradient against the fp32 oracle."""
    O, SPMM, *_ = env
    cfg, ocfg = _mid_cfg(env)
    for c in (ocfg.text, ocfg.prop, cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sd = O.init_state_dict(ocfg, seed=4)
    B, Lt = 8, 256
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=41)
    assert int(mask.sum(1).max()) == Lt and int(mask.sum(1).min()) < Lt
    mpm = torch.bernoulli(torch.full((B, 53), 0.5), generator=torch.Generator().manual_seed(8))
    neg = (torch.arange(B).roll(3), torch.arange(B).roll(5))
    m = _mk(SPMM, cfg, sd).train()
    assert m.engine.pack_text
    losses = m(prop, ids, mask, alpha=0.4, mpm_mask=mpm.cuda(), neg_idx=tuple(_cuda(*neg)))
    sum(losses).backward()
    got = np.array([float(x) for x in losses])
    names = O.trainable_names(ocfg)
    for n in names:
        sd[n].requires_grad_(True)
    O._finish_tied(sd)
    ref_l = O.spmm_forward(sd, ocfg, prop, ids, mask, 0.4, mpm_mask=mpm, neg_idx=neg, train=True)
    sum(ref_l).backward()
    ref = np.array([float(x) for x in ref_l])
    total_r = torch.sqrt(sum((sd[n].grad.double() ** 2).sum() for n in names if sd[n].grad is not None)).item()
    err2 = 0.0
    for n in names:
        if sd[n].grad is not None:
            err2 += (m.store.g(n).detach().cpu().reshape(sd[n].grad.shape) - sd[n].grad).norm().item() ** 2
    glob = err2 ** 0.5 / total_r
    print("Lt=256 H=768 hip", got, "oracle", ref, "diff", np.abs(got - ref), f"whole gradient |g|={total_r:.4f} relative L2 error {glob:.5f}")
    assert_losses(got, ref, "lt256_h768")
    assert glob < 1.5e-2


def test_seq_len_256_matches_reference_golden(env, golden_dir):
    """HIP path vs the REAL reference at BASELINE configs[4]'s sequence length (fixture tests/golden/fwd_tiny_b4_l256.npz: toy widths, ragged
    lengths 128..256, packed rows, the 256-key attention kernels forward and backward): losses, whole-gradient norm and the gradient norms of
    four named tensors."""
    O, SPMM, tiny_config, *_ = env
    g = np.load(os.path.join(golden_dir, "fwd_tiny_b4_l256.npz"))
    cfg = tiny_config()
    for c in (cfg.text, cfg.prop):
        c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sd = O.closed_form_state_dict(O.tiny_cfg())
    m = _mk(SPMM, cfg, sd).train()
    B, Lt = int(g["B"]), int(g["Lt"])
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=int(g["seed"]))
    losses = m(prop, ids, mask, alpha=float(g["alpha"]), mpm_mask=torch.from_numpy(g["mpm_mask"]).cuda(),
               neg_idx=tuple(_cuda(torch.from_numpy(g["prop_neg_idx"]), torch.from_numpy(g["text_neg_idx"]))))
    sum(losses).backward()
    got = np.array([float(x) for x in losses])
    gn = m.store.grad.double().norm().item()
    print("Lt=256 golden: hip", got, "reference", g["losses"], "diff", np.abs(got - g["losses"]), "grad norm", gn, float(g["grad_norm"]))
    assert_losses(got, g["losses"], "lt256_golden")
    np.testing.assert_allclose(gn, float(g["grad_norm"]), rtol=5e-3)
    for k in g.files:
        if k.startswith("gradsum::"):
            hn = m.store.g(k[9:]).double().norm().item()
            assert abs(hn - g[k][2]) <= max(6e-2 * g[k][2], 5e-4 * float(g["grad_norm"])), (k, hn, g[k][2])


def test_wide_model_matches_reference_golden(env, golden_dir):
    """HIP path vs the REAL reference at the published widths (H=768, 12 heads, I=3072, E=256; 2 text layers + 1 PV layer),
    fixture tests/golden/fwd_wide768_b4_l16.npz: losses within 2e-2 absolute (the closed-form weights are ~0.08 in magnitude,
    as in the toy-width goldens: measured 5e-4 / 3e-3 / 1.3e-2 / 5e-3), whole-gradient norm within 1 % (measured 0.4 %), the
    gradient norms of four named tensors within max(6 %, 5e-4 of the whole norm), the queue columns written by the step within 2e-3."""
    O, SPMM, tiny_config, SPMMConfig, BertConfig = env
    g = np.load(os.path.join(golden_dir, "fwd_wide768_b4_l16.npz"))
    t = BertConfig(num_hidden_layers=2, fusion_layer=1, add_cross_attention=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    p = BertConfig(num_hidden_layers=1, fusion_layer=1, vocab_size=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg = SPMMConfig(text=t, prop=p, embed_dim=256, queue_size=16)
    ot = O.BertCfg(num_hidden_layers=2, fusion_layer=1)
    op = O.BertCfg(num_hidden_layers=1, fusion_layer=1, vocab_size=1)
    sd = O.closed_form_state_dict(O.SPMMCfg(text=ot, prop=op, embed_dim=256, queue_size=16))
    m = _mk(SPMM, cfg, sd).train()
    B, Lt = int(g["B"]), int(g["Lt"])
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=int(g["seed"]))
    losses = m(prop, ids, mask, alpha=float(g["alpha"]), mpm_mask=torch.from_numpy(g["mpm_mask"]).cuda(),
               neg_idx=tuple(_cuda(torch.from_numpy(g["prop_neg_idx"]), torch.from_numpy(g["text_neg_idx"]))))
    sum(losses).backward()
    got = np.array([float(x) for x in losses])
    gn = m.store.grad.double().norm().item()
    print("wide: hip", got, "reference", g["losses"], "grad norm", gn, float(g["grad_norm"]))
    assert_losses(got, g["losses"], "wide_golden")
    np.testing.assert_allclose(gn, float(g["grad_norm"]), rtol=1e-2)
    for k in g.files:
        if k.startswith("gradsum::"):
            # same per-tensor criterion as test_gradients_match_oracle: 6 % of the tensor's own norm or the bf16 summation-noise
            # floor of 5e-4 x the whole-gradient norm (key weights have near-zero true gradients: softmax is shift invariant)
            got_n, ref_n = m.store.g(k[9:]).double().norm().item(), float(g[k][2])
            print("  ", k[9:], got_n, ref_n)
            assert abs(got_n - ref_n) <= max(6e-2 * ref_n, 5e-4 * float(g["grad_norm"])), (k, got_n, ref_n)
    sdo = m.state_dict()
    np.testing.assert_allclose(sdo["prop_queue"][:, :B].cpu().numpy(), g["prop_queue_head"], atol=2e-3)
    np.testing.assert_allclose(sdo["text_queue"][:, :B].cpu().numpy(), g["text_queue_head"], atol=2e-3)
    assert int(sdo["queue_ptr"]) == int(g["queue_ptr"][0])


def test_conditional_and_stochastic_generation(env):
    """Generation conditioned on a subset of the properties (the rest replaced by the mask token, d_pv2smiles_single.py:66-70)
    and the stochastic candidate branch (:37-40): the masked PV encoding matches the oracle's, and sampled hypotheses are well
    formed with scores equal to the teacher-forced log-probabilities of the uncached forward."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    sd = _peaky_lm(O.closed_form_state_dict(O.tiny_cfg()), sep_gap=0.4)
    sd["property_mask"] = torch.randn_like(sd["property_mask"]) * 0.5
    om = O.OracleModule(sd, O.tiny_cfg())
    m = _mk(SPMM, tiny_config(), sd).eval()
    props = torch.randn(4, 53, generator=torch.Generator().manual_seed(8))
    pmask = torch.zeros(53); pmask[20:] = 1
    pe_h = decode.encode_properties(m, props, pmask)
    pe_o = decode.encode_properties(om, props, pmask)
    assert (pe_h.cpu() - pe_o).abs().max().item() < 6e-2
    assert (pe_o - decode.encode_properties(om, props)).abs().max().item() > 1e-2           # the mask changes the encoding
    g = torch.Generator(device="cuda").manual_seed(3)
    hyps = decode.beam_search_batched(m, props, k=4, max_steps=10, prop_mask=pmask, stochastic=True, generator=g)
    n = 0
    for i, hs in enumerate(hyps):
        for p, seq in hs:
            n += 1
            assert seq[0] == decode.CLS_ID and seq[-1] == decode.SEP_ID and p <= 0.0
            text = torch.tensor([seq], device="cuda")
            logits = m.text_encoder(text, attention_mask=torch.ones_like(text), encoder_hidden_states=pe_h[i:i + 1],
                                    encoder_attention_mask=torch.ones(1, 54, dtype=torch.long, device="cuda"), return_dict=True,
                                    is_decoder=True, return_logits=True)
            lp = torch.log_softmax(logits.float(), -1)[0, :-1].gather(1, text[0, 1:, None]).sum().item()
            assert abs(lp - p) < 3e-2 * (len(seq) - 1)
    assert n >= 4


def test_graph_replayed_decode_equals_eager(env):
    """beam_search_batched(graph=True): one decode position captured as a hipGraph (step index, cache slot, ancestry column and
    beam bookkeeping all in device memory) and replayed -- same hypotheses and scores as the eager loop; and the same again with the
    beam bookkeeping as tensor operations instead of the one-launch kernel (the fallback for more than 8 beams)."""
    O, SPMM, tiny_config, *_ = env
    from spmm_amd import decode
    for sep_gap in (0.4, 0.7):
        sd = _peaky_lm(O.closed_form_state_dict(O.tiny_cfg()), sep_gap=sep_gap)
        m = _mk(SPMM, tiny_config(), sd).eval()
        props = torch.randn(6, 53, generator=torch.Generator().manual_seed(4)) * 2
        eager = decode.beam_search_batched(m, props, k=5, max_steps=14, graph=False)
        graphed = decode.beam_search_batched(m, props, k=5, max_steps=14, graph=True)
        runs = [graphed]
        decode.FUSED_BEAM_STEP = False                # the tensor-op bookkeeping (BeamBook.update / update_dev), eager and replayed
        try:
            runs += [decode.beam_search_batched(m, props, k=5, max_steps=14, graph=False), decode.beam_search_batched(m, props, k=5, max_steps=14, graph=True)]
        finally:
            decode.FUSED_BEAM_STEP = True
        for other in runs:
            assert len(eager) == len(other)
            for a, b in zip(eager, other):
                assert [s for _, s in a] == [s for _, s in b]
                assert all(abs(pa - pb) < 1e-5 for (pa, _), (pb, _) in zip(a, b))
        assert sum(len(h) for h in eager) >= 6
