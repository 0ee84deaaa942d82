"""DEV-ONLY generator of tests/golden/*.npz -- runs the REAL reference once, in this container.

    python oracle/make_golden.py

Imports /root/reference through oracle/ref_shim.py, loads the closed-form weights of
``spmm_oracle.closed_form_state_dict`` (no weight blobs, no RNG stream to match), wraps
``torch.bernoulli`` / ``torch.multinomial`` to RECORD the reference's random draws
(SPMM_models.py:85,166,174), and stores inputs + draws + outputs.  The fixtures are data
only; no reference source text is written anywhere.  Nothing here runs on the GPU box.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import spmm_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SCHED = {'sched': 'cosine', 'lr': 5e-5, 'epochs': 30, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 5e-5,
         'warmup_epochs': 20, 'cooldown_epochs': 0}
SCHED_RAMP = dict(SCHED, lr=1e-3, warmup_lr=1e-4, warmup_epochs=2, epochs=4)   # exercises warm-up + cosine
OPT = {'opt': 'adamW', 'lr': 5e-5, 'weight_decay': 0.02}


def ref_model(cfg: O.SPMMCfg, dropout: float, sched=SCHED, opt=OPT, loader_len=10):
    def over(c):
        return dict(hidden_size=c.hidden_size, num_attention_heads=c.num_attention_heads,
                    intermediate_size=c.intermediate_size, num_hidden_layers=c.num_hidden_layers,
                    fusion_layer=c.fusion_layer, encoder_width=c.encoder_width, vocab_size=c.vocab_size,
                    hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    config = {'embed_dim': cfg.embed_dim, 'temp': cfg.temp, 'mlm_probability': 0.15, 'queue_size': cfg.queue_size,
              'momentum': cfg.momentum, 'alpha': cfg.alpha, 'schedular': sched, 'optimizer': opt,
              'loader_len': loader_len}
    m = ref_shim.build_reference_spmm(over(cfg.text), over(cfg.prop), config)
    sd = O.closed_form_state_dict(cfg)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    # load_state_dict copies into both aliases of tied tensors; they stay tied
    assert m.text_encoder.cls.predictions.decoder.weight.data_ptr() == \
        m.text_encoder.bert.embeddings.word_embeddings.weight.data_ptr()
    return m


class Recorder:
    """Record the reference's bernoulli / multinomial draws."""

    def __enter__(self):
        self.bern, self.multi = [], []
        self._b, self._m = torch.bernoulli, torch.multinomial

        def bern(*a, **k):
            r = self._b(*a, **k)
            self.bern.append(r.clone())
            return r

        def multi(*a, **k):
            r = self._m(*a, **k)
            self.multi.append(r.clone())
            return r

        torch.bernoulli, torch.multinomial = bern, multi
        return self

    def __exit__(self, *a):
        torch.bernoulli, torch.multinomial = self._b, self._m


def checks(m, names):
    sd = m.state_dict()
    return {f"chk::{n}": np.array([sd[n].double().sum().item(), sd[n].double().abs().sum().item()]) for n in names}


CHK = ["text_encoder.bert.encoder.layer.1.crossattention.self.key.weight",
       "property_encoder.encoder.layer.0.intermediate.dense.weight",
       "text_encoder.bert.embeddings.word_embeddings.weight",
       "text_encoder_m.bert.encoder.layer.1.output.dense.weight", "property_proj_m.weight", "temp"]


def fixture_forward(name, cfg, B, Lt, alpha, seed):
    torch.manual_seed(1234)
    m = ref_model(cfg, dropout=0.1)
    m.eval()
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=seed)
    with Recorder() as rec:
        losses = m(prop, ids, mask, alpha=alpha)
    assert len(rec.bern) == 1 and len(rec.multi) == 2 * B
    neg = torch.stack([r.reshape(()) for r in rec.multi])
    out = dict(prop=prop.numpy(), ids=ids.numpy(), mask=mask.numpy(), alpha=np.float64(alpha),
               mpm_mask=rec.bern[0].numpy(), prop_neg_idx=neg[:B].numpy(), text_neg_idx=neg[B:].numpy(),
               losses=np.array([float(x) for x in losses], dtype=np.float64),
               prop_queue=m.prop_queue.numpy().copy(), text_queue=m.text_queue.numpy().copy(),
               queue_ptr=m.queue_ptr.numpy().copy(), temp=np.float64(m.temp.item()))
    out.update(checks(m, CHK))
    # a second forward from the mutated state (queue/EMA/ptr carry-over), new batch, other alpha
    prop2, ids2, mask2 = O.synthetic_batch(B, Lt, seed=seed + 1)
    with Recorder() as rec:
        losses2 = m(prop2, ids2, mask2, alpha=0.0)
    neg2 = torch.stack([r.reshape(()) for r in rec.multi])
    out.update(mpm_mask2=rec.bern[0].numpy(), prop_neg_idx2=neg2[:B].numpy(), text_neg_idx2=neg2[B:].numpy(),
               losses2=np.array([float(x) for x in losses2], dtype=np.float64),
               queue_ptr2=m.queue_ptr.numpy().copy())
    # intermediates from submodule calls on the (now mutated) module, for block-level parity
    with torch.no_grad():
        prop_in = torch.sin(torch.arange(B * 54 * cfg.text.hidden_size, dtype=torch.float32) * 0.11
                            ).reshape(B, 54, -1) * 0.5
        pe = m.property_encoder(inputs_embeds=prop_in, return_dict=True).last_hidden_state
        pec = m.property_encoder(inputs_embeds=prop_in, is_decoder=True, return_dict=True).last_hidden_state
        te = m.text_encoder.bert(ids, attention_mask=mask, return_dict=True, mode='text').last_hidden_state
        fu = m.text_encoder.bert(encoder_embeds=pe, attention_mask=torch.ones(B, 54, dtype=torch.long),
                                 encoder_hidden_states=te, encoder_attention_mask=mask, return_dict=True,
                                 mode='fusion').last_hidden_state
        lg = m.text_encoder(ids, attention_mask=mask, encoder_hidden_states=pe,
                            encoder_attention_mask=torch.ones(B, 54, dtype=torch.long), return_dict=True,
                            is_decoder=True, return_logits=True)
    out.update(blk_prop_in=prop_in.numpy(), blk_prop_enc=pe.numpy(), blk_prop_enc_causal=pec.numpy(),
               blk_text_enc=te.numpy(), blk_fusion=fu.numpy(), blk_logits=lg.numpy())
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, out["losses"], out["losses2"])


def wide_cfg() -> O.SPMMCfg:
    """The published widths (H=768, 12 heads, I=3072, E=256) at reduced depth: 2 text layers (1 fusion) + 1 PV layer."""
    t = O.BertCfg(num_hidden_layers=2, fusion_layer=1)
    p = O.BertCfg(num_hidden_layers=1, fusion_layer=1, vocab_size=1)
    return O.SPMMCfg(text=t, prop=p, embed_dim=256, queue_size=16)


def fixture_wide(name, cfg, B, Lt, alpha, seed):
    """Losses + gradient norm of the REAL reference at the real widths (train mode, dropout 0): small fixture (inputs are
    regenerated from the seed, weights are closed-form), pins the oracle and the HIP path beyond the toy width."""
    torch.manual_seed(1234)
    m = ref_model(cfg, dropout=0.0)
    m.train()
    prop, ids, mask = O.synthetic_batch(B, Lt, seed=seed)
    with Recorder() as rec:
        losses = m(prop, ids, mask, alpha=alpha)
    sum(losses).backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
    neg = torch.stack([r.reshape(()) for r in rec.multi])
    out = dict(alpha=np.float64(alpha), seed=np.int64(seed), B=np.int64(B), Lt=np.int64(Lt), mpm_mask=rec.bern[0].numpy(),
               prop_neg_idx=neg[:B].numpy(), text_neg_idx=neg[B:].numpy(),
               losses=np.array([float(x) for x in losses], dtype=np.float64), grad_norm=np.float64(gn),
               queue_ptr=m.queue_ptr.numpy().copy(), temp=np.float64(m.temp.item()),
               prop_queue_head=m.prop_queue[:, :B].numpy().copy(), text_queue_head=m.text_queue[:, :B].numpy().copy())
    for n in ("text_encoder.bert.encoder.layer.1.crossattention.self.key.weight", "property_encoder.encoder.layer.0.intermediate.dense.weight",
              "text_proj.weight", "itm_head.weight"):
        g = dict(m.named_parameters())[n].grad
        out["gradsum::" + n] = np.array([g.double().sum().item(), g.double().abs().sum().item(), g.double().norm().item()])
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, out["losses"], gn)


def fixture_train(name, cfg, B, Lt, steps, seed):
    """Train-mode trace with dropout 0 (train() but p=0): losses, grad-norm, lr, temp, ptr, param checksums.
    Drives the reference's own configure_optimizers() (AdamW + scheduler factory) and restates only the
    Lightning-dependent glue of training_step (SPMM_models.py:348-380)."""
    torch.manual_seed(1234)
    loader_len = 4
    m = ref_model(cfg, dropout=0.0, sched=SCHED_RAMP, opt=dict(OPT, lr=1e-3), loader_len=loader_len)
    m.train()
    (optimizer,), (scheduler,) = m.configure_optimizers()
    rows, draws, grads0 = [], [], {}
    plan = [(0, 0), (0, 100), (0, 200), (1, 0), (2, 0)][:steps]       # (epoch, batch_idx): hits both cadences
    for s, (epoch, batch_idx) in enumerate(plan):
        prop, ids, mask = O.synthetic_batch(B, Lt, seed=seed + s)
        optimizer.zero_grad()
        alpha = m.config['alpha'] if epoch > 0 else m.config['alpha'] * min(1., batch_idx / m.loader_len)
        with Recorder() as rec:
            losses = m(prop, ids, mask, alpha=alpha)
        loss = sum(losses)
        loss.backward()
        if s == 0:
            for n, p in m.named_parameters():
                if p.grad is not None and (p.numel() <= 4096 or n in CHK):
                    grads0["grad0::" + n] = p.grad.numpy().copy()
            grads0["grad0_none"] = np.array([n for n, p in m.named_parameters()
                                             if p.requires_grad and p.grad is None])
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.)
        optimizer.step()
        lr_used = optimizer.param_groups[0]['lr']
        warm = m.warmup_steps
        if epoch > 0 and batch_idx == 0:
            scheduler.step(epoch + warm)
        elif epoch == 0 and batch_idx % 100 == 0 and batch_idx <= warm * 100:
            scheduler.step(batch_idx // 100)
        neg = torch.stack([r.reshape(()) for r in rec.multi])
        draws.append((rec.bern[0].numpy(), neg[:B].numpy(), neg[B:].numpy()))
        row = dict(losses=[float(x) for x in losses], grad_norm=float(gn), lr_used=lr_used,
                   lr_next=optimizer.param_groups[0]['lr'], temp=m.temp.item(), ptr=int(m.queue_ptr), alpha=alpha)
        row.update({k: v for k, v in checks(m, CHK).items()})
        rows.append(row)
        print(name, s, row['losses'], row['grad_norm'], row['lr_used'], row['lr_next'])
    out = dict(plan=np.array(plan), seed=np.int64(seed), B=np.int64(B), Lt=np.int64(Lt), loader_len=np.int64(loader_len),
               losses=np.array([r['losses'] for r in rows]), grad_norm=np.array([r['grad_norm'] for r in rows]),
               lr_used=np.array([r['lr_used'] for r in rows]), lr_next=np.array([r['lr_next'] for r in rows]),
               temp=np.array([r['temp'] for r in rows]), ptr=np.array([r['ptr'] for r in rows]),
               alpha=np.array([r['alpha'] for r in rows]),
               mpm_mask=np.stack([d[0] for d in draws]), prop_neg_idx=np.stack([d[1] for d in draws]),
               text_neg_idx=np.stack([d[2] for d in draws]))
    for k in rows[0]:
        if k.startswith("chk::"):
            out[k] = np.stack([r[k] for r in rows])
    out.update(grads0)
    np.savez_compressed(os.path.join(OUT, name), **out)


def fixture_lr():
    """Scheduler known-answer table straight from the reference's scheduler package."""
    ref_shim._install()
    from scheduler import create_scheduler
    import SPMM_models
    rows = []
    for sched in (SCHED, SCHED_RAMP):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=sched['lr'])
        s, _ = create_scheduler(SPMM_models.AttrDict(sched), opt)
        vals = [opt.param_groups[0]['lr']]
        for t in range(0, 60):
            s.step(t)
            vals.append(opt.param_groups[0]['lr'])
        rows.append(vals)
    np.savez_compressed(os.path.join(OUT, "lr_schedule"), table=np.array(rows))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    cfg = O.tiny_cfg()
    fixture_forward("fwd_tiny_b4_l16", cfg, B=4, Lt=16, alpha=0.4, seed=7)
    fixture_forward("fwd_tiny_b8_l24", cfg, B=8, Lt=24, alpha=0.25, seed=11)
    fixture_train("train_tiny_b4_l16", cfg, B=4, Lt=16, steps=5, seed=21)
    fixture_lr()
    fixture_wide("fwd_wide768_b4_l16", wide_cfg(), B=4, Lt=16, alpha=0.3, seed=33)
    # BASELINE configs[4]'s sequence length on the REAL reference (toy widths, ragged lengths 128..256): pins the 256-long attention path
    fixture_wide("fwd_tiny_b4_l256", cfg, B=4, Lt=256, alpha=0.3, seed=35)
