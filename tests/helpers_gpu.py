"""Helpers shared by the GPU test modules."""


def _cuda(*ts):
    return [t.cuda() for t in ts]


def _tiny_train_model(env, dropout=True):
    O, SPMM, tiny_config, *_ = env
    cfg = tiny_config()
    if not dropout:
        for c in (cfg.text, cfg.prop):
            c.hidden_dropout_prob = c.attention_probs_dropout_prob = 0.0
    sched = {'sched': 'cosine', 'lr': 1e-3, 'epochs': 4, 'min_lr': 1e-5, 'decay_rate': 1, 'warmup_lr': 1e-4,
             'warmup_epochs': 2, 'cooldown_epochs': 0}
    tc = {'embed_dim': 64, 'temp': 0.07, 'queue_size': 16, 'momentum': 0.995, 'alpha': 0.4, 'schedular': sched,
          'optimizer': {'opt': 'adamW', 'lr': 1e-3, 'weight_decay': 0.02}}
    m = SPMM(config=tc, spmm_config=cfg, loader_len=10)
    m.load_state_dict(O.closed_form_state_dict(O.tiny_cfg()))
    return m.train()
