"""DEV-ONLY generator of tests/golden/decode_wide768_k5.npz -- the REAL reference's PV -> SMILES beam search at the PUBLISHED size
(config_bert.json / config_bert_property.json: H = 768, 12 text layers with fusion at 6, 6 PV layers), run once in this container.

    python oracle/make_golden_decode_wide.py            (a few minutes of CPU)

Same procedure as make_golden_decode.py (`evaluate` d_pv2smiles_batched.py:18-59 and `generate` d_pv2smiles_single.py:26-44 imported from
/root/reference and driven unchanged, stand-in tokenizer, stubbed RDKit imports), on the reference module loaded with
`spmm_oracle.init_state_dict(full_cfg(), seed=0)` -- the reference's own initial distributions, reproducible from the seed on the GPU box,
so the fixture carries no weights -- and one LM-head bias per molecule (`make_golden_decode.peaky_bias`: well separated next-token
distributions with [SEP] near the top, so the searches end after 2 ... ~30 tokens; random-init weights alone never rank [SEP] first).
The fixture is data only: property vectors, bias parameters, best hypothesis per molecule and its log-probability."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import spmm_oracle as O  # noqa: E402
import make_golden as MG  # noqa: E402
import make_golden_decode as MGD  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
# (bias seed, [SEP] gap) per molecule: picked from a scan of 28 combinations (`--scan`) for a spread of hypothesis lengths -- 2 ... 35 tokens
# and one search that finishes nothing within the reference's 100 steps
CASES = [(4, 0.9), (5, 0.9), (6, 0.9), (8, 1.2), (11, 1.2), (12, 0.9), (0, 0.9), (1, 0.9)]


def wide_ref_model():
    cfg = O.full_cfg()

    def over(c):
        return dict(hidden_size=c.hidden_size, num_attention_heads=c.num_attention_heads, intermediate_size=c.intermediate_size,
                    num_hidden_layers=c.num_hidden_layers, fusion_layer=c.fusion_layer, encoder_width=c.encoder_width, vocab_size=c.vocab_size,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    config = {'embed_dim': cfg.embed_dim, 'temp': cfg.temp, 'mlm_probability': 0.15, 'queue_size': cfg.queue_size, 'momentum': cfg.momentum,
              'alpha': cfg.alpha, 'schedular': MG.SCHED, 'optimizer': MG.OPT, 'loader_len': 10}
    m = ref_shim.build_reference_spmm(over(cfg.text), over(cfg.prop), config)
    missing = m.load_state_dict(O.init_state_dict(cfg, seed=0), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


if __name__ == "__main__":
    ref_shim._install()
    MGD._stub_modules()
    import d_pv2smiles_batched as ref_decode          # /root/reference (on sys.path through ref_shim)
    torch.set_num_threads(8)
    m = wide_ref_model()
    if "--scan" in sys.argv:
        CASES = [(sd, gp) for sd in range(14) for gp in (0.9, 1.2)]
    N, k = len(CASES), 5
    # every molecule's property vector comes from its own seed: a case keeps its search whatever else is in the list
    props = torch.stack([torch.randn(53, generator=torch.Generator().manual_seed(1000 + 16 * sd + int(10 * gp))) for sd, gp in CASES])
    best, t0 = [], time.time()
    for n, (seed, gap) in enumerate(CASES):
        b = MGD.peaky_bias(seed, gap)
        with torch.no_grad():
            m.text_encoder.cls.predictions.bias.copy_(b)
            m.text_encoder.cls.predictions.decoder.bias.copy_(b)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                _, cand = ref_decode.evaluate(m, [(props[n:n + 1], ["[CLS]"])], MGD.IdTokenizer(), "cpu", stochastic=False, k=k)
            best.append([int(t) for t in cand[0].split()])
        except IndexError:
            best.append([])
        print(f"molecule {n}: {len(best[-1])} tokens  ({time.time() - t0:.0f} s)", flush=True)
    if "--scan" in sys.argv:
        print("scan:", [(c, len(x)) for c, x in zip(CASES, best)])
        sys.exit(0)
    L = max(len(x) for x in best)
    arr = np.zeros((N, max(L, 1)), dtype=np.int64)
    for n, x in enumerate(best):
        arr[n, :len(x)] = x
    np.savez_compressed(os.path.join(OUT, "decode_wide768_k5"), props=props.numpy(), k=np.int64(k), bias_seed=np.array([c[0] for c in CASES]),
                        sep_gap=np.array([c[1] for c in CASES]), best_ids=arr, best_len=np.array([len(x) for x in best], dtype=np.int64),
                        init_seed=np.int64(0))
    print("decode_wide768_k5:", [len(x) for x in best])
