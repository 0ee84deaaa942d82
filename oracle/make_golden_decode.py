"""DEV-ONLY generator of tests/golden/decode_tiny_k5.npz -- runs the REAL reference's PV -> SMILES beam search once, in this container.

    python oracle/make_golden_decode.py

`evaluate` (d_pv2smiles_batched.py:18-59) and `generate` (d_pv2smiles_single.py:26-44) are imported from /root/reference and driven
unchanged over 18 synthetic property vectors with the reference SPMM module (toy widths, closed-form weights).  Every molecule gets its
own LM-head bias (seed, distance of [SEP] from the top -- `CASES`), chosen by a scan so that the searches differ: best hypotheses of 1 to 32
tokens, and two molecules for which NO hypothesis finishes within the reference's 100 steps (its `evaluate` then raises IndexError on an empty
list, d_pv2smiles_batched.py:46; recorded as length 0).  The two scripts import RDKit-based helpers that are not installed here and are not on
this path (`calculate_property`, the dataset class, `Chem`): those names are stubbed so that the modules import; nothing of them runs.
The tokenizer handed to `evaluate` is a stand-in that renders ids as decimal strings, so that the hypotheses come back as token ids.
The fixture is data only (inputs + the best hypothesis per molecule); nothing here runs on the GPU box."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import spmm_oracle as O  # noqa: E402
import make_golden as MG  # noqa: E402  (ref_model: the reference module with closed-form weights)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def peaky_bias(seed=10, sep_gap=0.5, vocab=300):
    """tests/test_step_gpu.py::_peaky_lm: well separated next-token distributions, [SEP] close to the top."""
    g = torch.Generator().manual_seed(seed)
    b = torch.randn(vocab, generator=g) * 1.5
    b[3] = b.max() - sep_gap
    return b


def _stub_modules():
    for name in ("rdkit", "rdkit.Chem", "rdkit.Chem.Descriptors", "rdkit.RDLogger", "calc_property", "dataset", "tqdm"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    sys.modules["rdkit"].RDLogger = sys.modules["rdkit.RDLogger"]
    sys.modules["calc_property"].calculate_property = None
    sys.modules["dataset"].SMILESDataset_pretrain = None
    sys.modules["tqdm"].tqdm = lambda it, *a, **k: it


class IdTokenizer:
    cls_token_id, sep_token_id = 2, 3

    @staticmethod
    def convert_ids_to_tokens(ids):
        return [str(int(i)) for i in ids]

    @staticmethod
    def convert_tokens_to_string(tokens):
        return " ".join(tokens)


# (bias seed, [SEP] gap) per molecule
CASES = [(0, 0.6), (0, 1.2), (2, 0.6), (2, 0.9), (4, 0.6), (4, 0.9), (5, 0.6), (5, 0.9), (7, 0.9), (8, 0.6), (10, 0.6), (10, 1.2), (11, 0.9),
         (1, 0.6), (9, 0.9), (0, 1.05), (7, 1.05), (11, 1.05)]


if __name__ == "__main__":
    import contextlib
    import io
    ref_shim._install()
    _stub_modules()
    import d_pv2smiles_batched as ref_decode          # /root/reference (on sys.path through ref_shim)
    cfg = O.tiny_cfg()
    m = MG.ref_model(cfg, dropout=0.0)
    N, k = len(CASES), 5
    props = torch.randn(N, 53, generator=torch.Generator().manual_seed(12)) * 2
    best = []
    for n, (seed, gap) in enumerate(CASES):
        b = peaky_bias(seed, gap)
        with torch.no_grad():
            m.text_encoder.cls.predictions.bias.copy_(b)
            m.text_encoder.cls.predictions.decoder.bias.copy_(b)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                _, cand = ref_decode.evaluate(m, [(props[n:n + 1], ["[CLS]"])], IdTokenizer(), "cpu", stochastic=False, k=k)
            best.append([int(t) for t in cand[0].split()])      # ids of the best hypothesis without its final [SEP] (the reference strips it)
        except IndexError:
            best.append([])                                     # nothing finished in 100 steps
    L = max(len(x) for x in best)
    arr = np.zeros((N, L), dtype=np.int64)
    for n, x in enumerate(best):
        arr[n, :len(x)] = x
    np.savez_compressed(os.path.join(OUT, "decode_tiny_k5"), props=props.numpy(), k=np.int64(k), bias_seed=np.array([c[0] for c in CASES]),
                        sep_gap=np.array([c[1] for c in CASES]), best_ids=arr, best_len=np.array([len(x) for x in best], dtype=np.int64))
    print("decode_tiny_k5:", [len(x) for x in best])
