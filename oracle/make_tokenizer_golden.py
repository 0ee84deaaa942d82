"""Generates tests/golden/tokenizer_vocab300.npz (dev box only: reads the reference's vocabulary and uses the
`transformers` WordpieceTokenizer the reference wires in, SPMM_pretrain.py:19-20).  Test infrastructure.

Expected ids follow what BertTokenizer(do_basic_tokenize=False) + that WordpieceTokenizer produce for '[CLS]'+smiles with
padding='longest', truncation=True, max_length=100: [CLS] + pieces[:98] + [SEP].  (Under transformers 5 BertTokenizer itself
is tokenizers-backed and no longer honours the swapped-in wordpiece object -- SURVEY.md section 8c -- so the sequence is
assembled here from the WordpieceTokenizer output, which is unchanged since 4.30.)"""
import os
import numpy as np
from transformers import WordpieceTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
vocab_list = [l.rstrip("\n") for l in open("/root/reference/vocab_bpe_300.txt", encoding="utf-8")]
vocab = {t: i for i, t in enumerate(vocab_list)}
wp = WordpieceTokenizer(vocab=vocab, unk_token="[UNK]", max_input_chars_per_word=250)
smiles = [
    "CC(=O)OC1=CC=CC=C1C(=O)O",                                  # aspirin
    "CN1C=NC2=C1C(=O)N(C(=O)N2C)C",                              # caffeine
    "CC(C)CC1=CC=C(C=C1)C(C)C(=O)O",                             # ibuprofen
    "C1=CC=C(C=C1)C=O", "CCO", "C", "O=C=O", "[Na+].[Cl-]", "C[C@H](N)C(=O)O", "N#Cc1ccccc1", "c1ccc2c(c1)[nH]c1ccccc12",
    "CC(C)(C)c1ccc(O)cc1", "FC(F)(F)c1ccc(Cl)cc1Br", "C1CCC(CC1)N2CCN(CC2)C(=O)c3ccc(I)cc3", "OS(=O)(=O)O",
    "[Na+].O=S.Cl." * 18,                                        # 163 pieces in 234 characters: exercises truncation to 98
    "C" * 251,                                                   # longer than max_input_chars_per_word: one [UNK]
    "C?C", "",                                                   # an out-of-vocabulary character poisons the word; empty text
    "CC O",                                                      # whitespace splits words: second word has no '##' form
]
rows = []
for s in smiles:
    ids = [vocab.get(t, vocab["[UNK]"]) for t in wp.tokenize("[CLS]" + s)][:98]
    rows.append([vocab["[CLS]"]] + ids + [vocab["[SEP]"]])
L = max(len(r) for r in rows)
ids = np.zeros((len(rows), L), dtype=np.int64)
for i, r in enumerate(rows):
    ids[i, : len(r)] = r
np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "tokenizer_vocab300.npz"), vocab=np.array(vocab_list), smiles=np.array(smiles),
                    input_ids=ids, attention_mask=(ids != 0).astype(np.int64))
print(ids.shape, ids[:3, :12])
