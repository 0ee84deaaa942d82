"""SMILES WordPiece tokenizer: what `BertTokenizer(vocab_file, do_lower_case=False,
do_basic_tokenize=False)` with `WordpieceTokenizer(max_input_chars_per_word=250)` does for the reference
(SPMM_pretrain.py:19-20, d_smiles2pv.py:126-127) -- without the `transformers` dependency.

The reference feeds '[CLS]' + smiles: with basic tokenisation off, the whole string is one "word", so greedy
longest-match takes the piece '[CLS]' first and every later piece carries the '##' continuation prefix (the 300-piece
vocabulary holds the SMILES fragments only in their '##' form).  The tokenizer then adds its own [CLS] in front and [SEP]
behind; the model drops that first column (SPMM_models.py:357), so the ids the encoder sees are
[CLS] piece_1 ... piece_n [SEP] PAD...

Only what the path uses is implemented: batch call with padding='longest', truncation to max_length, `.input_ids` /
`.attention_mask` int64 tensors, `decode` for generated ids."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Iterable, List, Sequence, Union

import torch

PAD, UNK, CLS, SEP = "[PAD]", "[UNK]", "[CLS]", "[SEP]"


class SmilesWordPiece:
    def __init__(self, vocab: Union[str, Sequence[str]], max_input_chars_per_word: int = 250):
        if isinstance(vocab, str):
            with open(vocab, encoding="utf-8") as f:
                vocab = [line.rstrip("\n") for line in f]
        self.itos: List[str] = list(vocab)
        self.vocab: Dict[str, int] = {t: i for i, t in enumerate(self.itos)}
        for t in (PAD, UNK, CLS, SEP):
            if t not in self.vocab:
                raise ValueError(f"vocabulary lacks {t}")
        self.pad_token_id, self.unk_token_id = self.vocab[PAD], self.vocab[UNK]
        self.cls_token_id, self.sep_token_id = self.vocab[CLS], self.vocab[SEP]
        self.max_chars = max_input_chars_per_word
        self._longest = max(len(t) for t in self.itos)

    # -- greedy longest-match-first over one whitespace-delimited word ------------------------------------------------
    def _word(self, word: str) -> List[str]:
        if len(word) > self.max_chars:
            return [UNK]
        out, start, n = [], 0, len(word)
        while start < n:
            end = min(n, start + self._longest)          # no piece is longer than the longest vocabulary entry
            piece = None
            while end > start:
                cand = word[start:end] if start == 0 else "##" + word[start:end]
                if cand in self.vocab:
                    piece = cand
                    break
                end -= 1
            if piece is None:
                return [UNK]                             # one unmatched position poisons the whole word
            out.append(piece)
            start = end
        return out

    def tokenize(self, text: str) -> List[str]:
        return [p for w in text.split() for p in self._word(w)]

    def convert_tokens_to_ids(self, tokens: Iterable[str]) -> List[int]:
        return [self.vocab.get(t, self.unk_token_id) for t in tokens]

    def encode(self, text: str, max_length: int | None = None) -> List[int]:
        ids = self.convert_tokens_to_ids(self.tokenize(text))
        if max_length is not None:
            ids = ids[: max(0, max_length - 2)]          # truncation keeps room for the two special tokens
        return [self.cls_token_id] + ids + [self.sep_token_id]

    def __call__(self, texts: Union[str, Sequence[str]], padding="longest", truncation=True, max_length: int | None = 100,
                 return_tensors="pt"):
        if isinstance(texts, str):
            texts = [texts]
        if padding not in ("longest", True) or return_tensors != "pt":
            raise ValueError("SmilesWordPiece supports padding='longest', return_tensors='pt' (what the pretraining path uses)")
        rows = [self.encode(t, max_length if truncation else None) for t in texts]
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = torch.tensor(r, dtype=torch.long)
        out = SimpleNamespace(input_ids=ids, attention_mask=(ids != self.pad_token_id).long())
        out.to = lambda dev: SimpleNamespace(input_ids=ids.to(dev), attention_mask=out.attention_mask.to(dev))
        return out

    def decode(self, ids: Iterable[int], skip_special_tokens: bool = True) -> str:
        """Generated ids -> SMILES text (d_pv2smiles_batched.py:61-66 strips the specials and the '##' markers)."""
        parts = []
        for i in ids:
            t = self.itos[int(i)]
            if skip_special_tokens and t in (PAD, UNK, CLS, SEP):
                continue
            parts.append(t[2:] if t.startswith("##") else t)
        return "".join(parts)
