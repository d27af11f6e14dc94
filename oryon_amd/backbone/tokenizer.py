"""Byte-level BPE tokenizer with CLIP's conventions (lower-cased text, '</w>' word-end marker, <|startoftext|> /
<|endoftext|> specials, context 77) - the role of models/tokenizer.py:64-151 in the reference, which is itself a copy
of the published openai/CLIP tokenizer.  The merge table (`bpe_simple_vocab_16e6.txt.gz`) is not part of the
reference tree; give its path to use real prompts.  Synthetic runs feed token ids directly (`Oryon.forward` accepts a
LongTensor [B, 80, 77] under xs['prompt_tokens']).
Pinned to the reference's tokenizer on a fabricated merge table (tests/golden/g10_tokenizer.npz, tests/test_backbone.py): same ids
for templates, punctuation, HTML entities, blanks, case, non-ASCII bytes and over-long prompts (plain truncation to the context
length WITHOUT re-inserting the end token, models/tokenizer.py:141-143).  One step of the reference is not reproduced: `ftfy.fix_text`
(models/tokenizer.py:52, mojibake repair; the package is not available here) - a no-op on the datasets' plain-ASCII object names
and templates, a divergence on text with broken encodings."""
from __future__ import annotations

import gzip
import html
from functools import lru_cache
from typing import Dict, List, Tuple

import regex as re
import torch


@lru_cache()
def _byte_table() -> Dict[int, str]:
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, extra = {b: chr(b) for b in keep}, 0
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class SimpleTokenizer:
    def __init__(self, bpe_path: str, context_length: int = 77):
        self.context_length = context_length
        self.byte_enc = _byte_table()
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges: List[Tuple[str, str]] = [tuple(m.split()) for m in lines[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_enc.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                              re.IGNORECASE)

    def _bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            ranked = [(self.ranks[(a, b)], (a, b)) for a, b in zip(word, word[1:]) if (a, b) in self.ranks]
            if not ranked:
                break
            best = min(ranked)[1]
            # merge every occurrence of the best-ranked pair, left to right
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        text = re.sub(r"\s+", " ", html.unescape(html.unescape(text))).strip().lower()
        ids: List[int] = []
        for tok in re.findall(self.pat, text):
            tok = "".join(self.byte_enc[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._bpe(tok).split(" "))
        return ids

    def __call__(self, texts, context_length: int = None) -> torch.Tensor:
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        sot, eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        out = torch.zeros(len(texts), L, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = ([sot] + self.encode(t) + [eot])[:L]          # the reference truncates and does not restore the end token
            out[i, :len(ids)] = torch.tensor(ids)
        return out
