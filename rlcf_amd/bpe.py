"""Byte-level BPE tokenizer with the CLIP vocabulary contract (reference: `clip.tokenize`, TPT/clip/clip.py:197-233,
and `SimpleTokenizer`, TPT/clip/simple_tokenizer.py:62-132).  Written from the published algorithm; the merges file
(`bpe_simple_vocab_16e6.txt.gz`, OpenAI's data) is NOT shipped here — pass its path.  Host side, once per dataset.

    tok = ClipBPE("/path/to/bpe_simple_vocab_16e6.txt.gz")
    rlcf_amd.clip_store.set_tokenizer(tok.tokenize)
"""
from __future__ import annotations

import gzip
import html
from functools import lru_cache
from typing import Dict, List, Tuple, Union

import regex as re
import torch

N_MERGES = 49152 - 256 - 2          # merges kept from the file (vocabulary 49408 = 256*2 + merges + 2 specials)


def _byte_alphabet() -> Dict[int, str]:
    """Reversible byte -> printable unicode map: printable latin-1 bytes map to themselves, the rest to 256+."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _clean(text: str) -> str:
    try:                                   # the reference runs ftfy.fix_text first; identity for ASCII class names
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class ClipBPE:
    def __init__(self, bpe_path: str):
        self.byte_enc = _byte_alphabet()
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")
        merges: List[Tuple[str, str]] = [tuple(m.split()) for m in lines[1:N_MERGES + 1]]
        alphabet = list(self.byte_enc.values())          # NB: dict order = byte order 0..255
        # the published vocabulary orders the 256 symbols as: the printable bytes first (in byte order), then the remapped ones
        printable = [c for b, c in self.byte_enc.items() if ord(c) < 256]
        remapped = [c for b, c in self.byte_enc.items() if ord(c) >= 256]
        alphabet = printable + remapped
        vocab = alphabet + [c + "</w>" for c in alphabet] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                              re.IGNORECASE)

    @lru_cache(maxsize=65536)
    def _bpe(self, token: str) -> Tuple[str, ...]:
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        return word

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for tok in re.findall(self.pat, _clean(text).lower()):
            if tok in ("<|startoftext|>", "<|endoftext|>"):
                ids.append(self.encoder[tok])
                continue
            sym = "".join(self.byte_enc[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self._bpe(sym))
        return ids

    def tokenize(self, texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.Tensor:
        """-> int64 [n, context_length]: [SOT] + ids + [EOT], zero padded (clip.py:197-233)."""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids)
        return out
