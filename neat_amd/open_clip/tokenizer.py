"""CLIP byte-pair tokenizer with the interface of the reference's ``SimpleTokenizer`` (src/open_clip/tokenizer.py:127-259).

Own implementation of the published algorithm (OpenAI CLIP BPE): text cleaning -> regex pre-tokenisation -> UTF-8
bytes mapped to printable code points -> greedy lowest-rank pair merging -> ids; ``__call__`` adds SOT (49406) / EOT
(49407), truncates to ``context_length`` keeping EOT last, pads with 0.

The merge table ``bpe_simple_vocab_16e6.txt.gz`` is an external data asset (OpenAI CLIP / open_clip ship it); it is NOT
vendored here.  It is looked up in: the ``bpe_path`` argument, ``$NEAT_BPE_VOCAB``, an installed ``open_clip`` package,
a reference checkout next to this repository.  ``ftfy`` is used when importable (the reference requires it); without
it text is passed through unchanged, which is identical for text that needs no mojibake repair.
Host-side integer work, not accelerated: a batch of 256 captions tokenises in milliseconds.
"""
from __future__ import annotations

import gzip
from itertools import chain as _chain_cls
import html
import os
import re as _re
from typing import List, Optional, Union

import numpy as np
import torch

_chain = _chain_cls.from_iterable
DEFAULT_CONTEXT_LENGTH = 77
N_MERGES = 49152 - 256 - 2

_CANDIDATES = (
    "/root/reference/src/open_clip/bpe_simple_vocab_16e6.txt.gz",
)


def default_bpe() -> str:
    env = os.environ.get("NEAT_BPE_VOCAB")
    if env:
        return env
    try:
        import importlib.util
        spec = importlib.util.find_spec("open_clip")
        if spec and spec.submodule_search_locations:
            p = os.path.join(list(spec.submodule_search_locations)[0], "bpe_simple_vocab_16e6.txt.gz")
            if os.path.exists(p):
                return p
    except Exception:
        pass
    for p in _CANDIDATES:
        if os.path.exists(p):
            return p
    raise FileNotFoundError("CLIP BPE merge table bpe_simple_vocab_16e6.txt.gz not found: pass bpe_path= or set "
                            "NEAT_BPE_VOCAB (the file ships with OpenAI CLIP / open_clip)")


def byte_symbols() -> List[str]:
    """symbol[b] for every byte value: printable Latin-1 bytes stand for themselves, the rest take 256, 257, ..."""
    keep = set(range(ord("!"), ord("~") + 1)) | set(range(0xA1, 0xAC + 1)) | set(range(0xAE, 0xFF + 1))
    out, nxt = [], 256
    for b in range(256):
        if b in keep:
            out.append(chr(b))
        else:
            out.append(chr(nxt))
            nxt += 1
    return out


try:                                    # looked up ONCE: a failing import inside the per-caption path costs ~0.1 ms each
    import ftfy as _ftfy
except ImportError:
    _ftfy = None


def _clean_lower(text: str) -> str:
    if _ftfy is not None:
        text = _ftfy.fix_text(text)
    if "&" in text:                     # html.unescape only ever changes text that holds an entity
        text = html.unescape(html.unescape(text))
    text = text.strip()
    return _re.sub(r"\s+", " ", text).strip().lower()


class SimpleTokenizer:
    def __init__(self, bpe_path: Optional[str] = None, additional_special_tokens: Optional[List[str]] = None,
                 context_length: Optional[int] = DEFAULT_CONTEXT_LENGTH, clean: str = "lower"):
        import regex
        if clean != "lower":
            raise NotImplementedError("SimpleTokenizer: only clean='lower' (the CLIP default) is built")
        with gzip.open(bpe_path or default_bpe()) as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(ln.split()) for ln in lines[1:N_MERGES + 1]]
        self.sym = byte_symbols()
        # vocabulary order: bytes in the order printable-first then the remapped ones, the same with '</w>', the merges
        order = sorted(range(256), key=lambda b: (ord(self.sym[b]) >= 256, ord(self.sym[b])))
        base = [self.sym[b] for b in order]
        vocab = base + [s + "</w>" for s in base] + ["".join(m) for m in merges]
        special = ["<start_of_text>", "<end_of_text>"] + list(additional_special_tokens or [])
        vocab += special
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.special = set(special)
        self.pat = regex.compile("|".join(regex.escape(s) for s in special) +
                                 r"""|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
        self.vocab_size = len(vocab)
        self.all_special_ids = [self.encoder[t] for t in special]
        self.sot_token_id, self.eot_token_id = self.all_special_ids[0], self.all_special_ids[1]
        self.context_length = context_length
        self._cache = {}
        self._ids = {}                  # pre-token (as matched by the regex) -> list of vocabulary ids
        self._words = {}                # whitespace-delimited lower-cased ASCII word -> list of vocabulary ids

    def _merge(self, token: str) -> List[str]:
        """Greedy BPE of one pre-token (already in byte symbols): repeatedly fuse every occurrence of the adjacent
        pair with the lowest merge rank."""
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        if token in self.special:
            out = [token]
        else:
            parts = list(token[:-1]) + [token[-1] + "</w>"]
            while len(parts) > 1:
                best, best_rank = None, None
                for a, b in zip(parts, parts[1:]):
                    r = self.rank.get((a, b))
                    if r is not None and (best_rank is None or r < best_rank):
                        best, best_rank = (a, b), r
                if best is None:
                    break
                fused, i = [], 0
                while i < len(parts):
                    if i + 1 < len(parts) and parts[i] == best[0] and parts[i + 1] == best[1]:
                        fused.append(parts[i] + parts[i + 1])
                        i += 2
                    else:
                        fused.append(parts[i])
                        i += 1
                parts = fused
            out = parts
        self._cache[token] = out
        return out

    def _encode_slow(self, text: str) -> List[int]:
        ids = []
        known = self._ids
        for tok in self.pat.findall(_clean_lower(text)):
            hit = known.get(tok)
            if hit is None:
                sym = "".join(self.sym[b] for b in tok.encode("utf-8"))
                hit = known[tok] = [self.encoder[p] for p in self._merge(sym)]
            ids.extend(hit)
        return ids

    def _word_ids(self, word: str):
        """ids of one whitespace-delimited, lower-cased word, or None when the word needs the full cleaning path (anything
        but printable ASCII, or a character that html.unescape / the special-token alternatives could act on)."""
        if not (word.isascii() and word.isprintable()) or "&" in word or "<" in word:
            return None
        return self._encode_slow(word)

    def encode(self, text: str) -> List[int]:
        """Fast path for plain captions: cleaning (whitespace runs -> one space, lower-casing) and the pre-tokenising regex
        never look across whitespace, and on printable ASCII without '&' / '<' neither ftfy nor html.unescape nor the
        special-token alternatives change anything -- so a caption is the concatenation of its words' ids, and words come
        from a cache (no regex call at all for a known word: 16.7 -> 4 ms per 4 x 256 captions).  Any other word sends
        the whole caption through the full path."""
        cache = self._words
        words = text.lower().split()
        try:                                              # every word known: the whole caption in C-level loops
            return list(_chain(map(cache.__getitem__, words)))
        except KeyError:
            pass
        ids = []
        for w in words:
            hit = cache.get(w)
            if hit is None:
                hit = self._word_ids(w)
                if hit is None:
                    return self._encode_slow(text)
                cache[w] = hit
            ids.extend(hit)
        return ids

    def decode(self, tokens) -> str:
        inv = {s: b for b, s in enumerate(self.sym)}
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytes(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")

    def __call__(self, texts: Union[str, List[str]], context_length: Optional[int] = None) -> torch.LongTensor:
        if isinstance(texts, str):
            texts = [texts]
        n = context_length or self.context_length
        if not n:
            raise ValueError("SimpleTokenizer: context_length must be set")
        # one array for the batch, filled by ONE scatter: ids of all captions back to back + their (row, column) positions
        sot, eot = self.sot_token_id, self.eot_token_id
        rows_ids = []
        for text in texts:
            ids = self.encode(text)
            if len(ids) > n - 2:                          # truncation keeps EOT last (reference tokenizer.py:250-259)
                ids = ids[:n - 2]
            rows_ids.append(ids)
        lens = np.fromiter(map(len, rows_ids), dtype=np.int64, count=len(rows_ids))
        out = np.zeros((len(texts), n), dtype=np.int64)
        out[:, 0] = sot
        out[np.arange(len(texts)), lens + 1] = eot
        total = int(lens.sum())
        if total:
            flat = np.fromiter(_chain(rows_ids), dtype=np.int64, count=total)
            starts = np.cumsum(lens) - lens
            rows = np.repeat(np.arange(len(texts)), lens)
            out[rows, np.arange(total) - np.repeat(starts, lens) + 1] = flat
        return torch.from_numpy(out)


def get_tokenizer(model_name: str = "", context_length: Optional[int] = None, **kwargs):
    """Reference factory.get_tokenizer(model_name) for the CLIP ViT configs (all use the simple BPE, context 77)."""
    return SimpleTokenizer(context_length=context_length or DEFAULT_CONTEXT_LENGTH, **kwargs)
