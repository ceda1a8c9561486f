"""Byte-level BPE tokenizer for CLIP prompts (host side, pure integer/string work).

Drop-in for ``SimpleTokenizer`` of the reference (/root/reference/clipnet/simple_tokenizer.py:62-132,
identical twin in CLIP/clip/simple_tokenizer.py): same vocabulary (49 408 entries: 256 byte
symbols, 256 end-of-word byte symbols, 48 894 merges, ``<|startoftext|>`` = 49406,
``<|endoftext|>`` = 49407), same ``encode`` / ``decode`` / ``encoder`` / ``decoder`` surface.
Token ids must be bit-exact: ``tests/test_tokenizer.py`` checks them against ids produced by the
reference for ~1 650 prompts (tests/golden/g0_tokens.json).

The merge table is data (``data/clip_bpe_merges.txt.xz``, re-packed from the public CLIP vocabulary by
``data/make_bpe_table.py``).
"""
from __future__ import annotations

import html
import lzma
import os
from functools import lru_cache
from typing import Dict, List, Tuple

import regex as re

try:  # the reference requires ftfy (simple_tokenizer.py:6,50); it only matters for mojibake input
    import ftfy as _ftfy

    def _fix_text(s: str) -> str:
        return _ftfy.fix_text(s)
except ImportError:  # pragma: no cover - ftfy is absent from the build image
    def _fix_text(s: str) -> str:
        return s

SOT_TEXT = "<|startoftext|>"
EOT_TEXT = "<|endoftext|>"
_EOW = "</w>"


def default_bpe() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "clip_bpe_merges.txt.xz")


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """Reversible byte -> printable unicode map (reference simple_tokenizer.py:15-35): printable
    latin-1 bytes map to themselves, the other 68 bytes to code points 256.. in byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    order = sorted(keep, key=lambda b: (0 if 0x21 <= b <= 0x7E else 1 if 0xA1 <= b <= 0xAC else 2, b))
    table = {b: chr(b) for b in order}
    extra = 0
    for b in range(256):
        if b not in keep:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _read_merges(path: str) -> List[Tuple[str, str]]:
    opener = lzma.open if path.endswith(".xz") else open
    with opener(path, "rt", encoding="utf-8") as f:
        rows = f.read().split("\n")
    return [tuple(r.split()) for r in rows if r]  # type: ignore[misc]


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None):
        merges = _read_merges(bpe_path or default_bpe())
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {c: b for b, c in self.byte_encoder.items()}
        symbols = list(self.byte_encoder.values())
        vocab = symbols + [s + _EOW for s in symbols] + [a + b for a, b in merges] + [SOT_TEXT, EOT_TEXT]
        self.encoder: Dict[str, int] = {tok: i for i, tok in enumerate(vocab)}
        self.decoder: Dict[int, str] = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks: Dict[Tuple[str, str], int] = {m: i for i, m in enumerate(merges)}
        self.cache: Dict[str, str] = {SOT_TEXT: SOT_TEXT, EOT_TEXT: EOT_TEXT}
        self.pat = re.compile(
            r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
            re.IGNORECASE)

    # -- BPE: repeatedly fuse the adjacent pair of lowest merge rank, all occurrences left to right
    def bpe(self, token: str) -> str:
        hit = self.cache.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + _EOW]
        if len(parts) == 1:
            return token + _EOW
        ranks = self.bpe_ranks
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = ranks.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            a, b = best
            fused, i, n = [], 0, len(parts)
            while i < n:
                if i + 1 < n and parts[i] == a and parts[i + 1] == b:
                    fused.append(a + b)
                    i += 2
                else:
                    fused.append(parts[i])
                    i += 1
            parts = fused
        out = " ".join(parts)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        text = html.unescape(html.unescape(_fix_text(text))).strip()
        text = re.sub(r"\s+", " ", text).strip().lower()
        ids: List[int] = []
        for tok in re.findall(self.pat, text):
            sym = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[p] for p in self.bpe(sym).split(" "))
        return ids

    def decode(self, tokens) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        raw = bytearray(self.byte_decoder[c] for c in text)
        return raw.decode("utf-8", errors="replace").replace(_EOW, " ")
