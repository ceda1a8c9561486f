#!/usr/bin/env python3
"""Re-pack the CLIP BPE merge table (a data table, not code) into ``clip_bpe_merges.txt.xz``.

Source of the table: the OpenAI CLIP vocabulary file the reference ships as
/root/reference/clipnet/bpe_simple_vocab_16e6.txt.gz (read at clipnet/simple_tokenizer.py:66-67:
line 0 is a header, lines 1..48894 are the merges actually used).  Only those 48 894 merge lines
are kept, one ``left right`` pair per line, LZMA-compressed.  Run in the build container only.
"""
import gzip
import lzma
import os

SRC = "/root/reference/clipnet/bpe_simple_vocab_16e6.txt.gz"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_bpe_merges.txt.xz")

lines = gzip.open(SRC).read().decode("utf-8").split("\n")
merges = lines[1:49152 - 256 - 2 + 1]
assert len(merges) == 48894 and all(len(m.split()) == 2 for m in merges)
with lzma.open(DST, "wt", encoding="utf-8", preset=9) as f:
    f.write("\n".join(merges))
print(DST, os.path.getsize(DST), "bytes,", len(merges), "merges")
