"""Token ids must be bit-exact with the reference tokenizer (fixture g0: ids produced by
clipnet.tokenize / CLIP_models_adapter_prior2.tokenize in the build container)."""
import json
import os

import pytest
import torch

from hoigen_amd import clip

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g0():
    return json.load(open(f"{G}/g0_tokens.json"))


def test_all_prompt_groups_bit_exact(g0):
    n = 0
    for name, grp in g0.items():
        if name.startswith("_"):
            continue
        ids = clip.tokenize(grp["text"])
        assert ids.dtype == torch.int64 and ids.shape == (len(grp["text"]), 77)
        for row, want, eot in zip(ids, grp["ids"], grp["eot"]):
            assert row[: len(want)].tolist() == want
            assert int(row[len(want):].abs().sum()) == 0
            assert int(row.argmax()) == eot            # EOT index used by encode_text (model.py:350)
            n += 1
    assert n >= 1600


def test_int32_variant_matches(g0):
    ids64 = clip.tokenize(g0["hoi600"]["text"][:20])
    ids32 = clip.tokenize(g0["hoi600"]["text"][:20], dtype=torch.int32)   # adapter-variant tokenize dtype
    assert ids32.dtype == torch.int32 and torch.equal(ids64, ids32.long())


def test_known_answer():
    ids = clip.tokenize("a photo of a person riding a bicycle")[0]
    assert ids[:10].tolist() == [49406, 320, 1125, 539, 320, 2533, 6765, 320, 11652, 49407]


def test_too_long_raises_and_truncate(g0):
    s = g0["_truncate"]["text"]
    with pytest.raises(RuntimeError):
        clip.tokenize(s)
    assert clip.tokenize(s, truncate=True)[0].tolist() == g0["_truncate"]["ids"]


def test_str_and_context_length():
    assert clip.tokenize("hello").shape == (1, 77)
    assert clip.tokenize(["hello", "world"], context_length=16).shape == (2, 16)


def test_decode_roundtrip():
    tk = clip._tokenizer
    s = "a photo of a person holding a hair drier"
    assert tk.decode(tk.encode(s)).strip() == s
