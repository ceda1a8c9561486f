"""Host logic of the generation pipeline (hoigen_amd/generation.py; main_tip_finetune.py:759-824) that needs no GPU: how many loop
iterations go through the kernels as one step (the text tower's equal passes of at most 65 536 rows should come out full), and the
HOI -> object index table of the human / object branches (main_tip_finetune.py:772-779: target = HOI_IDX_TO_OBJ_IDX)."""
import json
import os
import types

import torch

from hoigen_amd.generation import FeatureSampler

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sampler(n_per_branch, tokens_run):
    s = object.__new__(FeatureSampler)
    s.branches = {k: types.SimpleNamespace(target=torch.zeros(n, dtype=torch.long)) for k, n in n_per_branch.items()}
    s._tokens_run = lambda: tokens_run
    return s


def test_auto_batch_fills_the_text_passes():
    s = _sampler({"hoi": 600, "human": 600, "object": 600}, 14)      # HICO: 1 800 prompts at 14 executed tokens
    k = s._auto_batch(100)
    rows = k * 1800 * 14
    passes = -(-rows // FeatureSampler.TEXT_PASS_ROWS)
    assert k == 13 and passes == 5 and rows / (passes * FeatureSampler.TEXT_PASS_ROWS) > 0.999
    assert s._auto_batch(5) <= 5                                      # never more than the iterations asked for
    assert s._auto_batch(1) == 1
    # a loop whose single iteration already overflows a pass still makes progress
    big = _sampler({"hoi": 6000}, 77)
    assert big._auto_batch(100) >= 1
    # the choice is the best fill among 1 .. 16, first best wins
    for n, lt in ((1800, 13), (1800, 16), (681, 77), (117, 20)):
        s2 = _sampler({"a": n}, lt)
        k2 = s2._auto_batch(100)
        fill = lambda k: (k * n * lt) / (-(-(k * n * lt) // s2.TEXT_PASS_ROWS) * s2.TEXT_PASS_ROWS)
        assert 1 <= k2 <= 16 and all(fill(k2) >= fill(j) - 1e-9 for j in range(1, 17))


def test_hoi_to_object_table_of_the_hico_names():
    """hico_sampler() maps every HOI name to the object whose name it ends with (longest match): checked on the reference's own 600 +
    80 names (tests/golden/g0_tokens.json) without building a model - the same code path as generation.hico_sampler's obj_of."""
    names = json.load(open(os.path.join(G, "g0_tokens.json")))["_classnames"]
    hoi, obj = list(names["hoi"]), list(names["obj"])
    by_len = sorted(range(len(obj)), key=lambda i: -len(obj[i]))

    def obj_of(name):
        nm = name.replace("_", " ")
        for i in by_len:
            if nm.endswith(obj[i].replace("_", " ")):
                return i
        return None

    table = [obj_of(n) for n in hoi]
    assert len(hoi) == 600 and len(obj) == 80 and None not in table
    assert len(set(table)) == 80                                      # every object has at least one HOI
    for n, i in zip(hoi, table):
        assert n.replace("_", " ").endswith(obj[i].replace("_", " "))
    # "hot dog" / "dog", "wine glass" / ... : the longest name wins
    for n, i in zip(hoi, table):
        longer = [j for j in range(80) if j != i and n.replace("_", " ").endswith(obj[j].replace("_", " "))]
        assert all(len(obj[j]) <= len(obj[i]) for j in longer)
