"""Seeded inputs of the cache-model fixtures (shared by make_golden_cache.py and tests/test_cache_model.py):
only the reference's OUTPUTS are stored in g7_cache.npz, the inputs are regenerated from these seeds with the
portable integer-hash generator of hoigen_amd.synth."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hoigen_amd import synth  # noqa: E402

CASES = [(15, 229, 117, 512), (200, 300, 600, 512)]          # (R pairs, S cached samples, C classes, D)


def _unit(n, d, seed):
    x = synth.hg_normal((n, d), seed, 1.0).astype(np.float64)
    return (x / np.sqrt((x * x).sum(1, keepdims=True))).astype(np.float32)


def cache_case(case):
    R, S, C, D = CASES[case]
    rng = np.random.RandomState(1000 + case)
    cls = rng.randint(0, C, size=S)
    lab = np.zeros((S, C), np.float32)
    lab[np.arange(S), cls] = 1
    lab[rng.rand(S, C) < 0.01] = 1                               # multi-hot labels
    lens = np.maximum(lab.sum(0), 1).astype(np.float32)
    s = 50 * case
    return dict(human=_unit(R, D, s + 1), object=_unit(R, D, s + 2), union=_unit(R, D, s + 3),
                w_ho=np.concatenate([_unit(S, D, s + 4), _unit(S, D, s + 5)], 1), w_u=_unit(S, D, s + 6),
                w_text=_unit(C, D, s + 7) * np.float32(3.0),
                b_ho=(-1.0 + 0.1 * synth.hg_normal((S,), s + 8, 1.0)).astype(np.float32),
                b_u=-np.ones(S, np.float32), label=lab, lens=lens)
