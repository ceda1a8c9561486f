#!/usr/bin/env python3
"""g9_cache_keys.npz: the reference's own ``build_clip_cache_model`` (utils.py:6-61, executed from the reference tree
in the build container) on seeded inputs: a stand-in loader that yields the crops' targets and a stand-in
``clip_model.image_encoder`` that returns prescribed un-normalised features (the encoder itself is pinned by g2).
Stored: the features, the per-crop verb lists and the reference's (cache_keys, cache_values) for seed 1234.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cache_keys.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from hoigen_amd import synth  # noqa: E402

REF = "/root/reference/utils.py"
N, D, C, SHOT, BATCH = 90, 512, 24, 2, 8


def inputs():
    rng = np.random.RandomState(77)
    feats = synth.hg_normal((N, D), 900, 1.0)
    verbs = []
    for i in range(N):
        k = 1 + (rng.rand() < 0.3)
        v = sorted(set(int(x) for x in rng.randint(0, C - 3, size=k)))     # the last 3 classes stay empty
        verbs.append(v)
    return feats, verbs


def main():
    src = open(REF).read().split("\n")
    ns = {"torch": torch, "os": os, "tqdm": lambda x: x}
    exec(compile("\n".join(src[5:94]), REF, "exec"), ns)          # def build_clip_cache_model (lines 6-94)
    feats, verbs = inputs()
    ft = torch.from_numpy(feats)

    class Enc:
        def __init__(self):
            self.pos = 0

        def __call__(self, images):
            b = images.shape[0]
            out = ft[self.pos:self.pos + b]
            self.pos += b
            return out, torch.zeros(b, 1)

    clip_model = types.SimpleNamespace(image_encoder=Enc())
    clip_model.cuda = lambda: clip_model
    loader = []
    for b0 in range(0, N, BATCH):
        idx = range(b0, min(N, b0 + BATCH))
        images = [(None, torch.zeros(3, 2, 2)) for _ in idx]
        target = [{"verb": torch.tensor(verbs[i])} for i in idx]
        loader.append((images, target))
    args = types.SimpleNamespace(clip_load_cache=False, num_classes=C, num_shot=SHOT, dataset="hicodet", zs=False, zs_type="x")
    torch.Tensor.cuda = lambda self, *a, **k: self                 # the build container has no GPU
    saved = {}
    torch_save = torch.save
    torch.save = lambda obj, path: saved.__setitem__(os.path.basename(path), obj)
    cwd = os.getcwd()
    os.chdir("/tmp")                                               # the reference creates ./caches/dataset
    torch.manual_seed(1234)
    keys, values = ns["build_clip_cache_model"](args, clip_model, loader)
    os.chdir(cwd)
    torch.save = torch_save
    np.savez_compressed(f"{HERE}/g9_cache_keys.npz", features=feats,
                        verbs=np.array([v + [-1] * (2 - len(v)) for v in verbs], np.int32),
                        cache_keys=keys.numpy(), cache_values=values.numpy(), seed=np.int64(1234),
                        num_classes=np.int64(C), num_shot=np.int64(SHOT))
    print("g9:", keys.shape, values.shape)


if __name__ == "__main__":
    main()
