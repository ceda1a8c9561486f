"""A call's result must not depend on what the context did before it (workspaces are grow-only and reused, pad rows of tiles read what
earlier, larger calls left behind - ADVICE r4 found such a leak in the fused in_proj + attention kernel).  A long-lived model runs a
random sequence of calls of many sizes across the entry points; every result must equal, bit for bit, the same call on a second model
whose context has only ever seen calls of that one size class.  Variant A and variant C image towers, text tower (ids), VAE."""
import json
import os
import random

import numpy as np
import pytest
import torch

from hoigen_amd import clip, synth, vae
from hoigen_amd.model import build_model

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_results_do_not_depend_on_the_calls_before_them():
    d = torch.device("cuda:0")
    g0 = json.load(open(f"{G}/g0_tokens.json"))
    sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
    sdc = dict(sd)
    sdc.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, 1)))
    long_a, long_c = build_model(sd).to(d), build_model(sdc, use_adapter=True).to(d)
    gen = torch.Generator(device=d).manual_seed(123)
    imgs = torch.randn(96, 3, 224, 224, device=d, generator=gen)
    pri = torch.randn(96, 14, 64, device=d, generator=gen)
    mask = torch.zeros(96, 14, dtype=torch.bool, device=d)
    mask[::2, 10:] = True
    ids = clip.tokenize(g0["hoi600"]["text"]).to(d)
    E, Gn = vae.Encoder().to(d), vae.Generator().to(d)
    E.load_state_dict(synth.to_torch(synth.encoder_state_dict(2)))
    Gn.load_state_dict(synth.to_torch(synth.generator_state_dict(3)))
    V = vae.VAE(E, Gn)
    x = vae.l2_normalize(torch.randn(40000, 512, device=d, generator=gen))
    eps = torch.randn(40000, 512, device=d, generator=gen)

    def call(models, kind, n, off):
        ma, mc = models
        if kind == "A":
            return (ma.visual(imgs[off:off + n]).float(),)
        if kind == "C":
            g, l = mc.visual(imgs[off:off + n], (pri[off:off + n], mask[off:off + n]))
            return g.float(), l.float()
        if kind == "T":
            return (ma.encode_text(ids[off:off + n]).float(),)
        return tuple(t.float() for t in V(x[off:off + n], eps[off:off + n]))

    rng = random.Random(7)
    sizes = {"A": (1, 2, 3, 7, 32, 40, 64, 96), "C": (1, 3, 32, 40, 64), "T": (1, 5, 40, 64, 300, 600), "V": (1, 100, 1000, 33000, 40000)}
    maxn = {"A": 96, "C": 96, "T": 600, "V": 40000}
    plan = [(k, n, rng.randrange(0, maxn[k] - n + 1)) for k in sizes for n in sizes[k]]
    rng.shuffle(plan)
    plan = plan + plan[::-1]      # every size is seen both before and after every other
    got = [(k, n, off, call((long_a, long_c), k, n, off)) for k, n, off in plan]
    assert all(torch.isfinite(t).all() for *_, ts in got for t in ts)
    # the same calls on fresh models (one pair per call: nothing before it)
    checked = set()
    for k, n, off, ts in got:
        key = (k, n, off)
        if key in checked:
            continue
        checked.add(key)
        fresh = (build_model(sd).to(d), build_model(sdc, use_adapter=True).to(d)) if k in "ACT" else (None, None)
        ref = call(fresh, k, n, off)
        for a, b in zip(ts, ref):
            assert torch.equal(a, b), f"{k} n={n} off={off}: the result depends on the calls before it"
        # (both occurrences of the call in the long sequence agree as well)
        for k2, n2, off2, ts2 in got:
            if (k2, n2, off2) == key:
                for a, b in zip(ts2, ts):
                    assert torch.equal(a, b)
        del fresh
