"""Where a folded text tower loses its parity margin - a study on the GPU against the CPU oracle (option text_ln_fold).
The weight-side fold (setting 2, the vision tower's form) multiplies fp16(x - mu) by W' = fp16(W * gamma): the weights are rounded a
second time, which the separate-LayerNorm path (and the reference: fp16-representable W) never does.  With gamma = 1 (synth ln_jitter =
0) W' = W: the excess error of setting 2 vanishes there - it IS that second rounding - and the activation-side fold (setting 1, the
default: gamma multiplied into the copy the residual GEMM hands on, W untouched) has no such term at any gamma.  128 prompts x 77
tokens, oracle = oracle/clip_oracle.py encode_text (fp32 activations, the reference's weight rounding)."""
import json
import os

import numpy as np
import pytest
import torch

from hoigen_amd import clip, synth
from hoigen_amd.model import build_model

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _errors(ln_jitter):
    from oracle import clip_oracle as co
    d = torch.device("cuda:0")
    g0 = json.load(open(f"{G}/g0_tokens.json"))
    raw = synth.clip_state_dict(synth.VIT_B16, 0, ln_jitter=ln_jitter)
    ids = clip.tokenize(g0["hoi600"]["text"][:128])
    ref = co.encode_text(co.reference_weight_rounding(raw), ids).double()
    m = build_model(synth.to_torch(raw)).to(d)
    m.truncate_text = False
    out = {}
    for fold in (0, 1, 2):
        m.set_option("text_ln_fold", fold)
        big = torch.cat([ids] * 5)[:600].to(d)       # (>= 512 rows x tokens either way; the folded path needs M >= 512)
        e = m.encode_text(big).double().cpu()[:128]
        rows = (e - ref).norm(dim=1) / ref.norm(dim=1)
        out[fold] = (float((e - ref).norm() / ref.norm()), float(rows.max()))
    m.set_option("text_ln_fold", 1)
    return out


def test_excess_error_of_the_folded_text_tower_is_the_second_weight_rounding():
    res = {j: _errors(j) for j in (0.1, 0.0)}
    for j, r in res.items():
        print(f"\nln_jitter {j}: separate LayerNorm {r[0][0]:.3e} (worst prompt {r[0][1]:.3e}) | gamma in the activation copy {r[1][0]:.3e} "
              f"(worst {r[1][1]:.3e}) | gamma in the weights {r[2][0]:.3e} (worst {r[2][1]:.3e})")
    for j in res:
        for fold in (0, 1, 2):
            assert res[j][fold][0] <= 1e-3
    assert abs(res[0.0][2][0] - res[0.0][1][0]) < 3e-5, "with gamma = 1 the two folds must agree (no second rounding of the weights)"
    assert res[0.1][2][0] > res[0.1][1][0] + 5e-5, "with gamma != 1 the weight-side fold pays for fp16(W * gamma)"
