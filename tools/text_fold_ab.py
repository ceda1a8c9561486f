#!/usr/bin/env python3
"""Text tower with LayerNorm folded into its GEMMs (option text_ln_fold) against the separate LayerNorm kernels: config 3 time at 77 and
truncated tokens and rel-L2 (whole / worst prompt) against the reference's own outputs (tests/golden/g3_vitb16_text.npz)."""
import json, os, sys, time
import numpy as np, torch
torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoigen_amd import clip, synth
from hoigen_amd.model import build_model
dev = torch.device("cuda:0")
G = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
g0 = json.load(open(f"{G}/g0_tokens.json")); g3 = dict(np.load(f"{G}/g3_vitb16_text.npz"))
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
ids = clip.tokenize(g0["hoi600"]["text"]).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for fold, hilo in ((0, 1), (1, 1), (1, 0), (0, 1), (1, 1), (1, 0)):
    m.set_option("text_ln_fold", fold)
    m.set_option("stream_hilo", hilo)
    out = {}
    for trunc in (False, True):
        m.truncate_text = trunc
        e = m.encode_text(ids).float().cpu().numpy().astype(np.float64)
        ref = g3["hoi600"].astype(np.float64)
        whole = np.linalg.norm(e - ref) / np.linalg.norm(ref)
        worst = (np.linalg.norm(e - ref, axis=1) / np.linalg.norm(ref, axis=1)).max()
        out["trunc" if trunc else "full77"] = {"ms": round(t(lambda: m.encode_text(ids)), 4), "rel_l2": float(f"{whole:.3e}"), "worst_prompt": float(f"{worst:.3e}")}
    for name in ("obj81", "verb117"):
        if name in g0 and name in g3:
            m.truncate_text = False
            e = m.encode_text(clip.tokenize(g0[name]["text"]).to(dev)).float().cpu().numpy().astype(np.float64)
            ref = g3[name].astype(np.float64)
            out[name] = [float(f"{np.linalg.norm(e - ref) / np.linalg.norm(ref):.3e}"), float(f"{(np.linalg.norm(e - ref, axis=1) / np.linalg.norm(ref, axis=1)).max():.3e}")]
    print(json.dumps({"text_ln_fold": fold, "stream_hilo": hilo, **out}))
