#!/usr/bin/env python3
"""Round-5 sweeps on the final library (library default paths), one line per series: variant A and variant C (14 prior tokens) over batch
sizes, encode_text over prompt counts at 77 tokens and truncated, per setting of text_ln_fold."""
import json, os, sys, time
import torch
torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoigen_amd import clip, synth
from hoigen_amd.model import build_model

dev = torch.device("cuda", 0)
sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
mA = build_model(sd).to(dev)
sdc = dict(sd); sdc.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, 21)))
mC = build_model(sdc, use_adapter=True, adapter_pos="all").to(dev)
g0 = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g0_tokens.json")))
ids_all = clip.tokenize((g0["hoi600"]["text"] * 4)[:2400]).to(dev)

def t(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

batches = [1, 4, 16, 32, 64, 128, 192, 256, 512]
xa = torch.randn(512, 3, 224, 224, device=dev)
pri = torch.randn(512, 14, 64, device=dev); mask = torch.zeros(512, 14, dtype=torch.bool, device=dev); mask[:, 10:] = True
print("variant A (default path) batch:ms  " + " ".join(f"{B}:{t(lambda: mA.visual(xa[:B]), max(5, min(40, 1500 // B))):.3f}" for B in batches), flush=True)
print("variant C (14 priors)    batch:ms  " + " ".join(f"{B}:{t(lambda: mC.visual(xa[:B], (pri[:B], mask[:B])), max(5, min(40, 1500 // B))):.3f}" for B in batches), flush=True)
for fold in (1, 0, 2):
    mA.set_option("text_ln_fold", fold)
    for trunc in (False, True):
        mA.truncate_text = trunc
        print(f"encode_text text_ln_fold={fold} {'truncated' if trunc else '77 tokens'} prompts:ms  " +
              " ".join(f"{n}:{t(lambda: mA.encode_text(ids_all[:n]), 10):.3f}" for n in (1, 8, 64, 117, 600, 1200, 2400)), flush=True)
mA.set_option("text_ln_fold", 1)
