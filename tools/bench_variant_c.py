#!/usr/bin/env python3
"""Variant C (`visual(x, prior)` with trained instance adapters: the detector's real call,
upt_tip_cache_model_free_finetune_distill3.py:1615) against variant A at batch 256, N = 14 prior tokens (4 padded).
Prints one JSON line; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)   # inference only
from hoigen_amd import synth
from hoigen_amd.model import build_model

dev = torch.device("cuda:0")
B, N = int(os.environ.get("B", 256)), 14
sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
mA = build_model(sd).to(dev)
sdc = dict(sd); sdc.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, 21)))
mC = build_model(sdc, use_adapter=True, adapter_pos="all").to(dev)
g = torch.Generator(device=dev).manual_seed(1)
img = torch.randn(B, 3, 224, 224, device=dev, generator=g)
pri = torch.randn(B, N, 64, device=dev, generator=g)
mask = torch.zeros(B, N, dtype=torch.bool, device=dev); mask[:, N - 4:] = True

def t(fn, reps=int(os.environ.get("REPS", 10))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

mA.visual.set_option("last_block_row0", 0)      # every row of every block, like the headline
a = t(lambda: mA.visual(img))
c_prior = t(lambda: mC.visual(img, (pri, mask)))
c_none = t(lambda: mC.visual(img, None))
print(json.dumps({"batch": B, "variant_A_ms": round(a, 3), "variant_C_prior_ms": round(c_prior, 3),
                  "variant_C_noprior_ms": round(c_none, 3), "C_over_A": round(c_prior / a, 3),
                  "flops_per_crop_C_gflop": 35.127 + 0.1541 + 12 * 0.0494}))
