"""Diagnostics: the generation pipeline of main_tip_finetune.py:759-824 (three HICO branches x 600 targets, `ITERS` iterations) through
FeatureSampler; prints ms per iteration; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from hoigen_amd import synth, vae
from hoigen_amd.generation import hico_sampler
from hoigen_amd.model import build_model

dev = torch.device("cuda:0")
g0 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g0_tokens.json")))
model = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
sampler = hico_sampler(model, g0["_classnames"], seed=70)
iters = int(os.environ.get("ITERS", 100))
print("auto batch_iters:", sampler._auto_batch(iters))
for bi in [int(x) for x in os.environ.get("BATCH_ITERS", "1,4,8,0").split(",")]:
    sampler.sample(iterations=min(iters, 2 * max(bi, 13)), batch_iters=bi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feat, tgt = sampler.sample(iterations=iters, batch_iters=bi)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch_iters {bi}: {iters} iterations {dt * 1e3:.1f} ms = {dt / iters * 1e3:.3f} ms per iteration, {feat.shape[0] / dt:.0f} features/s, finite {bool(torch.isfinite(feat).all())}")
