#!/usr/bin/env python3
"""Config 3 alone: encode_text over the 600 HICO prompts at all 77 tokens (no truncation), REPS calls; run under
rocprofv3 --kernel-trace --stats for the per-kernel split (tools/gpu_trace_text77.sh)."""
import json, os, sys, time
import torch
torch.set_grad_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoigen_amd import clip, synth
from hoigen_amd.model import build_model

dev = torch.device("cuda:0")
g0 = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g0_tokens.json")))
model = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
ids = clip.tokenize(g0["hoi600"]["text"]).to(dev)
model.truncate_text = False
for k, v in [kv.split("=") for kv in os.environ.get("OPTS", "").split()]:
    model.set_option(k, int(v))
for _ in range(3):
    model.encode_text(ids)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = int(os.environ.get("REPS", 20))
for _ in range(reps):
    model.encode_text(ids)
torch.cuda.synchronize()
print(json.dumps({"config3_full77_ms": round((time.perf_counter() - t0) / reps * 1e3, 4), "opts": os.environ.get("OPTS", "")}))
