#!/usr/bin/env python3
"""encode_image latency / throughput over batch sizes (library default path), one JSON line per batch size."""
import json, os, sys, time
import torch
torch.set_grad_enabled(False)   # inference only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoigen_amd import synth
from hoigen_amd.model import build_model

dev = torch.device("cuda", 0)
model = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
for B in [int(b) for b in os.environ.get("BATCHES", "1 2 4 8 16 32 64 128 256 512").split()]:
    x = torch.randn(B, 3, 224, 224, device=dev)
    for _ in range(3):
        model.visual(x)
    torch.cuda.synchronize()
    n = max(5, min(50, 2000 // B))
    t0 = time.perf_counter()
    for _ in range(n):
        model.visual(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"batch": B, "ms": round(dt * 1e3, 3), "crops_per_s": round(B / dt, 1)}), flush=True)
