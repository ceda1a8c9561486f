"""Diagnostics: encode_image throughput against the batch size (does a working set that fits the 256 MiB Infinity Cache pay?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from hoigen_amd import synth
from hoigen_amd.model import build_model
dev = torch.device("cuda:0")
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev)
for B in [int(b) for b in os.environ.get("BATCHES", "256,128,64,32,192,256").split(",")]:
    xs = [x[i:i + B] for i in range(0, 256, B)]
    for _ in range(3):
        for t in xs:
            m.encode_image(t)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        for t in xs:
            m.encode_image(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("batch %3d x %d: %.3f ms per 256 crops, %.0f crops/s" % (B, len(xs), dt * 1e3, 256 / dt))
