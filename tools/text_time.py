"""Diagnostics: config 3 (encode_text over 600 prompts x 77 tokens, truncation off); run under `rocprofv3 --kernel-trace --stats`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_grad_enabled(False)
from hoigen_amd import synth
from hoigen_amd.model import build_model
dev = torch.device("cuda:0")
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(dev)
g0 = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g0_tokens.json")))
rows = g0["hoi600"]["ids"]
ids = np.zeros((len(rows), 77), np.int64)
for i, r in enumerate(rows): ids[i, :len(r)] = r
toks = torch.from_numpy(ids).to(dev)
m.truncate_text = os.environ.get("TRUNC", "0") == "1"
for _ in range(3): m.encode_text(toks)
torch.cuda.synchronize()
n = int(os.environ.get("ITERS", 10))
t0 = time.perf_counter()
for _ in range(n): m.encode_text(toks)
torch.cuda.synchronize()
print("encode_text 600 x 77 (truncate=%s): %.3f ms" % (m.truncate_text, (time.perf_counter() - t0) / n * 1e3))
