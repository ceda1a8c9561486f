"""encode_text over the 600 HICO prompts (77 tokens and truncated) with option mlp_pair 0 / 1 / 2, alternating, ms per call."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hoigen_amd import synth
from hoigen_amd.model import build_model
torch.set_grad_enabled(False)
d = torch.device("cuda:0")
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(d)
g0 = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g0_tokens.json")))
rows = g0["hoi600"]["ids"]
ids = np.zeros((len(rows), 77), np.int64)
for i, r in enumerate(rows): ids[i, :len(r)] = r
ids = torch.from_numpy(ids).to(d)
def timed(n=20):
    for _ in range(3): m.encode_text(ids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m.encode_text(ids)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rnd in range(3):
    for trunc in (False, True):
        m.truncate_text = trunc
        out = []
        for pair in (0, 1, 2):
            m.set_option("mlp_pair", pair)
            out.append(f"mlp_pair={pair}: {timed():.3f} ms")
        print(f"round {rnd} {'truncated' if trunc else '77 tokens'}: " + " | ".join(out), flush=True)
