"""encode_text over the 600 HICO prompts at 77 tokens: option mlp_pair with different tile orders / work splits, alternating, ms per call.
usage: python tools/text_pair_sweep.py ["pair:chunk:slots ..."]"""
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
m.truncate_text = False
cfgs = [tuple(int(v) for v in t.split(":")) for t in (sys.argv[1] if len(sys.argv) > 1 else
        "0:32:32 1:32:32 1:32:30 1:32:28 1:8:32 1:4:32 1:2:32 1:8:30 1:4:28 2:8:32").split()]
def timed(n=20):
    for _ in range(3): m.encode_text(ids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m.encode_text(ids)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
want = None
for rnd in range(3):
    out = []
    for pair, ch, sl in cfgs:
        m.set_option("mlp_pair", pair); m.set_option("mlp_pair_chunk", ch); m.set_option("mlp_pair_fc_slots", sl)
        if rnd == 0:
            got = m.encode_text(ids)
            if want is None: want = got.clone()
            assert torch.equal(got, want), (pair, ch, sl)
        out.append(f"{pair}:{ch}:{sl} {timed():.3f}")
    print(f"round {rnd}: " + " | ".join(out), flush=True)
