"""Timing of option mlp_pair on the headline step (256 crops, every row of every block): ms per step for the two launches and for the
one-launch MLP with different lags, alternating, hipEvents around 20 steps each; bit-equality checked first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import synth
from hoigen_amd.model import build_model

torch.set_grad_enabled(False)
d = torch.device("cuda:0")
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(d)
m.visual.set_option("last_block_row0", 0)
x = torch.randn(256, 3, 224, 224, device=d)
m.visual.set_option("mlp_pair", 0)
want = m.encode_image(x)
import os
SW = os.environ.get("PAIR_SWEEP", "32:8 30:8 30:4 30:25 28:8")
configs = [("two launches", 0, 30, 8)] + [(f"pair fc_slots {l} chunk {c}", 1, int(l), int(c)) for l, c in (t.split(":") for t in SW.split())]
for name, on, lag, ch in configs[1:]:
    m.visual.set_option("mlp_pair", on); m.visual.set_option("mlp_pair_fc_slots", lag); m.visual.set_option("mlp_pair_chunk", ch)
    got = m.encode_image(x)
    print(f"{name}: bit-identical {bool(torch.equal(got, want))} max |diff| {float((got.float() - want.float()).abs().max()):.3e}", flush=True)
for rnd in range(3):
    for name, on, lag, ch in configs:
        m.visual.set_option("mlp_pair", on); m.visual.set_option("mlp_pair_fc_slots", lag); m.visual.set_option("mlp_pair_chunk", ch)
        for _ in range(3): m.encode_image(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.encode_image(x)
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} {name}: {e0.elapsed_time(e1) / 20:.3f} ms/step", flush=True)
