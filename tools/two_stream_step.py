"""Diagnostics: the headline step (256 crops, every row of every block) as ONE call against TWO half batches on two streams and two
contexts with full-size persistent grids, so that the workgroups of one half's next kernel fill the CUs the other half's last round
leaves idle (launch tails: DESIGN.md 7).  SPLIT="128 128" rows per stream; STEPS, WARMUP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from hoigen_amd import synth
from hoigen_amd.model import build_model
dev = torch.device("cuda:0")
sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
split = [int(v) for v in os.environ.get("SPLIT", "128 128").split()]
B = sum(split)
models = [build_model(sd).to(dev) for _ in range(len(split) + 1)]
for m in models: m.visual.set_option("last_block_row0", 0)
x = torch.randn(B, 3, 224, 224, device=dev)
parts = list(torch.split(x, split))
streams = [torch.cuda.Stream() for _ in split]
steps, warm = int(os.environ.get("STEPS", 30)), int(os.environ.get("WARMUP", 5))


def one():
    return models[-1].encode_image(x)


def two():
    outs = []
    for m, p, s in zip(models, parts, streams):
        with torch.cuda.stream(s):
            outs.append(m.encode_image(p))
    return outs


def timed(fn):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ref = one().float()
outs = two()
torch.cuda.synchronize()      # (the side streams' results are read on the default stream)
got = torch.cat([o.float() for o in outs])
print("max |two streams - one call| =", (ref - got).abs().max().item())
for rnd in range(3):
    print("round %d: one call %.3f ms | %d streams (%s) %.3f ms" % (rnd, timed(one), len(split), split, timed(two)), flush=True)
