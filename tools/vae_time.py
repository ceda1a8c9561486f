"""Diagnostics: config 4 (VAE on 100 000 rows) a few times; run under `rocprofv3 --kernel-trace --stats`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from hoigen_amd import synth, vae
dev = torch.device("cuda:0")
E, G = vae.Encoder().to(dev).eval(), vae.Generator().to(dev).eval()
E.load_state_dict(synth.to_torch(synth.encoder_state_dict(2)))
G.load_state_dict(synth.to_torch(synth.generator_state_dict(3)))
R = int(os.environ.get("R", 100000))
x = torch.nn.functional.normalize(torch.randn(R, 512, device=dev), dim=-1)
eps = torch.randn(R, 512, device=dev)
if os.environ.get("ZERO"):      # all-zero operands: what the same instruction stream takes when the MFMA datapath does not toggle (power / clock)
    for m in (E, G):
        for prm in m.parameters():
            prm.data.zero_()
    x.zero_(); eps.zero_()
V = vae.VAE(E, G)
for _ in range(3):
    V(x, eps)
torch.cuda.synchronize()
n = int(os.environ.get("ITERS", 10))
t0 = time.perf_counter()
for _ in range(n):
    out = V(x, eps)
torch.cuda.synchronize()
print("vae_forward %d rows: %.3f ms" % (R, (time.perf_counter() - t0) / n * 1e3))
t0 = time.perf_counter()
for _ in range(n):
    G(eps)
torch.cuda.synchronize()
print("generator %d rows: %.3f ms" % (R, (time.perf_counter() - t0) / n * 1e3))
