"""Diagnostics: package power and shader clock (rocm-smi, bench.power_sample) under a burst of one VAE path: HG_VAE_FUSED=0|1|2, GEN=1 = generator only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
import bench
from hoigen_amd import synth, vae
dev = torch.device("cuda:0")
E, G = vae.Encoder().to(dev).eval(), vae.Generator().to(dev).eval()
E.load_state_dict(synth.to_torch(synth.encoder_state_dict(2)))
G.load_state_dict(synth.to_torch(synth.generator_state_dict(3)))
R = int(os.environ.get("R", 98304))
x = torch.nn.functional.normalize(torch.randn(R, 512, device=dev), dim=-1)
eps = torch.randn(R, 512, device=dev)
V = vae.VAE(E, G)
step = (lambda: G(eps)) if os.environ.get("GEN") else (lambda: V(x, eps))
for _ in range(3):
    step()
torch.cuda.synchronize()
pw = bench.power_sample(step, lambda: torch.cuda.synchronize(dev), seconds=3.0)
print("HG_VAE_FUSED", os.environ.get("HG_VAE_FUSED"), "GEN", os.environ.get("GEN"), pw)
