"""Diagnostics: variant C (ViT-B/16, trained adapters, 14 priors) against the reference fixture g2 - whole-tensor and worst-row
rel-L2 of the four outputs; run once per HG_ADAPTER_KCAT mode (the switch is read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_grad_enabled(False)
from hoigen_amd import synth
from hoigen_amd.model import build_model
d = torch.device("cuda:0")
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/g2_vitb16_image.npz")))
sd = synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))
sd.update(synth.to_torch(synth.adapter_state_dict(synth.VIT_B16, 1)))
m = build_model(sd, use_adapter=True).to(d)
img = torch.from_numpy(synth.crops(4, 224, seed=1234)).to(d)
pri, mask = synth.priors(4, n=14, dim=64, n_pad=4, seed=99)


def rel(a, b):
    a, b = a.float().cpu().double(), torch.from_numpy(b).double()
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    return float((a - b).norm() / b.norm()), float(((a - b).norm(dim=1) / b.norm(dim=1)).max())


gl, lo = m.visual(img, (torch.from_numpy(pri).to(d), torch.from_numpy(mask).to(d)))
g2_, l2_ = m.visual(img[:2], None)
out = {"mode": os.environ.get("HG_ADAPTER_KCAT", "default"),
       "prior global": rel(gl, g["c_prior_global"]), "prior local": rel(lo.permute(0, 2, 3, 1), np.transpose(g["c_prior_local"], (0, 2, 3, 1))),
       "no prior global": rel(g2_, g["c_noprior_global"]),
       "no prior local": rel(l2_.permute(0, 2, 3, 1), np.transpose(g["c_noprior_local"], (0, 2, 3, 1)))}
print({k: (v if isinstance(v, str) else tuple(float("%.2e" % x) for x in v)) for k, v in out.items()})
