#!/usr/bin/env python3
"""g10_adapter_layers.npz: the reference's variant C (CLIP_models_adapter_prior2.py) with adapter_num_layers = 2 on the
tiny config: the prior path chains mhsa_layers.0 and mhsa_layers.1 (:150,179,190-195).  Build container only.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_adapter_layers.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from hoigen_amd import synth  # noqa: E402


@torch.no_grad()
def main():
    _, adapter_mod = mg.load_reference()
    cfg = synth.TINY
    model = adapter_mod.build_model(mg.t(synth.clip_state_dict(cfg, 10)), use_adapter=True, adapter_pos="all",
                                    adapter_num_layers=2)
    asd = mg.t(synth.adapter_state_dict(cfg, 13, num_layers=2))
    missing, unexpected = model.load_state_dict(asd, strict=False)
    assert not unexpected and not [k for k in missing if "adaptermlp" in k], (unexpected, missing)
    model.eval()
    img = torch.from_numpy(synth.crops(3, 32, seed=11))
    pri, mask = synth.priors(3, n=6, dim=64, n_pad=2, seed=14)
    g, l = model.visual(img, (torch.from_numpy(pri), torch.from_numpy(mask)))
    g0, l0 = model.visual(img, None)
    np.savez_compressed(f"{HERE}/g10_adapter_layers.npz", prior_global=g.numpy(), prior_local=l.contiguous().numpy(),
                        noprior_global=g0.numpy(), noprior_local=l0.contiguous().numpy())
    print("g10:", g.shape, l.shape)


if __name__ == "__main__":
    main()
