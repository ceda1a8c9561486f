"""Device-resident feature sampling: the generation loop HOIGen runs at detector start-up
(/root/reference/main_tip_finetune.py:749-824; training-time twin finetune_ship.py:503-523).

    for _ in range(100):                      # per branch: hoi / human / object
        z     = randn(n, 512)
        bias  = netG(z)                       # Generator
        p     = prompt_learner(bias, target)  # [n,77,512]
        t     = text_encoder(p, tokenized[target])
        f     = mlp(t / t.norm(dim=-1, keepdim=True))
    gen_feature = cat([hoi..., human..., object...]);  gen_target likewise

Differences from the reference, none of them numerical: ε is drawn on the device (Philox) instead of on the
host; class names are tokenised once (the reference re-tokenises 3 x 100 times on the CPU, :783,793,802); the
three branches of an iteration go through the text tower in ONE call (1 800 prompts instead of 3 x 600).
Every stage runs on the HIP kernels (hg_generator, hg_assemble_prompts, hg_encode_text_embeds,
hg_l2_normalize, hg_mlp_net).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import vae
from .model import CLIP


@dataclass
class Branch:
    """One of the reference's hoi / human / object triples (main_tip_finetune.py:693-738)."""
    generator: vae.Generator
    prompt_learner: vae._PromptLearner
    mlp: vae.mlp_net
    target: torch.Tensor          # class index per generated row, e.g. arange(600) / HOI_IDX_TO_OBJ_IDX


class FeatureSampler:
    def __init__(self, clip_model: CLIP, branches: "Dict[str, Branch]"):
        self.clip = clip_model
        self.branches = branches
        self.text_encoder = vae.TextEncoder(clip_model)

    @torch.no_grad()
    def step(self, z: "Dict[str, torch.Tensor]") -> "Dict[str, torch.Tensor]":
        """One loop iteration for given latents ``z[name] [n_name, 512]`` -> features per branch."""
        prompts, toks, sizes, names = [], [], [], []
        for name, br in self.branches.items():
            tgt = br.target.to(z[name].device)
            bias = br.generator(z[name])
            prompts.append(br.prompt_learner(bias, tgt))
            toks.append(br.prompt_learner.tokenized_prompts[tgt])
            sizes.append(len(tgt))
            names.append(name)
        t = self.text_encoder(torch.cat(prompts, dim=0), torch.cat(toks, dim=0))   # one pass over the tower
        t = vae.l2_normalize(t)
        out, o = {}, 0
        for name, n in zip(names, sizes):
            out[name] = self.branches[name].mlp(t[o:o + n])
            o += n
        return out

    @torch.no_grad()
    def sample(self, iterations: int = 100, generator: Optional[torch.Generator] = None
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (gen_feature [iterations * sum(n), 512], gen_target) in the reference's concatenation order:
        all iterations of the first branch, then the second, ... (main_tip_finetune.py:817-824)."""
        dev = self.clip.positional_embedding.device
        feats: "Dict[str, List[torch.Tensor]]" = {k: [] for k in self.branches}
        for _ in range(iterations):
            z = {k: torch.randn(len(b.target), b.generator.dim, device=dev, generator=generator)
                 for k, b in self.branches.items()}
            for k, v in self.step(z).items():
                feats[k].append(v)
        gen_feature = torch.cat([torch.cat(feats[k], dim=0) for k in self.branches], dim=0)
        gen_target = torch.cat([b.target.to(dev).repeat(iterations) for b in self.branches.values()], dim=0)
        return gen_feature, gen_target
