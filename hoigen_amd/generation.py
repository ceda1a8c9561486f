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
    def __init__(self, clip_model: CLIP, branches: "Dict[str, Branch]", text_ln_fold: Optional[int] = None):
        """``text_ln_fold``: value of the library option of that name (include/hoigen_amd.h) for the duration of ``sample()``; None keeps
        the model's setting (default 1: LayerNorm folded with its weight in the activation copy).  2 = the weight folded into the GEMM
        weights: the HICO loop 2.8 -> 2.6 ms per iteration for 1.4e-4 of the 1e-3 parity tolerance against the reference."""
        self.clip = clip_model
        self.branches = branches
        self.text_encoder = vae.TextEncoder(clip_model)
        self.text_ln_fold = text_ln_fold

    def _tokens_run(self) -> int:
        """max(EOT) + 1 over every class of every branch: the tokens the truncated text tower reads (clipnet/model.py:350 selects the
        EOT row; causal attention: later positions cannot reach it)."""
        key = tuple(id(b.prompt_learner.tokenized_prompts) for b in self.branches.values())
        if getattr(self, "_lt_key", None) != key:
            self._lt = 1 + max(int(b.prompt_learner.tokenized_prompts.argmax(dim=-1).max()) for b in self.branches.values())
            self._lt_key = key
        return self._lt

    @torch.no_grad()
    def step(self, z: "Dict[str, torch.Tensor]", targets: "Optional[Dict[str, torch.Tensor]]" = None) -> "Dict[str, torch.Tensor]":
        """One loop iteration for given latents ``z[name] [n_name, 512]`` -> features per branch.  With a truncating text tower
        (``clip_model.truncate_text``, the default) the prompts are assembled with the tokens it reads only, each branch straight into
        its slice of the one batch the tower is called with.  ``targets[name]``: the class of every row of ``z[name]`` (default: the
        branch's own ``target``; ``sample`` passes the stacked targets of several iterations - the branches are not modified)."""
        names = list(self.branches)
        tg = {k: (targets[k] if targets is not None else self.branches[k].target) for k in names}
        sizes = [len(tg[k]) for k in names]
        dev = z[names[0]].device
        lt = self._tokens_run() if getattr(self.clip, "truncate_text", False) else None
        L = lt if lt is not None else 1 + self.branches[names[0]].prompt_learner.n_ctx + self.branches[names[0]].prompt_learner.token_suffix.shape[1]
        prompts = torch.empty(sum(sizes), L, z[names[0]].shape[1], device=dev, dtype=torch.float32)
        toks, o = [], 0
        for name, n in zip(names, sizes):
            br = self.branches[name]
            tgt = tg[name].to(dev)
            bias = br.generator(z[name])
            br.prompt_learner(bias, tgt, out=prompts[o:o + n], tokens=lt)
            toks.append(br.prompt_learner.tokenized_prompts[tgt])
            o += n
        t = self.text_encoder(prompts, torch.cat(toks, dim=0))   # one pass over the tower
        t = vae.l2_normalize(t)
        out, o = {}, 0
        for name, n in zip(names, sizes):
            out[name] = self.branches[name].mlp(t[o:o + n])
            o += n
        return out

    TEXT_PASS_ROWS = 65536      # rows (prompts x executed tokens) of one pass of the text tower (hg_api.hip: text_chunk_prompts)

    def _auto_batch(self, iterations: int) -> int:
        """Iterations per step that fill the text tower's passes best: rows of a step = k x prompts x executed tokens, cut into equal
        passes of at most TEXT_PASS_ROWS; the fill of the passes decides (1 800 prompts at 14 tokens: k = 13 -> five passes 99.98 % full,
        k = 8 -> four passes at 77 %)."""
        n = sum(len(b.target) for b in self.branches.values())
        lt = self._tokens_run()
        best, best_fill = 1, 0.0
        for k in range(1, min(16, iterations) + 1):
            rows = k * n * lt
            passes = -(-rows // self.TEXT_PASS_ROWS)
            fill = rows / (passes * self.TEXT_PASS_ROWS)
            if fill > best_fill + 1e-9:
                best, best_fill = k, fill
        return best

    @torch.no_grad()
    def sample(self, iterations: int = 100, generator: Optional[torch.Generator] = None, batch_iters: int = 0
               ) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (gen_feature [iterations * sum(n), 512], gen_target) in the reference's concatenation order:
        all iterations of the first branch, then the second, ... (main_tip_finetune.py:817-824).

        ``batch_iters`` iterations (0 = chosen so that the text tower's passes are full) go through the kernels as ONE step (the iterations are independent draws: their rows are stacked,
        iteration-major, per branch): one iteration's 1 800 prompts at 13-16 executed tokens are 113 row tiles of 256 on 256 CUs - less
        than half a round per GEMM; the latents are drawn per iteration and branch in the reference's order either way."""
        dev = self.clip.positional_embedding.device
        if batch_iters <= 0:
            batch_iters = self._auto_batch(iterations)      # (0: chosen from the shapes)
        feats: "Dict[str, List[torch.Tensor]]" = {k: [] for k in self.branches}
        done = 0
        prev_fold = self.clip.get_option("text_ln_fold") if self.text_ln_fold is not None else None
        if self.text_ln_fold is not None:
            self.clip.set_option("text_ln_fold", int(self.text_ln_fold))
        try:
            while done < iterations:
                k_it = min(max(1, batch_iters), iterations - done)
                zs = [{k: torch.randn(len(b.target), b.generator.dim, device=dev, generator=generator) for k, b in self.branches.items()}
                      for _ in range(k_it)]
                z = {k: torch.cat([zi[k] for zi in zs], dim=0) for k in self.branches}
                stacked = {k: b.target.repeat(k_it) for k, b in self.branches.items()}      # (passed down: the branches stay as they are)
                for k, v in self.step(z, stacked).items():
                    feats[k].append(v)
                done += k_it
        finally:
            if self.text_ln_fold is not None:
                self.clip.set_option("text_ln_fold", prev_fold)      # (what was in force before: the caller's choice, not a constant)
        gen_feature = torch.cat([torch.cat(feats[k], dim=0) for k in self.branches], dim=0)
        gen_target = torch.cat([b.target.to(dev).repeat(iterations) for b in self.branches.values()], dim=0)
        return gen_feature, gen_target


def hico_sampler(clip_model: CLIP, classnames: "Dict[str, Sequence[str]]", seed: int = 70) -> FeatureSampler:
    """The three-branch sampler of main_tip_finetune.py:693-824 on seeded synthetic Generator / mlp_net / context weights (the
    reference loads ./ckpt/<zs_type>/*_50.pth, not reachable offline): hoi = 600 HOI names, targets arange(600); human and object =
    the 80 human / object names, both with target HOI_IDX_TO_OBJ_IDX[i] as the reference has it (:772-779) - here the object index of
    the HOI's name.  ``classnames``: {"hoi": [...600], "hum": [...80], "obj": [...80]} (tests/golden/g0_tokens.json "_classnames")."""
    from . import synth
    dev = clip_model.positional_embedding.device
    m = clip_model
    hoi_names, obj_names = list(classnames["hoi"]), list(classnames["obj"])
    # object index of every HOI: the object is the tail of the HOI name ("ride bicycle" -> "bicycle"); longest match wins
    by_len = sorted(range(len(obj_names)), key=lambda i: -len(obj_names[i]))
    def obj_of(name):
        nm = name.replace("_", " ")
        for i in by_len:
            if nm.endswith(obj_names[i].replace("_", " ")):
                return i
        return 0
    hoi_to_obj = torch.tensor([obj_of(n) for n in hoi_names])
    spec = (("hoi", vae.PromptLearner_hoi, hoi_names, torch.arange(len(hoi_names))),
            ("human", vae.PromptLearner_h, list(classnames["hum"]), hoi_to_obj),
            ("object", vae.PromptLearner_o, obj_names, hoi_to_obj))
    branches = {}
    for i, (k, cls, names, tgt) in enumerate(spec):
        G_, M_ = vae.Generator().to(dev), vae.mlp_net(512, 512, 512).to(dev)
        G_.load_state_dict(synth.to_torch(synth.generator_state_dict(seed + i)))
        M_.load_state_dict(synth.to_torch(synth.mlp_net_state_dict(seed + 10 + i)))
        pl = cls(names, m).to(dev)
        with torch.no_grad():
            pl.ctx.copy_(torch.from_numpy(synth.hg_normal((pl.n_ctx, 512), seed + 20 + i, 0.02)).to(pl.ctx.dtype))
        branches[k] = Branch(G_, pl, M_, tgt)
    return FeatureSampler(m, branches)
