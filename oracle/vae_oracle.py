"""CPU ORACLE (test infrastructure only) — CoOp-VAE feature generator restatement.

Restates /root/reference/main_coop_vae.py (Encoder :261-279, reparameterise :445-447, Generator
:282-296, vae_loss :300-303, PromptLearner_*.forward :119-128) and ``mlp_net``
(/root/reference/finetune_ship.py:302-314, main_tip_finetune.py:313-324) as plain tensor algebra.
Pinned against outputs of the reference classes themselves (tests/golden/make_golden.py).
The product never imports this module.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

Tensor = torch.Tensor


def _lin(x: Tensor, sd: Dict[str, Tensor], name: str) -> Tensor:
    return x @ sd[name + ".weight"].to(x.dtype).T + sd[name + ".bias"].to(x.dtype)


def encoder(sd: Dict[str, Tensor], x: Tensor) -> Tuple[Tensor, Tensor]:
    """Encoder.forward (main_coop_vae.py:273-279): h = relu(net.0(x)); mean(h), log_var(h)."""
    h = torch.relu(_lin(x, sd, "net.0"))
    return _lin(h, sd, "mean"), _lin(h, sd, "log_var")


def reparameterise(mean: Tensor, log_var: Tensor, eps: Tensor) -> Tensor:
    """main_coop_vae.py:445-447: std = exp(0.5*log_var); z = std*eps + mean (eps given, not drawn)."""
    return torch.exp(0.5 * log_var) * eps.to(mean.dtype) + mean


def generator(sd: Dict[str, Tensor], z: Tensor) -> Tensor:
    """Generator.forward (main_coop_vae.py:293-296): net.2(relu(net.0(z)))."""
    return _lin(torch.relu(_lin(z, sd, "net.0")), sd, "net.2")


def vae_forward(sd_e, sd_g, x: Tensor, eps: Tensor):
    mean, log_var = encoder(sd_e, x)
    z = reparameterise(mean, log_var, eps)
    return mean, log_var, z, generator(sd_g, z)


def vae_loss(recon: Tensor, x: Tensor, mean: Tensor, log_var: Tensor) -> Tensor:
    """main_coop_vae.py:300-303."""
    rec = ((recon - x) ** 2).sum(1).mean()
    kld = (-0.5 * (1 + log_var - mean ** 2 - torch.exp(log_var))).sum(1).mean()
    return rec + kld


def mlp_net(sd: Dict[str, Tensor], x: Tensor) -> Tensor:
    """mlp_net.forward (finetune_ship.py:302-314): Linear-ReLU-Linear-ReLU-Linear."""
    h = torch.relu(_lin(x, sd, "net.0"))
    h = torch.relu(_lin(h, sd, "net.2"))
    return _lin(h, sd, "net.4")


def assemble_prompts(token_prefix: Tensor, token_suffix: Tensor, ctx: Tensor, bias: Tensor,
                     target: Tensor) -> Tensor:
    """PromptLearner_*.forward (main_coop_vae.py:119-128):
    cat([prefix[target] (R,1,D), ctx[None]+bias[:,None] (R,n_ctx,D), suffix[target]], dim=1)."""
    t = target.long()
    shifted = ctx.to(bias.dtype)[None, :, :] + bias[:, None, :]
    return torch.cat([token_prefix.to(bias.dtype)[t], shifted, token_suffix.to(bias.dtype)[t]], dim=1)
