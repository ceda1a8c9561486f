"""Deterministic synthetic weights and inputs (no torch RNG, no transcendental functions).

There is no network on the build or the GPU box, so ``ViT-B-16.pt`` (the checkpoint the
reference downloads, /root/reference/clipnet/clip.py:35) is not available.  Parity is therefore
pinned on *synthetic* state dicts that both sides can regenerate bit-for-bit:

* here (build container): ``tests/golden/make_golden.py`` feeds them to the imported reference and
  stores the reference outputs as fixtures;
* on the GPU box: the tests regenerate the same state dict from the same seed and compare the HIP
  path with the stored reference outputs.

The generator is a counter-based integer hash (splitmix64) mapped to an Irwin-Hall(4) variate, so
every step is exact integer / exactly-rounded float64 arithmetic: identical on every platform and
numpy version.  Shapes and standard deviations follow the reference's own initialisers
(/root/reference/clipnet/model.py:295-322, /root/reference/main_coop_vae.py:32-39).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

_SQRT3 = 1.7320508075688772


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hg_normal(shape, seed: int, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Approximately N(mean, std^2) float32 array, a pure function of (shape, seed).

    value = ((a+b+c+d)/65536 - 2) * sqrt(3) with a..d the four 16-bit fields of
    splitmix64(seed * 2^40 + index); variance of the sum of 4 U(0,1) is 1/3.
    """
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed & 0xFFFFFF) << np.uint64(40))
        h = _splitmix64(idx)
    m = np.uint64(0xFFFF)
    s = ((h & m) + ((h >> np.uint64(16)) & m) + ((h >> np.uint64(32)) & m) + ((h >> np.uint64(48)) & m))
    v = (s.astype(np.float64) / 65536.0 - 2.0) * _SQRT3
    return (v * std + mean).astype(np.float32).reshape(shape)


def _seed_of(base: int, name: str) -> int:
    return (base * 1000003 + zlib.crc32(name.encode())) & 0xFFFFFF


# ---------------------------------------------------------------------------------------------
# CLIP state dict (keys/shapes: /root/reference/clipnet/model.py:395-432)
# ---------------------------------------------------------------------------------------------

VIT_B16 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
               vision_patch_size=16, context_length=77, vocab_size=49408, transformer_width=512,
               transformer_heads=8, transformer_layers=12)

# heads = width // 64 is fixed by the reference (clipnet/model.py:268,417), so the smallest
# multi-head configuration has width 128.
TINY = dict(embed_dim=128, image_resolution=32, vision_layers=2, vision_width=128,
            vision_patch_size=16, context_length=16, vocab_size=512, transformer_width=128,
            transformer_heads=2, transformer_layers=2)


def _block(sd, prefix, width, layers_for_std, seed, ln_jitter):
    proj_std = (width ** -0.5) * ((2 * layers_for_std) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5

    def put(name, shape, std, mean=0.0):
        sd[prefix + name] = hg_normal(shape, _seed_of(seed, prefix + name), std, mean)

    put("attn.in_proj_weight", (3 * width, width), attn_std)
    put("attn.in_proj_bias", (3 * width,), 0.02)
    put("attn.out_proj.weight", (width, width), proj_std)
    put("attn.out_proj.bias", (width,), 0.02)
    put("ln_1.weight", (width,), ln_jitter, 1.0)
    put("ln_1.bias", (width,), ln_jitter)
    put("mlp.c_fc.weight", (4 * width, width), fc_std)
    put("mlp.c_fc.bias", (4 * width,), 0.02)
    put("mlp.c_proj.weight", (width, 4 * width), proj_std)
    put("mlp.c_proj.bias", (width,), 0.02)
    put("ln_2.weight", (width,), ln_jitter, 1.0)
    put("ln_2.bias", (width,), ln_jitter)


def clip_state_dict(cfg: dict = VIT_B16, seed: int = 0, ln_jitter: float = 0.1) -> "OrderedDict[str, np.ndarray]":
    """Synthetic CLIP state dict (numpy float32), 302 tensors for ViT-B/16.

    Biases and LayerNorm affine parameters are given non-trivial values (the reference initialises
    them to 0 / 1, which would hide bias- and affine-handling bugs).
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    vw, p = cfg["vision_width"], cfg["vision_patch_size"]
    grid = cfg["image_resolution"] // p
    tw = cfg["transformer_width"]
    scale = vw ** -0.5

    def put(name, shape, std, mean=0.0):
        sd[name] = hg_normal(shape, _seed_of(seed, name), std, mean)

    put("visual.class_embedding", (vw,), scale)
    put("visual.positional_embedding", (grid * grid + 1, vw), scale)
    put("visual.proj", (vw, cfg["embed_dim"]), scale)
    put("visual.conv1.weight", (vw, 3, p, p), (3 * p * p) ** -0.5)
    put("visual.ln_pre.weight", (vw,), ln_jitter, 1.0)
    put("visual.ln_pre.bias", (vw,), ln_jitter)
    for i in range(cfg["vision_layers"]):
        _block(sd, f"visual.transformer.resblocks.{i}.", vw, cfg["vision_layers"], seed, ln_jitter)
    put("visual.ln_post.weight", (vw,), ln_jitter, 1.0)
    put("visual.ln_post.bias", (vw,), ln_jitter)

    put("positional_embedding", (cfg["context_length"], tw), 0.01)
    put("text_projection", (tw, cfg["embed_dim"]), tw ** -0.5)
    sd["logit_scale"] = np.array(np.log(1 / 0.07), dtype=np.float32)
    for i in range(cfg["transformer_layers"]):
        _block(sd, f"transformer.resblocks.{i}.", tw, cfg["transformer_layers"], seed, ln_jitter)
    put("token_embedding.weight", (cfg["vocab_size"], tw), 0.02)
    put("ln_final.weight", (tw,), ln_jitter, 1.0)
    put("ln_final.bias", (tw,), ln_jitter)
    return sd


def stress_clip_state_dict(cfg: dict = VIT_B16, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """``clip_state_dict`` with what trained CLIP checkpoints have and random init lacks (SURVEY.md §7 "hard parts"):
    a few residual channels 50-100x larger than the rest, appearing both at the input of the towers and in the
    middle of the network, and MLP hidden units whose pre-activations reach the hundreds (QuickGELU / fp16 range).
    Deterministic; used by tests/golden/make_golden_stress.py (reference outputs) and the GPU test."""
    sd = clip_state_dict(cfg, seed)
    vw, tw = cfg["vision_width"], cfg["transformer_width"]
    vch = [5 % vw, (vw // 6 + 2) % vw, (2 * vw // 3 + 5) % vw]            # 5, 130, 517 at width 768
    tch = [7 % tw, (tw // 2 + 44) % tw, (tw - 3) % tw]
    sd["visual.ln_pre.bias"][vch[0]] += 60.0                              # outlier from the first block on
    sd["visual.ln_pre.weight"][vch[1]] *= 50.0                            # ... with a data-dependent sign / size
    vl, tl = cfg["vision_layers"], cfg["transformer_layers"]
    sd[f"visual.transformer.resblocks.{vl // 3}.mlp.c_proj.bias"][vch[2]] -= 80.0     # appears mid-network
    sd[f"visual.transformer.resblocks.{vl // 2}.attn.out_proj.bias"][vch[0]] -= 30.0
    for i in (0, vl // 2, vl - 1):                                        # hidden units with huge pre-activations
        w = sd[f"visual.transformer.resblocks.{i}.mlp.c_fc.weight"]
        w[11] *= 40.0
        w[(4 * vw) // 2 + 3] *= -60.0
        sd[f"visual.transformer.resblocks.{i}.mlp.c_fc.bias"][11] += 25.0
    sd["positional_embedding"][:, tch[0]] += 2.0                          # text: no pre-LayerNorm, x = tok + pos ~ 0.03
    sd["positional_embedding"][:, tch[1]] -= 3.0
    sd[f"transformer.resblocks.{tl // 3}.mlp.c_proj.bias"][tch[2]] += 4.0
    for i in (0, tl // 2, tl - 1):
        w = sd[f"transformer.resblocks.{i}.mlp.c_fc.weight"]
        w[13] *= 40.0
        w[(4 * tw) // 2 + 1] *= -60.0
    return sd


# ---------------------------------------------------------------------------------------------
# Adapter (variant C) parameters: /root/reference/CLIP_models_adapter_prior2.py:142-181
# ---------------------------------------------------------------------------------------------

def adapter_state_dict(cfg: dict = VIT_B16, seed: int = 1, layers=None, bottleneck: int = 64,
                       trained: bool = True, num_layers: int = 1) -> "OrderedDict[str, np.ndarray]":
    """Keys under ``visual.transformer.resblocks.{i}.adaptermlp.*``.

    ``trained=True`` gives non-zero ``up_proj`` and a visible ``scale`` so the adapter branch
    contributes (the reference zero-initialises ``up_proj`` and sets scale=1e-9, which makes the
    adapter an exact no-op: SURVEY.md §2.3).
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    vw = cfg["vision_width"]
    layers = range(cfg["vision_layers"]) if layers is None else layers
    d = bottleneck
    for i in layers:
        pre = f"visual.transformer.resblocks.{i}.adaptermlp."

        def put(name, shape, std, mean=0.0):
            sd[pre + name] = hg_normal(shape, _seed_of(seed, pre + name), std, mean)

        if trained:
            put("scale", (vw,), 0.02, 0.1)
            put("up_proj.weight", (vw, d), d ** -0.5)
            put("up_proj.bias", (vw,), 0.02)
        else:
            sd[pre + "scale"] = np.full((vw,), 1e-9, np.float32)
            sd[pre + "up_proj.weight"] = np.zeros((vw, d), np.float32)
            sd[pre + "up_proj.bias"] = np.zeros((vw,), np.float32)
        put("down_proj.weight", (d, vw), vw ** -0.5)
        put("down_proj.bias", (d,), 0.02)
        for tw in [f"mhsa_layers.{z}." for z in range(num_layers)] + ["mhsa."]:
            put(tw + "multihead_attn.in_proj_weight", (3 * d, d), d ** -0.5)
            put(tw + "multihead_attn.in_proj_bias", (3 * d,), 0.02)
            put(tw + "multihead_attn.out_proj.weight", (d, d), d ** -0.5)
            put(tw + "multihead_attn.out_proj.bias", (d,), 0.02)
            put(tw + "linear1.weight", (2 * d, d), d ** -0.5)
            put(tw + "linear1.bias", (2 * d,), 0.02)
            put(tw + "linear2.weight", (d, 2 * d), (2 * d) ** -0.5)
            put(tw + "linear2.bias", (d,), 0.02)
            for n in ("norm1", "norm2", "norm3"):
                put(tw + n + ".weight", (d,), 0.1, 1.0)
                put(tw + n + ".bias", (d,), 0.1)
    return sd


# ---------------------------------------------------------------------------------------------
# CoOp-VAE parameters: /root/reference/main_coop_vae.py:261-296, finetune_ship.py:302-314
# ---------------------------------------------------------------------------------------------

def encoder_state_dict(seed: int = 2, dim: int = 512, hidden: int = 2048, wstd: float = 0.02):
    sd = OrderedDict()
    sd["net.0.weight"] = hg_normal((hidden, dim), _seed_of(seed, "e.net.0.weight"), wstd)
    sd["net.0.bias"] = hg_normal((hidden,), _seed_of(seed, "e.net.0.bias"), 0.01)
    sd["mean.weight"] = hg_normal((dim, hidden), _seed_of(seed, "e.mean.weight"), wstd)
    sd["mean.bias"] = hg_normal((dim,), _seed_of(seed, "e.mean.bias"), 0.01)
    sd["log_var.weight"] = hg_normal((dim, hidden), _seed_of(seed, "e.log_var.weight"), wstd)
    sd["log_var.bias"] = hg_normal((dim,), _seed_of(seed, "e.log_var.bias"), 0.01)
    return sd


def generator_state_dict(seed: int = 3, dim: int = 512, hidden: int = 4096, wstd: float = 0.02):
    sd = OrderedDict()
    sd["net.0.weight"] = hg_normal((hidden, dim), _seed_of(seed, "g.net.0.weight"), wstd)
    sd["net.0.bias"] = hg_normal((hidden,), _seed_of(seed, "g.net.0.bias"), 0.01)
    sd["net.2.weight"] = hg_normal((dim, hidden), _seed_of(seed, "g.net.2.weight"), wstd)
    sd["net.2.bias"] = hg_normal((dim,), _seed_of(seed, "g.net.2.bias"), 0.01)
    return sd


def mlp_net_state_dict(seed: int = 4, dim: int = 512):
    sd = OrderedDict()
    for i in (0, 2, 4):
        sd[f"net.{i}.weight"] = hg_normal((dim, dim), _seed_of(seed, f"m.net.{i}.weight"), dim ** -0.5)
        sd[f"net.{i}.bias"] = hg_normal((dim,), _seed_of(seed, f"m.net.{i}.bias"), 0.01)
    return sd


# ---------------------------------------------------------------------------------------------
# Inputs
# ---------------------------------------------------------------------------------------------

def crops(batch: int, resolution: int = 224, seed: int = 1234) -> np.ndarray:
    """N(0,1) crops [B,3,R,R]: post-normalisation statistics of the real pipeline."""
    return hg_normal((batch, 3, resolution, resolution), seed, 1.0)


def tiny_tokens(n: int, cfg: dict = TINY, seed: int = 7) -> np.ndarray:
    """Synthetic token ids for the tiny vocabulary: SOT=V-2, body, EOT=V-1 (the max id), zero pad."""
    L, V = cfg["context_length"], cfg["vocab_size"]
    out = np.zeros((n, L), np.int64)
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(n * L, dtype=np.uint64) + (np.uint64(seed) << np.uint64(40))).reshape(n, L)
    for i in range(n):
        body = 2 + int(h[i, 0] % np.uint64(L - 3))          # 2 .. L-2 body tokens
        out[i, 0] = V - 2
        out[i, 1:1 + body] = (h[i, 1:1 + body] % np.uint64(V - 3)).astype(np.int64) + 1
        out[i, 1 + body] = V - 1
    return out


def priors(batch: int, n: int = 14, dim: int = 64, n_pad: int = 4, seed: int = 99):
    """Synthetic prior tokens [B,N,64] and key-padding mask [B,N] (True = pad), last n_pad masked.

    ``get_prior`` (/root/reference/upt_tip_cache_model_free_finetune_distill3.py:1445-1539) is out of
    scope; its output contract is (priors, mask) as consumed at CLIP_models_adapter_prior2.py:187-195.
    """
    p = hg_normal((batch, n, dim), seed, 1.0)
    mask = np.zeros((batch, n), bool)
    for b in range(batch):
        k = (n_pad + b) % (n_pad + 1)      # 0..n_pad padded positions, varying per image
        if k:
            mask[b, n - k:] = True
    return p, mask


def to_torch(sd_np, device=None):
    """numpy state dict -> OrderedDict of torch tensors (optionally on ``device``)."""
    import torch

    out = OrderedDict()
    for k, v in sd_np.items():
        t = torch.from_numpy(np.array(v, copy=True))
        out[k] = t.to(device) if device is not None else t
    return out
