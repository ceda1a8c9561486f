"""CPU ORACLE — test infrastructure only, never the product path.

An independent restatement (plain tensor algebra on CPU; no nn.Module, no nn.MultiheadAttention,
no F.scaled_dot_product_attention, no F.conv2d) of the reference's CLIP hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product (``hoigen_amd``) never does and fails loudly without its HIP library.

Pinning: ``tests/golden/make_golden.py`` imports the *reference itself* from /root/reference in
the build container, runs it on the synthetic state dicts of ``hoigen_amd.synth`` and stores its
outputs under ``tests/golden/``; ``tests/test_oracle_vs_golden.py`` checks every function below
against those fixtures (fp32: <=2e-5 relative; integer artefacts bit-exact).  The reference has no
golden vectors of its own for this path and no pretrained weights are reachable, so
behaviour on the real ``ViT-B-16.pt`` weights is **parity unpinned**.

Every function cites the reference lines it restates (paths relative to /root/reference).
All functions take a ``dtype`` (torch.float32 to mirror the reference's CPU path, torch.float64 for
a higher-precision yardstick) and a state dict ``sd`` of CPU tensors keyed as in SURVEY.md §8b.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

# Keys the reference casts to fp16 in build_model -> convert_weights (clipnet/model.py:371-392):
# Conv/Linear weight+bias, MultiheadAttention in_proj_*, `proj`, `text_projection`.
_FP16_SUFFIXES = ("conv1.weight", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                  "attn.out_proj.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
                  "mlp.c_proj.bias")
_FP16_EXACT = ("visual.proj", "text_projection")


def reference_weight_rounding(sd_np: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
    """What the reference's weights are after ``build_model(sd)`` + ``.float()`` on CPU.

    clipnet/model.py:430 (convert_weights before load_state_dict) rounds the listed tensors to fp16;
    clipnet/clip.py:135-136 then casts everything back to fp32.  Adapter tensors (variant C,
    CLIP_models_adapter_prior2.py:980 has convert_weights commented out) are untouched — callers
    pass variant-C state dicts through ``as_tensors`` instead.
    """
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).float()
        if "adaptermlp" not in k and (k.endswith(_FP16_SUFFIXES) or k in _FP16_EXACT):
            t = t.half().float()
        out[k] = t
    return out


def as_tensors(sd_np: Dict[str, np.ndarray]) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in sd_np.items()}


# ------------------------------------------------------------------------------------------
# elementary ops
# ------------------------------------------------------------------------------------------

def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """clipnet/model.py:153-159 — nn.LayerNorm over the last dim, biased variance, eps 1e-5."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w.to(x.dtype) + b.to(x.dtype)


def quick_gelu(x: Tensor) -> Tensor:
    """clipnet/model.py:162-164 — x * sigmoid(1.702 x)."""
    return x / (1.0 + torch.exp(-1.702 * x))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """F.linear: x @ w^T + b with w [out,in]."""
    y = x @ w.to(x.dtype).transpose(-1, -2)
    return y if b is None else y + b.to(x.dtype)


def attention(x: Tensor, sd: Dict[str, Tensor], pre: str, heads: int, causal: bool) -> Tensor:
    """clipnet/model.py:171,181-183 — nn.MultiheadAttention(x,x,x) with packed in-proj.

    x is [B,L,D] (the reference runs sequence-first [L,B,D]; the maths is per batch element so the
    layout is immaterial).  Row blocks of in_proj_weight are ordered q|k|v; head h owns channels
    64h..64h+63; scores scaled by head_dim**-0.5; additive -inf mask strictly above the diagonal
    for the text tower (clipnet/model.py:324-330).
    """
    B, L, D = x.shape
    hd = D // heads
    qkv = linear(x, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"])
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    out = torch.empty_like(x)
    tri = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1) if causal else None
    for h in range(heads):
        sl = slice(h * hd, (h + 1) * hd)
        s = (q[..., sl] @ k[..., sl].transpose(1, 2)) * (hd ** -0.5)          # [B,L,L]
        if tri is not None:
            s = s.masked_fill(tri, float("-inf"))
        s = s - s.max(dim=-1, keepdim=True).values
        p = torch.exp(s)
        p = p / p.sum(dim=-1, keepdim=True)
        out[..., sl] = p @ v[..., sl]
    return linear(out, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"])


def mlp(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    """clipnet/model.py:173-177 — c_fc, QuickGELU, c_proj."""
    u = quick_gelu(linear(x, sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"]))
    return linear(u, sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])


# ------------------------------------------------------------------------------------------
# adapter (variant C)
# ------------------------------------------------------------------------------------------

def _cross_attention(q_in: Tensor, mem: Tensor, sd: Dict[str, Tensor], pre: str, nhead: int,
                     key_padding_mask: Optional[Tensor]) -> Tensor:
    """nn.MultiheadAttention(64, 2)(query=tgt, key=memory, value=memory, key_padding_mask=mask) as
    used at CLIP_models_adapter_prior2.py:62-65 (eval mode: dropout off).  head_dim = 32."""
    B, Lq, D = q_in.shape
    hd = D // nhead
    w, b = sd[pre + "in_proj_weight"].to(q_in.dtype), sd[pre + "in_proj_bias"].to(q_in.dtype)
    q = q_in @ w[:D].T + b[:D]
    k = mem @ w[D:2 * D].T + b[D:2 * D]
    v = mem @ w[2 * D:].T + b[2 * D:]
    out = torch.empty_like(q)
    for h in range(nhead):
        sl = slice(h * hd, (h + 1) * hd)
        s = (q[..., sl] @ k[..., sl].transpose(1, 2)) * (hd ** -0.5)          # [B,Lq,Lk]
        if key_padding_mask is not None:
            s = s.masked_fill(key_padding_mask[:, None, :], float("-inf"))
        s = s - s.max(dim=-1, keepdim=True).values
        p = torch.exp(s)
        p = p / p.sum(dim=-1, keepdim=True)
        out[..., sl] = p @ v[..., sl]
    return linear(out, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def _decoder_layer_post(tgt: Tensor, mem: Tensor, sd: Dict[str, Tensor], pre: str,
                        mask: Optional[Tensor]) -> Tensor:
    """CLIP_models_adapter_prior2.py:51-72 forward_post: cross-attn, +res, norm2, FFN(relu), +res,
    norm3 (norm1 / self-attention are commented out in the reference)."""
    t2 = _cross_attention(tgt, mem, sd, pre + "multihead_attn.", 2, mask)
    tgt = layer_norm(tgt + t2, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    t2 = linear(torch.relu(linear(tgt, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])),
                sd[pre + "linear2.weight"], sd[pre + "linear2.bias"])
    return layer_norm(tgt + t2, sd[pre + "norm3.weight"], sd[pre + "norm3.bias"])


def adapter(x: Tensor, sd: Dict[str, Tensor], pre: str,
            prior: Optional[Tuple[Tensor, Tensor]]) -> Tensor:
    """CLIP_models_adapter_prior2.py:183-203 Adapter.forward, x [B,L,D] batch-first.

    down = relu(down_proj(x)); with a prior: one post-norm decoder layer (``mhsa_layers.0``) whose
    memory is the prior tokens [B,N,64] with key_padding_mask (True = pad); without: ``mhsa`` with
    memory = down itself; up_proj; times per-channel ``scale``.
    """
    down = torch.relu(linear(x, sd[pre + "down_proj.weight"], sd[pre + "down_proj.bias"]))
    if prior is not None:
        ctx, mask = prior
        z = 0                              # adapter_num_layers clones, applied one after the other (:179,190-195)
        while pre + f"mhsa_layers.{z}.linear1.weight" in sd:
            down = _decoder_layer_post(down, ctx.to(x.dtype), sd, pre + f"mhsa_layers.{z}.", mask)
            z += 1
    else:
        down = _decoder_layer_post(down, down, sd, pre + "mhsa.", None)
    up = linear(down, sd[pre + "up_proj.weight"], sd[pre + "up_proj.bias"])
    return up * sd[pre + "scale"].to(x.dtype)


# ------------------------------------------------------------------------------------------
# towers
# ------------------------------------------------------------------------------------------

def resblock(x: Tensor, sd: Dict[str, Tensor], pre: str, heads: int, causal: bool,
             prior=None, use_adapter: bool = False) -> Tensor:
    """clipnet/model.py:185-188; variant C prepends the adapter (CLIP_models_adapter_prior2.py:453-459)."""
    if use_adapter:
        x = x + adapter(x, sd, pre + "adaptermlp.", prior)
    x = x + attention(layer_norm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"]), sd, pre, heads, causal)
    x = x + mlp(layer_norm(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"]), sd, pre)
    return x


def n_layers(sd: Dict[str, Tensor], prefix: str) -> int:
    return len([k for k in sd if k.startswith(prefix) and k.endswith(".attn.in_proj_weight")])


def patchify(img: Tensor, p: int) -> Tensor:
    """Patch matrix of the stride-p conv (clipnet/model.py:220-222): token t = g*row + col,
    column index = c*p*p + ky*p + kx.  Pure indexing — bit-exact."""
    B, C, H, W = img.shape
    gh, gw = H // p, W // p
    a = img.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
    return a.reshape(B, gh * gw, C * p * p)


def vision_tokens(sd: Dict[str, Tensor], img: Tensor, dtype=torch.float32, prior=None,
                  adapter_layers: Sequence[int] = (), collect: Optional[list] = None) -> Tensor:
    """clipnet/model.py:219-229: patch GEMM, [cls; patches] + pos, ln_pre, transformer.  -> [B,L,D]"""
    w = sd["visual.conv1.weight"]
    width, p = w.shape[0], w.shape[-1]
    x = patchify(img.to(dtype), p) @ w.reshape(width, -1).to(dtype).T                # [B,g*g,D]
    cls = sd["visual.class_embedding"].to(dtype).expand(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].to(dtype)
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    if collect is not None:
        collect.append(x.clone())
    heads = width // 64
    if prior is not None:
        prior = (prior[0].to(dtype), prior[1])
    for i in range(n_layers(sd, "visual.")):
        x = resblock(x, sd, f"visual.transformer.resblocks.{i}.", heads, False, prior, i in adapter_layers)
        if collect is not None:
            collect.append(x.clone())
    return x


def encode_image(sd: Dict[str, Tensor], img: Tensor, dtype=torch.float32, collect=None) -> Tensor:
    """CLIP.encode_image, variant A (clipnet/model.py:231-236,336-337): ln_post(CLS) @ proj."""
    x = vision_tokens(sd, img, dtype, collect=collect)
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return x @ sd["visual.proj"].to(dtype)


def visual_with_prior(sd: Dict[str, Tensor], img: Tensor, prior=None,
                      adapter_layers: Sequence[int] = (), dtype=torch.float32):
    """Variant C VisionTransformer.forward (CLIP_models_adapter_prior2.py:489-506): ln_post and proj
    on all tokens; returns (global [B,E], local [B,E,g,g])."""
    B, _, H, W = img.shape
    p = sd["visual.conv1.weight"].shape[-1]
    x = vision_tokens(sd, img, dtype, prior, adapter_layers)
    x = layer_norm(x, sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"].to(dtype)
    return x[:, 0, :], x[:, 1:, :].reshape(B, H // p, W // p, -1).permute(0, 3, 1, 2)


def text_transformer(sd: Dict[str, Tensor], x: Tensor, collect=None) -> Tensor:
    """positional add + causal transformer + ln_final on embedded prompts x [T,L,D]
    (clipnet/model.py:342-346; main_coop_vae.py:55-59)."""
    D = x.shape[-1]
    x = x + sd["positional_embedding"].to(x.dtype)[: x.shape[1]]
    if collect is not None:
        collect.append(x.clone())
    for i in range(n_layers(sd, "transformer.")):
        x = resblock(x, sd, f"transformer.resblocks.{i}.", D // 64, True)
        if collect is not None:
            collect.append(x.clone())
    return layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])


def eot_index(tokens: Tensor) -> Tensor:
    """clipnet/model.py:350 — ``text.argmax(dim=-1)``: EOT (49407) is the largest id; INT, exact."""
    return tokens.argmax(dim=-1)


def encode_text(sd: Dict[str, Tensor], tokens: Tensor, dtype=torch.float32, collect=None) -> Tensor:
    """CLIP.encode_text (clipnet/model.py:339-352)."""
    x = sd["token_embedding.weight"].to(dtype)[tokens.long()]
    x = text_transformer(sd, x, collect)
    x = x[torch.arange(x.shape[0]), eot_index(tokens)]
    return x @ sd["text_projection"].to(dtype)


def text_encoder_embeds(sd: Dict[str, Tensor], prompts: Tensor, tokenized: Tensor,
                        dtype=torch.float32) -> Tensor:
    """TextEncoder.forward(prompts, tokenized_prompts) (main_coop_vae.py:54-63)."""
    x = text_transformer(sd, prompts.to(dtype))
    x = x[torch.arange(x.shape[0]), eot_index(tokenized)]
    return x @ sd["text_projection"].to(dtype)


def l2_normalize(x: Tensor) -> Tensor:
    """x / x.norm(dim=-1, keepdim=True) (main_coop_vae.py:438,466)."""
    return x / torch.sqrt((x * x).sum(dim=-1, keepdim=True))
