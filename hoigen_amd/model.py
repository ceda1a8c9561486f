"""CLIP façade: the reference's module tree (same attribute names, same ``state_dict`` keys and
shapes) whose forward passes run on the hand-written HIP kernels through the C ABI.

Mirrors, as one implementation with a switch, the two CLIP variants HOIGen actually runs:

* variant A — pristine OpenAI CLIP, /root/reference/clipnet/model.py:153-432
  (``VisionTransformer.forward(x) -> [B,E]``);
* variant C — adapter + prior, /root/reference/CLIP_models_adapter_prior2.py:142-203,423-506,774-984
  (``VisionTransformer.forward(x, prior) -> ([B,E], [B,E,g,g])``, adapters before every selected block).

The modules are parameter holders: nothing here computes with torch.  ``forward`` of the towers calls
``hg_encode_image`` / ``hg_encode_image_prior`` / ``hg_encode_text_ids`` / ``hg_encode_text_embeds``; if the
native library or a HIP device is missing a ``RuntimeError`` is raised (no CPU fallback).  Inference
only: outputs do not carry autograd history (SURVEY.md §8b "Ownership / errors / threading").
"""
from __future__ import annotations

import ctypes as C
import functools
import math
import random
import weakref
from collections import OrderedDict
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from . import _lib


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"hoigen_amd: {what} must be on a HIP device (got {t.device}); the hot path has no "
                           "CPU implementation")


def _tensors_in(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors_in(o)


def _inference_only(fn):
    """The HIP path records no autograd graph.  The reference back-propagates through two of these entry points
    (VAE training through the frozen text tower, main_coop_vae.py:465-471; adapter / prompt fine-tuning,
    main_tip_finetune.py:955-1031).  A caller that asks for gradients - grad mode on and either an input that
    requires grad, or a module in ``train()`` mode that owns trainable parameters (where the reference would also apply
    the adapters' dropout, CLIP_models_adapter_prior2.py:51-72) - gets an error instead of silently frozen
    parameters.  ``eval()`` modules and ``torch.no_grad()`` callers (every inference call site of the reference) pass."""

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if torch.is_grad_enabled():
            what = None
            if any(t.is_floating_point() and t.requires_grad for t in _tensors_in(list(args) + list(kwargs.values()))):
                what = "an input that requires grad"
            elif any(m.training and any(p.requires_grad for p in m.parameters())
                     for m in ([self] if isinstance(self, nn.Module) else getattr(self, "_guard_modules", ()))):
                what = "a module in train() mode with trainable parameters"
            if what is not None:
                raise RuntimeError(
                    f"hoigen_amd: {type(self).__name__}.{fn.__name__} was called with {what} while grad mode is on. "
                    "The HIP path is inference-only: it records no autograd graph and applies no train-mode dropout, so "
                    "parameters would silently receive no gradients.  Call it under torch.no_grad() / in eval() mode, or "
                    "use the reference's PyTorch modules for training (INTEGRATION.md, 'Training entry points').")
        return fn(self, *args, **kwargs)

    return wrapper


class _Ctx:
    """Owns one native context (weights + workspace) for one façade model on one device."""

    def __init__(self):
        self.handle = None
        self.device_index = None
        self.options = {}            # hg_set_option values, re-applied whenever the native context is (re)created

    def get(self, device: torch.device) -> int:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self.handle is None or self.device_index != idx:
            self.close()
            h = _lib.lib().hg_create(idx)
            if not h:
                raise RuntimeError(f"hoigen_amd: hg_create({idx}) failed - no HIP device; there is no CPU fallback")
            self.handle, self.device_index = h, idx
            for k, v in self.options.items():
                self.check(_lib.lib().hg_set_option(h, k.encode(), int(v)), f"hg_set_option({k})")
        return self.handle

    def set_option(self, key: str, value: int) -> None:
        """Behaviour option of this model's native context (include/hoigen_amd.h: hg_set_option), e.g. ``last_block_row0``,
        ``ln_fuse``, ``adapter_fuse``, ``adapter_fold``, ``chunk_rows``.  Survives a move to another device."""
        if value is None:      # (what get_option returned before a context existed: back to "nothing applied")
            if self.handle is None:
                self.options.pop(key, None)
            return
        if self.handle is not None:
            self.check(_lib.lib().hg_set_option(self.handle, key.encode(), int(value)), f"hg_set_option({key})")
        self.options[key] = int(value)

    def get_option(self, key: str):
        """Current value of a behaviour option: the native context's (hg_get_option) once it exists, else the value recorded for it,
        else None (= the library default, nothing applied yet).  ``set_option(key, get_option(key))`` restores what was in force;
        None is accepted by ``set_option`` for that purpose."""
        if self.handle is not None:
            v = _lib.C.c_int32()
            self.check(_lib.lib().hg_get_option(self.handle, key.encode(), _lib.C.byref(v)), f"hg_get_option({key})")
            return int(v.value)
        return self.options.get(key)

    def check(self, rc: int, what: str) -> None:
        if rc != 0:
            msg = _lib.lib().hg_last_error(self.handle)
            raise RuntimeError(f"hoigen_amd: {what} failed ({rc}): {msg.decode() if msg else '?'}")

    def close(self):
        if self.handle is not None:
            try:
                _lib.lib().hg_destroy(self.handle)
            except Exception:  # pragma: no cover - interpreter shutdown
                pass
            self.handle = None

    def __del__(self):
        self.close()

    def __deepcopy__(self, memo):      # a copied module gets its own native context (with the same options)
        c = _Ctx()
        c.options = dict(self.options)
        return c

    def __getstate__(self):            # never pickle a native handle
        return {"options": dict(self.options)}

    def __setstate__(self, state):
        self.handle, self.device_index = None, None
        self.options = dict(state.get("options", {})) if isinstance(state, dict) else {}


def _sig(params) -> tuple:
    return tuple((p.data_ptr(), p._version, p.dtype) for p in params)


# ---------------------------------------------------------------------------------------------
# parameter holders (names = the reference's attribute names)
# ---------------------------------------------------------------------------------------------

class LayerNorm(nn.Module):
    """clipnet/model.py:153-159 (fp32 statistics).  Parameters ``weight``, ``bias``."""

    def __init__(self, dim: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.normalized_shape = (dim,)
        self.eps = 1e-5


class Linear(nn.Module):
    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class Conv2dParams(nn.Module):
    """visual.conv1 (bias-free stride-p conv, clipnet/model.py:207)."""

    def __init__(self, out_ch: int, patch: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_ch, 3, patch, patch))
        nn.init.normal_(self.weight, std=(3 * patch * patch) ** -0.5)


class QuickGELU(nn.Module):
    """clipnet/model.py:162-164 — fused into the c_fc GEMM epilogue; no parameters."""


class MultiheadAttention(nn.Module):
    """Parameters of nn.MultiheadAttention(d, h): in_proj_weight [3d,d], in_proj_bias, out_proj."""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)


class DecoderLayer(nn.Module):
    """TransformerDecoderLayer(64, 2, 128) of the adapter (CLIP_models_adapter_prior2.py:27-45)."""

    def __init__(self, d_model: int, nhead: int, dim_ff: int):
        super().__init__()
        self.multihead_attn = MultiheadAttention(d_model, nhead)
        self.linear1 = Linear(d_model, dim_ff)
        self.linear2 = Linear(dim_ff, d_model)
        self.norm1 = LayerNorm(d_model)   # present in the state dict, unused by forward_post
        self.norm2 = LayerNorm(d_model)
        self.norm3 = LayerNorm(d_model)


class Adapter(nn.Module):
    """CLIP_models_adapter_prior2.py:142-181 with the reference's constructor arguments
    (bottleneck 64, init 'lora', learnable per-channel scale initialised to 1e-9)."""

    def __init__(self, d_model: int, bottleneck: int = 64, adapter_num_layers: int = 1):
        super().__init__()
        if adapter_num_layers < 1:
            raise ValueError("adapter_num_layers must be >= 1")
        self.adapter_num_layers = adapter_num_layers
        self.n_embd, self.down_size = d_model, bottleneck
        self.scale = nn.Parameter(torch.ones(d_model) * 1e-9)
        self.down_proj = Linear(d_model, bottleneck)
        self.up_proj = Linear(bottleneck, d_model)
        with torch.no_grad():
            nn.init.zeros_(self.up_proj.weight)
            nn.init.zeros_(self.down_proj.bias)
            nn.init.zeros_(self.up_proj.bias)
        # _get_clones(layer, N): N layers applied one after the other on the prior path (adapter...:179,190-195)
        self.mhsa_layers = nn.ModuleList([DecoderLayer(bottleneck, 2, bottleneck * 2) for _ in range(adapter_num_layers)])
        self.mhsa = DecoderLayer(bottleneck, 2, bottleneck * 2)


class _MLP(nn.Module):
    def __init__(self, d_model: int):
        super().__init__()
        self.c_fc = Linear(d_model, d_model * 4)
        self.gelu = QuickGELU()
        self.c_proj = Linear(d_model * 4, d_model)


class ResidualAttentionBlock(nn.Module):
    """clipnet/model.py:167-188; variant C adds ``adaptermlp`` (CLIP_models_adapter_prior2.py:423-459)."""

    def __init__(self, d_model: int, n_head: int, adapter: bool = False, adapter_num_layers: int = 1):
        super().__init__()
        self.attn = MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = _MLP(d_model)
        self.ln_2 = LayerNorm(d_model)
        self.adapter = adapter
        if adapter:
            self.adaptermlp = Adapter(d_model, 64, adapter_num_layers)


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int, adapter_layers: Optional[List[int]] = None,
                 adapter_num_layers: int = 1):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        al = set(adapter_layers or [])
        self.resblocks = nn.ModuleList(
            [ResidualAttentionBlock(width, heads, i in al, adapter_num_layers) for i in range(layers)])


def _block_struct(blk: ResidualAttentionBlock) -> _lib.hg_block_weights:
    t = _lib.tensor
    return _lib.hg_block_weights(
        t(blk.attn.in_proj_weight), t(blk.attn.in_proj_bias), t(blk.attn.out_proj.weight), t(blk.attn.out_proj.bias),
        t(blk.ln_1.weight), t(blk.ln_1.bias), t(blk.mlp.c_fc.weight), t(blk.mlp.c_fc.bias),
        t(blk.mlp.c_proj.weight), t(blk.mlp.c_proj.bias), t(blk.ln_2.weight), t(blk.ln_2.bias))


def _decoder_struct(d: DecoderLayer) -> _lib.hg_decoder_layer_weights:
    t = _lib.tensor
    a = d.multihead_attn
    return _lib.hg_decoder_layer_weights(
        t(a.in_proj_weight), t(a.in_proj_bias), t(a.out_proj.weight), t(a.out_proj.bias), t(d.linear1.weight),
        t(d.linear1.bias), t(d.linear2.weight), t(d.linear2.bias), t(d.norm2.weight), t(d.norm2.bias),
        t(d.norm3.weight), t(d.norm3.bias))


def _adapter_structs(blocks) -> "C.Array":
    arr = (_lib.hg_adapter_weights * len(blocks))()
    keep = []                                   # the extra-layer arrays must outlive the native call
    for i, blk in enumerate(blocks):
        if getattr(blk, "adapter", False):
            a = blk.adaptermlp
            t = _lib.tensor
            extra = list(a.mhsa_layers)[1:]
            earr = (_lib.hg_decoder_layer_weights * len(extra))(*[_decoder_struct(d) for d in extra]) if extra else None
            keep.append(earr)
            arr[i] = _lib.hg_adapter_weights(1, a.down_size, t(a.scale), t(a.down_proj.weight), t(a.down_proj.bias),
                                             t(a.up_proj.weight), t(a.up_proj.bias), _decoder_struct(a.mhsa_layers[0]),
                                             _decoder_struct(a.mhsa), len(extra), earr)
    arr._keep = keep
    return arr


class VisionTransformer(nn.Module):
    """clipnet/model.py:202-236; with ``returns_local=True`` the variant-C contract
    (CLIP_models_adapter_prior2.py:471-506)."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int,
                 returns_local: bool = False, adapter_layers: Optional[List[int]] = None,
                 adapter_num_layers: int = 1):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        self.returns_local = returns_local
        self.conv1 = Conv2dParams(width, patch_size)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, adapter_layers, adapter_num_layers)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._ctx = _Ctx()
        self._loaded_sig = None
        self._adapter_sig = None

    # -- weight hand-over -------------------------------------------------------------------
    def _base_params(self):
        return [p for n, p in self.named_parameters() if "adaptermlp" not in n]

    def _adapter_params(self):
        return [p for n, p in self.named_parameters() if "adaptermlp" in n]

    def _sync(self, device: torch.device) -> int:
        h = self._ctx.get(device)
        sig = _sig(self._base_params())
        asig = _sig(self._adapter_params())
        blocks = list(self.transformer.resblocks)
        if sig != self._loaded_sig:
            barr = (_lib.hg_block_weights * len(blocks))(*[_block_struct(b) for b in blocks])
            aarr = _adapter_structs(blocks)
            t = _lib.tensor
            w = _lib.hg_vit_weights(
                self.transformer.width, len(blocks), self.transformer.heads, self.patch_size, self.input_resolution,
                self.output_dim, t(self.conv1.weight), t(self.class_embedding), t(self.positional_embedding),
                t(self.ln_pre.weight), t(self.ln_pre.bias), t(self.ln_post.weight), t(self.ln_post.bias), t(self.proj),
                barr, aarr if any(getattr(b, "adapter", False) for b in blocks) else None)
            self._ctx.check(_lib.lib().hg_load_vit(h, C.byref(w)), "hg_load_vit")
            self._loaded_sig, self._adapter_sig = sig, asig
        elif asig != self._adapter_sig:
            aarr = _adapter_structs(blocks)
            self._ctx.check(_lib.lib().hg_update_adapters(h, aarr, len(blocks)), "hg_update_adapters")
            self._adapter_sig = asig
        return h

    # -- forward ------------------------------------------------------------------------------
    @_inference_only
    @torch.no_grad()
    def forward(self, x: torch.Tensor, prior=None):
        _require_cuda(x, "image batch")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.input_resolution or x.shape[3] != self.input_resolution:
            raise RuntimeError(f"hoigen_amd: expected images [B,3,{self.input_resolution},{self.input_resolution}], "
                               f"got {tuple(x.shape)}")
        out_dtype = self.conv1.weight.dtype
        h = self._sync(x.device)
        xf = x.detach().to(torch.float32).contiguous()
        B, E = xf.shape[0], self.output_dim
        s = _stream_ptr(x.device)
        glob = torch.empty(B, E, device=x.device, dtype=torch.float32)
        if not self.returns_local:
            if prior is not None:
                raise RuntimeError("hoigen_amd: this VisionTransformer (variant A) takes no prior")
            self._ctx.check(_lib.lib().hg_encode_image(h, xf.data_ptr(), B, glob.data_ptr(), s), "hg_encode_image")
            return glob.to(out_dtype)
        g = self.input_resolution // self.patch_size
        local = torch.empty(B, E, g, g, device=x.device, dtype=torch.float32)
        pp = mp = None
        N = 0
        if prior is not None:
            pri, mask = prior
            _require_cuda(pri, "prior tokens")
            if pri.dim() != 3 or pri.shape[0] != B or pri.shape[2] != 64:
                raise RuntimeError(f"hoigen_amd: priors must be [B,N,64], got {tuple(pri.shape)}")
            N = pri.shape[1]
            pri_f = pri.detach().to(torch.float32).contiguous()
            m8 = (mask if mask is not None else torch.zeros(B, N, dtype=torch.bool, device=x.device))
            m8 = m8.to(device=x.device, dtype=torch.uint8).contiguous()
            pp, mp = pri_f.data_ptr(), m8.data_ptr()
        self._ctx.check(_lib.lib().hg_encode_image_prior(h, xf.data_ptr(), pp, mp, B, N, glob.data_ptr(),
                                                         local.data_ptr(), s), "hg_encode_image_prior")
        return glob.to(out_dtype), local.to(out_dtype)

    @_inference_only
    @torch.no_grad()
    def encode_into(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """Variant A ``forward`` writing the fp32 embeddings into the caller's contiguous ``out [B,E]`` (e.g. this
        rank's slice of a pre-allocated all-gather buffer): no allocation, no dtype round trip."""
        _require_cuda(x, "image batch")
        if self.returns_local:
            raise RuntimeError("hoigen_amd: encode_into is the variant-A (global embedding) path")
        B, E = x.shape[0], self.output_dim
        if out.dtype != torch.float32 or tuple(out.shape) != (B, E) or not out.is_contiguous() or out.device != x.device:
            raise RuntimeError(f"hoigen_amd: out must be a contiguous fp32 [{B},{E}] tensor on {x.device}")
        h = self._sync(x.device)
        xf = x.detach().to(torch.float32).contiguous()
        self._ctx.check(_lib.lib().hg_encode_image(h, xf.data_ptr(), B, out.data_ptr(), _stream_ptr(x.device)),
                        "hg_encode_image")
        return out

    def set_option(self, key: str, value: int) -> None:
        """Behaviour option of this tower's native context (include/hoigen_amd.h: hg_set_option)."""
        self._ctx.set_option(key, value)

    def get_option(self, key: str):
        """The option's current value.  On a HIP device the native context is created if need be and asked (hg_get_option); a model still
        on the CPU answers with the value recorded for it or None (= library default: set_option(key, None) is a no-op then)."""
        dev = self.positional_embedding.device
        if dev.type == "cuda":
            self._ctx.get(dev)
        return self._ctx.get_option(key)

    @torch.no_grad()
    def forward_trace(self, x: torch.Tensor):
        """Test hook: (embedding [B,E], CLS rows after ln_pre and every block [layers+1,B,D])."""
        _require_cuda(x, "image batch")
        h = self._sync(x.device)
        xf = x.detach().to(torch.float32).contiguous()
        B = xf.shape[0]
        out = torch.empty(B, self.output_dim, device=x.device, dtype=torch.float32)
        tr = torch.empty(self.transformer.layers + 1, B, self.transformer.width, device=x.device, dtype=torch.float32)
        self._ctx.check(_lib.lib().hg_encode_image_trace(h, xf.data_ptr(), B, out.data_ptr(), tr.data_ptr(),
                                                         _stream_ptr(x.device)), "hg_encode_image_trace")
        return out, tr


class TokenEmbedding(nn.Module):
    """nn.Embedding stand-in (``clip_model.token_embedding(ids)``, main_coop_vae.py:85,111)."""

    def __init__(self, vocab: int, dim: int, owner: "CLIP"):
        super().__init__()
        self.num_embeddings, self.embedding_dim = vocab, dim
        self.weight = nn.Parameter(torch.empty(vocab, dim))
        nn.init.normal_(self.weight, std=0.02)
        object.__setattr__(self, "_owner", owner)

    @torch.no_grad()
    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        owner: CLIP = self._owner
        dev = self.weight.device
        _require_cuda(self.weight, "CLIP model")
        h = owner._sync_text(dev)
        flat = ids.detach().to(device=dev, dtype=torch.int32).contiguous().view(-1)
        out = torch.empty(flat.numel(), self.embedding_dim, device=dev, dtype=torch.float32)
        owner._ctx.check(_lib.lib().hg_token_embedding(h, flat.data_ptr(), flat.numel(), out.data_ptr(),
                                                       _stream_ptr(dev)), "hg_token_embedding")
        return out.view(*ids.shape, self.embedding_dim).to(self.weight.dtype)


class CLIP(nn.Module):
    """clipnet/model.py:239-368 / CLIP_models_adapter_prior2.py:774-911 (ViT towers only)."""

    def __init__(self, embed_dim: int, image_resolution: int, vision_layers: Union[Tuple[int, int, int, int], int],
                 vision_width: int, vision_patch_size: int, context_length: int, vocab_size: int,
                 transformer_width: int, transformer_heads: int, transformer_layers: int,
                 variant_c: bool = False, adapter_layers: Optional[List[int]] = None, adapter_num_layers: int = 1):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("hoigen_amd: ModifiedResNet towers are outside the hot path (HOIGen uses ViT-B/16)")
        self.context_length = context_length
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim, returns_local=variant_c,
                                        adapter_layers=adapter_layers, adapter_num_layers=adapter_num_layers)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = TokenEmbedding(vocab_size, transformer_width, self)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.initialize_parameters()
        self._ctx = _Ctx()
        self._text_sig = None
        #: run the causal text tower only up to max(EOT)+1 positions (identical selected outputs)
        self.truncate_text = True
        self._trunc_memo = (None, 0, 0, 0)

    def initialize_parameters(self):
        """clipnet/model.py:295-322."""
        nn.init.normal_(self.positional_embedding, std=0.01)
        for tower in (self.transformer, self.visual.transformer):
            proj_std = (tower.width ** -0.5) * ((2 * tower.layers) ** -0.5)
            attn_std = tower.width ** -0.5
            fc_std = (2 * tower.width) ** -0.5
            for block in tower.resblocks:
                nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
                nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
                nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
                nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)

    def build_attention_mask(self):
        """clipnet/model.py:324-330 (kept for API parity; the kernel applies the causal mask itself)."""
        mask = torch.empty(self.context_length, self.context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    # -- text weights -----------------------------------------------------------------------------
    def _text_params(self):
        return ([self.token_embedding.weight, self.positional_embedding, self.ln_final.weight, self.ln_final.bias,
                 self.text_projection] + list(self.transformer.parameters()))

    def _sync_text(self, device: torch.device) -> int:
        h = self._ctx.get(device)
        sig = _sig(self._text_params())
        if sig != self._text_sig:
            blocks = list(self.transformer.resblocks)
            barr = (_lib.hg_block_weights * len(blocks))(*[_block_struct(b) for b in blocks])
            t = _lib.tensor
            w = _lib.hg_text_weights(self.transformer.width, len(blocks), self.transformer.heads, self.context_length,
                                     self.vocab_size, self.text_projection.shape[1], t(self.token_embedding.weight),
                                     t(self.positional_embedding), t(self.ln_final.weight), t(self.ln_final.bias),
                                     t(self.text_projection), barr)
            self._ctx.check(_lib.lib().hg_load_text(h, C.byref(w)), "hg_load_text")
            self._text_sig = sig
        return h

    def _trunc_len(self, tokens: torch.Tensor, eot: Optional[torch.Tensor] = None) -> int:
        """max(EOT) + 1 over the call.  The grid depends on it, so the host has to read it: one tiny device->host
        copy, remembered for the token tensor it was computed from (the sampling loop passes the same
        ``tokenized_prompts`` every iteration, main_tip_finetune.py:759-824)."""
        # Inference tensors (torch.inference_mode) track no version counter: no memo for them.  The memo is keyed on
        # the live object, its version and its storage; a mutation that bumps none of them (torch.from_numpy shared
        # memory, .data writes) cannot be seen here - the native side clamps EOT into the truncated length and
        # reports HG_ERR_INVALID on the next call (hg_encode_text_ids), so a stale value never reads out of bounds.
        try:
            ver_now = tokens._version
        except RuntimeError:
            ver_now = None
        ref, ver, ptr, n = self._trunc_memo
        if (ver_now is not None and ref is not None and ref() is tokens and ver == ver_now
                and ptr == tokens.data_ptr()):      # the very same live tensor object
            return n
        if eot is None:
            eot = tokens.argmax(dim=-1)
        n = int(eot.max().item()) + 1
        self._trunc_memo = (weakref.ref(tokens), ver_now, tokens.data_ptr(), n) if ver_now is not None else (None, 0, 0, 0)
        return n

    def set_option(self, key: str, value: int) -> None:
        """Behaviour option (hg_set_option) for both towers of this model."""
        self.visual.set_option(key, value)
        self._ctx.set_option(key, value)

    def get_option(self, key: str):
        """The option's current value in the text tower's context (see VisionTransformer.get_option)."""
        dev = self.positional_embedding.device
        if dev.type == "cuda":
            self._ctx.get(dev)
        return self._ctx.get_option(key)

    # -- public API -----------------------------------------------------------------------------------
    def encode_image(self, image: torch.Tensor):
        """clipnet/model.py:336-337 / CLIP_models_adapter_prior2.py:875-876."""
        return self.visual(image)

    @_inference_only
    @torch.no_grad()
    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """clipnet/model.py:339-352: text [T,L] integer ids (int64 or int32), EOT = largest id."""
        dev = self.positional_embedding.device
        _require_cuda(self.positional_embedding, "CLIP model")
        if text.dim() != 2 or text.shape[1] > self.context_length:
            raise RuntimeError(f"hoigen_amd: text must be [T,<= {self.context_length}], got {tuple(text.shape)}")
        h = self._sync_text(dev)
        T, L = text.shape
        trunc = 0
        if self.truncate_text and T > 0:
            trunc = self._trunc_len(text)
        ids = text.detach().to(device=dev, dtype=torch.int32).contiguous()
        out = torch.empty(T, self.text_projection.shape[1], device=dev, dtype=torch.float32)
        rc = _lib.lib().hg_encode_text_ids(h, ids.data_ptr(), T, L, out.data_ptr(), trunc, _stream_ptr(dev))
        if rc:
            self._trunc_memo = (None, 0, 0, 0)      # a stale truncation length must not survive the error it caused
        self._ctx.check(rc, "hg_encode_text_ids")
        return out.to(self.dtype)

    @_inference_only
    @torch.no_grad()
    def encode_text_embeds(self, prompts: torch.Tensor, tokenized_prompts: torch.Tensor) -> torch.Tensor:
        """TextEncoder.forward(prompts, tokenized_prompts) (main_coop_vae.py:54-63) -> fp32 [R,E]."""
        _require_cuda(prompts, "prompts")
        dev = prompts.device
        h = self._sync_text(dev)
        R, L, D = prompts.shape
        if D != self.transformer.width or L > self.context_length:
            raise RuntimeError(f"hoigen_amd: prompts must be [R,<= {self.context_length},{self.transformer.width}]")
        eot = tokenized_prompts.argmax(dim=-1)
        trunc = self._trunc_len(tokenized_prompts, eot) if (self.truncate_text and R > 0) else 0
        eot32 = eot.to(device=dev, dtype=torch.int32).contiguous()
        pf = prompts.detach().to(torch.float32).contiguous()
        out = torch.empty(R, self.text_projection.shape[1], device=dev, dtype=torch.float32)
        rc = _lib.lib().hg_encode_text_embeds(h, pf.data_ptr(), eot32.data_ptr(), R, L, out.data_ptr(), trunc, _stream_ptr(dev))
        if rc:
            self._trunc_memo = (None, 0, 0, 0)
        self._ctx.check(rc, "hg_encode_text_embeds")
        return out

    def forward(self, image, text):
        """clipnet/model.py:354-368."""
        image_features = self.encode_image(image)
        if isinstance(image_features, tuple):
            image_features = image_features[0]
        text_features = self.encode_text(text)
        image_features = image_features / image_features.norm(dim=-1, keepdim=True)
        text_features = text_features / text_features.norm(dim=-1, keepdim=True)
        logit_scale = self.logit_scale.exp()
        logits_per_image = logit_scale * image_features @ text_features.t()
        return logits_per_image, logits_per_image.t()


def convert_weights(model: nn.Module):
    """clipnet/model.py:371-392 — cast conv/linear/attention/projection parameters to fp16 (adapter
    parameters excluded: variant C never converts, CLIP_models_adapter_prior2.py:980)."""
    for name, p in model.named_parameters():
        if "adaptermlp" in name:
            continue
        leaf = name.split(".")[-1]
        parent = name.rsplit(".", 1)[0] if "." in name else ""
        is_ln = parent.endswith(("ln_1", "ln_2", "ln_pre", "ln_post", "ln_final"))
        if name in ("visual.proj", "text_projection") or (
                not is_ln and leaf in ("weight", "bias", "in_proj_weight", "in_proj_bias")
                and (".attn" in name or ".mlp." in name or name == "visual.conv1.weight")):
            p.data = p.data.half()


def _infer_config(state_dict: dict) -> dict:
    """Hyper-parameters from tensor shapes (clipnet/model.py:398-418)."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("hoigen_amd: only ViT checkpoints are supported (no 'visual.proj' in the state dict)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    transformer_width = state_dict["ln_final.weight"].shape[0]
    return dict(embed_dim=state_dict["text_projection"].shape[1], image_resolution=vision_patch_size * grid_size,
                vision_layers=vision_layers, vision_width=vision_width, vision_patch_size=vision_patch_size,
                context_length=state_dict["positional_embedding"].shape[0],
                vocab_size=state_dict["token_embedding.weight"].shape[0], transformer_width=transformer_width,
                transformer_heads=transformer_width // 64,
                transformer_layers=len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks"))))


def build_model(state_dict: dict, use_adapter: Optional[bool] = None, adapter_pos: str = "all",
                adapter_num_layers: int = 1):
    """``build_model(state_dict)`` of clipnet/model.py:395-432 (variant A: weights round-tripped through
    fp16, strict load, eval mode) — or, when ``use_adapter`` is given, of
    CLIP_models_adapter_prior2.py:934-984 (variant C: fp32, ``strict=False``, adapters per ``adapter_pos``).
    """
    state_dict = OrderedDict(state_dict)
    cfg = _infer_config(state_dict)
    for key in ("input_resolution", "context_length", "vocab_size"):
        state_dict.pop(key, None)
    if use_adapter is None:
        model = CLIP(**cfg)
        convert_weights(model)
        model.load_state_dict(state_dict)
        return model.eval()
    vl = cfg["vision_layers"]
    if adapter_pos == "all":
        layers = list(range(vl))
    elif adapter_pos == "front":
        layers = list(range(vl // 2))
    elif adapter_pos == "end":
        layers = list(range(vl // 2, vl))
    elif adapter_pos == "last":
        layers = list(range(vl - 1, vl))
    elif adapter_pos == "random":
        layers = [random.randint(0, vl - 1) for _ in range(vl // 2)]
    else:
        raise ValueError(f"unknown adapter_pos {adapter_pos!r}")
    model = CLIP(**cfg, variant_c=True, adapter_layers=layers if use_adapter else [],
                 adapter_num_layers=adapter_num_layers)
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    print("[INFO] missing_keys:", [k for k in missing if "adaptermlp" not in k])
    print("[INFO] unexpected_keys:", unexpected)
    return model
