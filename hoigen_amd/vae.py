"""CoOp-VAE feature generator façade: the reference's classes with the same names, constructor
arguments, ``state_dict`` keys and call signatures, running on the HIP kernels.

Reference: /root/reference/main_coop_vae.py — ``TextEncoder`` :45-63, ``PromptLearner_{hoi,h,o}``
:66-258, ``Encoder`` :261-279, ``Generator`` :282-296, ``vae_loss`` :300-303; ``mlp_net``
/root/reference/finetune_ship.py:302-314 (twin main_tip_finetune.py:313-324).

Inference only (no autograd through the kernels); tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _lib
from . import clip as _clip
from .model import CLIP, Linear, _Ctx, _inference_only, _require_cuda, _sig, _stream_ptr


def weights_init(m):
    """main_coop_vae.py:32-39."""
    if isinstance(m, Linear):
        m.weight.data.normal_(0.0, 0.02)
        m.bias.data.fill_(0)


class _Slot:
    """One weight slot (``kind``: "vae" or "mlp") of the per-device context that every VAE-family module on that device shares:
    one workspace however many Encoder / Generator / VAE / mlp_net objects exist (the three-branch sampler of
    main_tip_finetune.py:759-824 holds nine).  The slot goes back to the pool when its module dies.  The shared context serves
    one call at a time: ``session`` holds the pool's lock for the duration of a module's load + forward (threads) and, when the
    calling stream differs from the one the context was last used on, makes it wait for that use (streams) - modules driven from
    different streams or threads stay correct, they just do not overlap.  At most ``HG_MAX_SLOTS`` live modules of a kind per device."""
    _lock = threading.RLock()      # re-entrant: a GC pass inside get() may finalise another module's slot on this thread
    _pools = {}          # device index -> {"ctx": _Ctx, "vae": [free slots], "mlp": [free slots], "stream": id, "event": Event}

    def __init__(self, kind: str):
        self.kind, self.idx, self.slot = kind, None, None

    def get(self, device: torch.device):
        """-> (context, handle, slot, fresh): ``fresh`` = the slot was just taken, its weights must be (re)loaded."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        fresh = False
        with _Slot._lock:
            if self.idx != idx:
                self._give()
                pool = _Slot._pools.get(idx)
                if pool is None:
                    pool = _Slot._pools[idx] = {"ctx": _Ctx(), "vae": list(range(_lib.HG_MAX_SLOTS)),
                                                "mlp": list(range(_lib.HG_MAX_SLOTS))}
                if not pool[self.kind]:
                    raise RuntimeError(f"hoigen_amd: more than {_lib.HG_MAX_SLOTS} live {self.kind} modules on cuda:{idx} "
                                       "(HG_MAX_SLOTS in include/hoigen_amd.h)")
                self.idx, self.slot, fresh = idx, pool[self.kind].pop(0), True
            ctx = _Slot._pools[idx]["ctx"]
        return ctx, ctx.get(device), self.slot, fresh

    def session(self, device: torch.device):
        """Context manager around one module call: (context, handle, slot, fresh) with the shared context reserved."""
        return _Session(self, device)

    def _give(self):
        if self.idx is not None and self.idx in _Slot._pools:
            _Slot._pools[self.idx][self.kind].append(self.slot)
        self.idx, self.slot = None, None

    def __del__(self):
        try:
            with _Slot._lock:
                self._give()
        except Exception:      # pragma: no cover - interpreter shutdown
            pass

    def __deepcopy__(self, memo):      # a copied module takes its own slot
        return _Slot(self.kind)

    def __getstate__(self):
        return {"kind": self.kind}

    def __setstate__(self, state):
        self.kind, self.idx, self.slot = state["kind"], None, None


def set_option(key: str, value: int, device=None) -> None:
    """Behaviour option (include/hoigen_amd.h: hg_set_option) of the native context that the VAE-family modules of ``device`` share,
    e.g. ``vae_fused`` (Encoder -> reparameterise -> Generator as one kernel: 1 where it pays, 2 every row, 0 the GEMM path) or
    ``chunk_rows``.  Survives re-creation of the context."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _Slot._lock:
        pool = _Slot._pools.get(idx)
        if pool is None:
            pool = _Slot._pools[idx] = {"ctx": _Ctx(), "vae": list(range(_lib.HG_MAX_SLOTS)), "mlp": list(range(_lib.HG_MAX_SLOTS))}
        pool["ctx"].set_option(key, value)      # (None = what get_option returned before a context existed: nothing to undo)


def get_option(key: str, device=None):
    """Current value of a behaviour option in the VAE-family context of ``device`` (None: library default, no context yet)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _Slot._lock:
        pool = _Slot._pools.get(idx)
        if pool is None:
            pool = _Slot._pools[idx] = {"ctx": _Ctx(), "vae": list(range(_lib.HG_MAX_SLOTS)), "mlp": list(range(_lib.HG_MAX_SLOTS))}
        pool["ctx"].get(dev)      # (the native context answers: created here if this is the first use of the device)
        return pool["ctx"].get_option(key)


class _Session:
    def __init__(self, slot: _Slot, device: torch.device):
        self.slot, self.device = slot, device

    def __enter__(self):
        _Slot._lock.acquire()
        try:
            got = self.slot.get(self.device)
            pool = _Slot._pools[self.slot.idx]
            cur = torch.cuda.current_stream(self.device)
            ev = pool.get("event")
            if ev is not None and pool.get("stream") != cur.cuda_stream:
                cur.wait_event(ev)              # the workspace's previous user ran on another stream
            self.pool, self.cur = pool, cur
            return got
        except BaseException:
            _Slot._lock.release()
            raise

    def __exit__(self, *exc):
        try:
            ev = self.pool.get("event")
            if ev is None:
                ev = self.pool["event"] = torch.cuda.Event()
            ev.record(self.cur)
            self.pool["stream"] = self.cur.cuda_stream
        finally:
            _Slot._lock.release()
        return False


class _Seq(nn.Module):
    """Numbered children like nn.Sequential (keys ``net.0.weight`` ...); ReLUs carry no parameters."""

    def __init__(self, layers: Sequence[Tuple[int, nn.Module]]):
        super().__init__()
        for i, m in layers:
            self.add_module(str(i), m)

    def __getitem__(self, i: int) -> nn.Module:
        return getattr(self, str(i))


def _f32(x: torch.Tensor) -> torch.Tensor:
    return x.detach().to(torch.float32).contiguous()


# ---------------------------------------------------------------------------------------------
class Encoder(nn.Module):
    """main_coop_vae.py:261-279: Linear(512,2048)+ReLU -> mean, log_var Linear(2048,512)."""

    def __init__(self, dim: int = 512, hidden: int = 2048):
        super().__init__()
        self.dim, self.hidden = dim, hidden
        self.net = _Seq([(0, Linear(dim, hidden))])
        self.mean = Linear(hidden, dim)
        self.log_var = Linear(hidden, dim)
        self.apply(weights_init)
        self._slot = _Slot("vae")
        self._sig = None

    def _weights(self, w: _lib.hg_vae_weights):
        t = _lib.tensor
        w.dim, w.enc_hidden = self.dim, self.hidden
        w.enc_w0, w.enc_b0 = t(self.net[0].weight), t(self.net[0].bias)
        w.enc_mean_w, w.enc_mean_b = t(self.mean.weight), t(self.mean.bias)
        w.enc_logvar_w, w.enc_logvar_b = t(self.log_var.weight), t(self.log_var.bias)

    @_inference_only
    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        _require_cuda(x, "Encoder input")
        with self._slot.session(x.device) as (ctx, h, slot, fresh):
            if fresh:
                self._sig = None
            sig = _sig(self.parameters())
            if sig != self._sig:
                w = _lib.hg_vae_weights()
                self._weights(w)
                ctx.check(_lib.lib().hg_load_vae(h, slot, C.byref(w)), "hg_load_vae")
                self._sig = sig
            xf = _f32(x)
            R = xf.shape[0]
            mean, logvar = torch.empty_like(xf), torch.empty_like(xf)
            zeros = torch.zeros_like(xf)          # eps = 0: z is not requested
            ctx.check(_lib.lib().hg_vae_forward(h, slot, xf.data_ptr(), zeros.data_ptr(), R, mean.data_ptr(),
                                                      logvar.data_ptr(), None, None, _stream_ptr(x.device)),
                            "hg_vae_forward")
            return mean, logvar


class Generator(nn.Module):
    """main_coop_vae.py:282-296: Linear(512,4096) -> ReLU -> Linear(4096,512)."""

    def __init__(self, dim: int = 512, hidden: int = 4096):
        super().__init__()
        self.dim, self.hidden = dim, hidden
        self.net = _Seq([(0, Linear(dim, hidden)), (2, Linear(hidden, dim))])
        self.apply(weights_init)
        self._slot = _Slot("vae")
        self._sig = None

    def _weights(self, w: _lib.hg_vae_weights):
        t = _lib.tensor
        w.dim, w.gen_hidden = self.dim, self.hidden
        w.gen_w0, w.gen_b0 = t(self.net[0].weight), t(self.net[0].bias)
        w.gen_w2, w.gen_b2 = t(self.net[2].weight), t(self.net[2].bias)

    @_inference_only
    @torch.no_grad()
    def forward(self, z: torch.Tensor) -> torch.Tensor:
        _require_cuda(z, "Generator input")
        with self._slot.session(z.device) as (ctx, h, slot, fresh):
            if fresh:
                self._sig = None
            sig = _sig(self.parameters())
            if sig != self._sig:
                w = _lib.hg_vae_weights()
                self._weights(w)
                ctx.check(_lib.lib().hg_load_vae(h, slot, C.byref(w)), "hg_load_vae")
                self._sig = sig
            zf = _f32(z)
            out = torch.empty_like(zf)
            ctx.check(_lib.lib().hg_generator(h, slot, zf.data_ptr(), zf.shape[0], out.data_ptr(),
                                                    _stream_ptr(z.device)), "hg_generator")
            return out


class VAE:
    """Fused Encoder -> reparameterise -> Generator (main_coop_vae.py:444-448) in one native call.

    ``eps`` is an input (the reference draws it with torch.randn at :446); pass ``eps=None`` to draw it
    with torch on the device.
    """

    def __init__(self, netE: Encoder, netG: Generator):
        self.netE, self.netG = netE, netG
        self._guard_modules = (netE, netG)
        self._slot = _Slot("vae")
        self._sig = None

    @_inference_only
    @torch.no_grad()
    def __call__(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None):
        _require_cuda(x, "VAE input")
        with self._slot.session(x.device) as (ctx, h, slot, fresh):
            if fresh:
                self._sig = None
            sig = _sig(list(self.netE.parameters()) + list(self.netG.parameters()))
            if sig != self._sig:
                w = _lib.hg_vae_weights()
                self.netE._weights(w)
                self.netG._weights(w)
                ctx.check(_lib.lib().hg_load_vae(h, slot, C.byref(w)), "hg_load_vae")
                self._sig = sig
            xf = _f32(x)
            ef = torch.randn_like(xf) if eps is None else _f32(eps)
            mean, logvar, z, bias = (torch.empty_like(xf) for _ in range(4))
            ctx.check(_lib.lib().hg_vae_forward(h, slot, xf.data_ptr(), ef.data_ptr(), xf.shape[0], mean.data_ptr(),
                                                      logvar.data_ptr(), z.data_ptr(), bias.data_ptr(),
                                                      _stream_ptr(x.device)), "hg_vae_forward")
            return mean, logvar, z, bias


class mlp_net(nn.Module):
    """finetune_ship.py:302-314."""

    def __init__(self, input_dim: int, output_dim: int, hidden_dim: int):
        super().__init__()
        self.dims = (input_dim, hidden_dim, output_dim)
        self.net = _Seq([(0, Linear(input_dim, hidden_dim)), (2, Linear(hidden_dim, hidden_dim)),
                         (4, Linear(hidden_dim, output_dim))])
        self._slot = _Slot("mlp")
        self._sig = None

    @_inference_only
    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _require_cuda(x, "mlp_net input")
        with self._slot.session(x.device) as (ctx, h, slot, fresh):
            if fresh:
                self._sig = None
            sig = _sig(self.parameters())
            if sig != self._sig:
                t = _lib.tensor
                w = _lib.hg_mlp_weights(self.dims[0], self.dims[1], self.dims[2], t(self.net[0].weight),
                                        t(self.net[0].bias), t(self.net[2].weight), t(self.net[2].bias),
                                        t(self.net[4].weight), t(self.net[4].bias))
                ctx.check(_lib.lib().hg_load_mlp(h, slot, C.byref(w)), "hg_load_mlp")
                self._sig = sig
            xf = _f32(x)
            out = torch.empty(xf.shape[0], self.dims[2], device=x.device, dtype=torch.float32)
            ctx.check(_lib.lib().hg_mlp_net(h, slot, xf.data_ptr(), xf.shape[0], out.data_ptr(), _stream_ptr(x.device)),
                            "hg_mlp_net")
            return out


# ---------------------------------------------------------------------------------------------
_util_ctx = _Ctx()


@torch.no_grad()
def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """``x / x.norm(dim=-1, keepdim=True)`` (main_coop_vae.py:438,466) on the device."""
    _require_cuda(x, "input")
    h = _util_ctx.get(x.device)
    xf = _f32(x)
    out = torch.empty_like(xf)
    _util_ctx.check(_lib.lib().hg_l2_normalize(h, xf.data_ptr(), xf.shape[0], xf.shape[1], out.data_ptr(),
                                               _stream_ptr(x.device)), "hg_l2_normalize")
    return out


@torch.no_grad()
def vae_loss(recon_x, x, mean, log_var, target=None) -> torch.Tensor:
    """Forward value of main_coop_vae.py:300-303 (``target`` is unused there as well)."""
    _require_cuda(x, "input")
    h = _util_ctx.get(x.device)
    a, b, m, lv = _f32(recon_x), _f32(x), _f32(mean), _f32(log_var)
    loss = torch.empty(1, device=x.device, dtype=torch.float32)
    _util_ctx.check(_lib.lib().hg_vae_loss(h, a.data_ptr(), b.data_ptr(), m.data_ptr(), lv.data_ptr(), b.shape[0],
                                           b.shape[1], loss.data_ptr(), _stream_ptr(x.device)), "hg_vae_loss")
    return loss[0]


class TextEncoder(nn.Module):
    """main_coop_vae.py:45-63 — shares the CLIP model's text tower (no copies)."""

    def __init__(self, clip_model: CLIP):
        super().__init__()
        object.__setattr__(self, "_clip", clip_model)
        self.transformer = clip_model.transformer
        self.positional_embedding = clip_model.positional_embedding
        self.ln_final = clip_model.ln_final
        self.text_projection = clip_model.text_projection
        self.dtype = clip_model.dtype

    @_inference_only
    def forward(self, prompts: torch.Tensor, tokenized_prompts: torch.Tensor) -> torch.Tensor:
        return self._clip.encode_text_embeds(prompts, tokenized_prompts)


class _PromptLearner(nn.Module):
    """PromptLearner_{hoi,h,o} (main_coop_vae.py:66-258): learnable context ``ctx [n_ctx,D]`` shifted by
    a per-sample ``bias`` between the SOT embedding and the class-name suffix.  The three reference
    classes differ only in ``n_ctx`` (5 / 4 / 4)."""

    N_CTX = 4

    def __init__(self, classnames: List[str], clip_model: CLIP):
        super().__init__()
        n_ctx = self.N_CTX
        self.dtype = clip_model.dtype
        ctx_dim = clip_model.ln_final.weight.shape[0]
        self.n_cls, self.n_ctx = len(classnames), n_ctx
        dev = clip_model.positional_embedding.device
        ctx_vectors = torch.empty(n_ctx, ctx_dim, dtype=self.dtype, device=dev)
        nn.init.normal_(ctx_vectors, std=0.02)
        self.prompt_prefix = " ".join(["X"] * n_ctx)
        self.ctx = nn.Parameter(ctx_vectors)
        self._names_key = None
        self.get_prefix_suffix_token(classnames, clip_model)

    def get_prefix_suffix_token(self, classnames: List[str], clip_model: CLIP):
        """main_coop_vae.py:103-117.  The reference re-tokenises every class name on the CPU each
        training / sampling step (:451, main_tip_finetune.py:783); the result depends only on the class
        names and the frozen token embedding, so it is cached per class-name list."""
        key = (tuple(classnames), clip_model.token_embedding.weight.data_ptr(),
               clip_model.token_embedding.weight._version)
        if key == self._names_key:
            return
        names = [n.replace("_", " ") for n in classnames]
        self.name_lens = [len(_clip._tokenizer.encode(n)) for n in names]
        prompts = [self.prompt_prefix + " " + n + "." for n in names]
        dev = clip_model.positional_embedding.device
        tokenized = torch.cat([_clip.tokenize(p) for p in prompts]).to(dev)
        with torch.no_grad():
            embedding = clip_model.token_embedding(tokenized).type(self.dtype)
        self.register_buffer("token_prefix", embedding[:, :1, :].contiguous())
        self.register_buffer("token_suffix", embedding[:, 1 + self.n_ctx:, :].contiguous())
        self.tokenized_prompts = tokenized
        self._names_key = key
        self._f32_key = None              # fp32 operands of hg_assemble_prompts are rebuilt on the next forward

    @_inference_only
    @torch.no_grad()
    def forward(self, bias: torch.Tensor, target: torch.Tensor, out: Optional[torch.Tensor] = None,
                tokens: Optional[int] = None) -> torch.Tensor:
        """main_coop_vae.py:119-128 -> prompts [R, L, D].

        Extensions for the device-resident sampling loop (hoigen_amd.generation): ``tokens`` = only the first ``tokens`` positions of
        every prompt (what a text tower truncated to max(EOT) + 1 reads: 13-16 of 77), ``out`` = a caller-owned fp32 ``[R, L, D]``
        buffer, e.g. a slice of the batch the tower is called with (no concatenation afterwards)."""
        _require_cuda(bias, "bias")
        h = _util_ctx.get(bias.device)
        bf = _f32(bias)
        R, D = bf.shape
        C_, Ls = self.token_suffix.shape[0], self.token_suffix.shape[1]
        L_full = 1 + self.n_ctx + Ls
        L = L_full if tokens is None else int(tokens)
        if not (1 + self.n_ctx < L <= L_full):
            raise RuntimeError(f"hoigen_amd: tokens must lie in ({1 + self.n_ctx}, {L_full}] (got {L})")
        tgt = target.to(device=bias.device, dtype=torch.int32).contiguous()
        if out is None:
            out = torch.empty(R, L, D, device=bias.device, dtype=torch.float32)
        elif (out.shape != (R, L, D) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != bias.device):
            raise RuntimeError(f"hoigen_amd: out must be a contiguous fp32 [{R}, {L}, {D}] tensor on {bias.device}")
        # fp32 views of the (usually fp16: clip_model.dtype) buffers.  They are kept on the object: a temporary made
        # inline (`_f32(t).data_ptr()`) is freed before the kernel is enqueued and its block can be handed to the
        # next temporary.  Rebuilt when the buffers or ctx change (get_prefix_suffix_token, load_state_dict, .to()).
        key = _sig([self.token_prefix, self.token_suffix, self.ctx]) + (str(bias.device), L)
        if key != getattr(self, "_f32_key", None):
            suf = self.token_suffix if L == L_full else self.token_suffix[:, :L - 1 - self.n_ctx, :]
            self._f32_ops = tuple(_f32(t).to(bias.device).contiguous() for t in (self.token_prefix, suf, self.ctx))
            self._f32_key = key
        pre, suf, ctx = self._f32_ops
        with torch.cuda.device(bias.device):
            _util_ctx.check(_lib.lib().hg_assemble_prompts(
                h, pre.data_ptr(), suf.data_ptr(), ctx.data_ptr(), bf.data_ptr(), tgt.data_ptr(), R, C_, L, self.n_ctx,
                D, out.data_ptr(), _stream_ptr(bias.device)), "hg_assemble_prompts")
        return out


class PromptLearner_hoi(_PromptLearner):
    N_CTX = 5      # main_coop_vae.py:70


class PromptLearner_h(_PromptLearner):
    N_CTX = 4      # main_coop_vae.py:135


class PromptLearner_o(_PromptLearner):
    N_CTX = 4      # main_coop_vae.py:200
