"""ctypes binding of ``libhoigen_amd.so`` (the C ABI declared in ``include/hoigen_amd.h``).

This is the only place the native library is touched.  There is **no CPU fallback**: if the library
is missing, or no HIP device is present when a compute entry point is called, a ``RuntimeError`` is
raised.  The structures below mirror the header field by field.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading
from typing import Dict, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# HG_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants inside one GPU session)
LIB_PATH = os.environ.get("HG_LIB_PATH") or os.path.join(HERE, "csrc", "libhoigen_amd.so")
HG_MAX_SLOTS = 16
HG_F32, HG_F16 = 0, 1


class hg_tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32)]


_BLOCK_FIELDS = ["in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias", "ln_1_weight",
                 "ln_1_bias", "c_fc_weight", "c_fc_bias", "c_proj_weight", "c_proj_bias", "ln_2_weight",
                 "ln_2_bias"]


class hg_block_weights(C.Structure):
    _fields_ = [(n, hg_tensor) for n in _BLOCK_FIELDS]


_DEC_FIELDS = ["attn_in_proj_weight", "attn_in_proj_bias", "attn_out_proj_weight", "attn_out_proj_bias",
               "linear1_weight", "linear1_bias", "linear2_weight", "linear2_bias", "norm2_weight",
               "norm2_bias", "norm3_weight", "norm3_bias"]


class hg_decoder_layer_weights(C.Structure):
    _fields_ = [(n, hg_tensor) for n in _DEC_FIELDS]


class hg_adapter_weights(C.Structure):
    _fields_ = [("present", C.c_int32), ("bottleneck", C.c_int32), ("scale", hg_tensor),
                ("down_proj_weight", hg_tensor), ("down_proj_bias", hg_tensor),
                ("up_proj_weight", hg_tensor), ("up_proj_bias", hg_tensor),
                ("prior_layer", hg_decoder_layer_weights), ("self_layer", hg_decoder_layer_weights),
                ("n_extra_prior_layers", C.c_int32), ("extra_prior_layers", C.POINTER(hg_decoder_layer_weights))]


class hg_vit_weights(C.Structure):
    _fields_ = [("width", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("patch_size", C.c_int32),
                ("input_resolution", C.c_int32), ("output_dim", C.c_int32),
                ("conv1_weight", hg_tensor), ("class_embedding", hg_tensor), ("positional_embedding", hg_tensor),
                ("ln_pre_weight", hg_tensor), ("ln_pre_bias", hg_tensor), ("ln_post_weight", hg_tensor),
                ("ln_post_bias", hg_tensor), ("proj", hg_tensor),
                ("blocks", C.POINTER(hg_block_weights)), ("adapters", C.POINTER(hg_adapter_weights))]


class hg_text_weights(C.Structure):
    _fields_ = [("width", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("context_length", C.c_int32), ("vocab_size", C.c_int32), ("output_dim", C.c_int32),
                ("token_embedding", hg_tensor), ("positional_embedding", hg_tensor),
                ("ln_final_weight", hg_tensor), ("ln_final_bias", hg_tensor), ("text_projection", hg_tensor),
                ("blocks", C.POINTER(hg_block_weights))]


class hg_vae_weights(C.Structure):
    _fields_ = [("dim", C.c_int32), ("enc_hidden", C.c_int32), ("gen_hidden", C.c_int32),
                ("enc_w0", hg_tensor), ("enc_b0", hg_tensor), ("enc_mean_w", hg_tensor), ("enc_mean_b", hg_tensor),
                ("enc_logvar_w", hg_tensor), ("enc_logvar_b", hg_tensor), ("gen_w0", hg_tensor),
                ("gen_b0", hg_tensor), ("gen_w2", hg_tensor), ("gen_b2", hg_tensor)]


class hg_mlp_weights(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden_dim", C.c_int32), ("out_dim", C.c_int32),
                ("w0", hg_tensor), ("b0", hg_tensor), ("w2", hg_tensor), ("b2", hg_tensor),
                ("w4", hg_tensor), ("b4", hg_tensor)]


class hg_cache_weights(C.Structure):
    _fields_ = [("weight", hg_tensor), ("bias", hg_tensor), ("labels", hg_tensor), ("sample_lens", hg_tensor),
                ("S", C.c_int32), ("K", C.c_int32), ("C", C.c_int32), ("post_div", C.c_float)]


class hg_prof_rec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ms", C.c_float)]


HG_PROF_OFF, HG_PROF_ALL, HG_PROF_ATTENTION, HG_PROF_QKV_ATTN = -1, -2, 100, 101

_P = C.c_void_p
_I = C.c_int
# name -> (restype, argtypes); must list every symbol of include/hoigen_amd.h (tests check this)
SIGNATURES = {
    "hg_create": (_P, [_I]),
    "hg_destroy": (None, [_P]),
    "hg_last_error": (C.c_char_p, [_P]),
    "hg_version": (C.c_char_p, []),
    "hg_load_vit": (_I, [_P, C.POINTER(hg_vit_weights)]),
    "hg_load_text": (_I, [_P, C.POINTER(hg_text_weights)]),
    "hg_load_vae": (_I, [_P, _I, C.POINTER(hg_vae_weights)]),
    "hg_load_mlp": (_I, [_P, _I, C.POINTER(hg_mlp_weights)]),
    "hg_update_adapters": (_I, [_P, C.POINTER(hg_adapter_weights), _I]),
    "hg_encode_image": (_I, [_P, _P, _I, _P, _P]),
    "hg_encode_image_prior": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "hg_encode_image_trace": (_I, [_P, _P, _I, _P, _P, _P]),
    "hg_encode_text_ids": (_I, [_P, _P, _I, _I, _P, _I, _P]),
    "hg_encode_text_embeds": (_I, [_P, _P, _P, _I, _I, _P, _I, _P]),
    "hg_token_embedding": (_I, [_P, _P, _I, _P, _P]),
    "hg_vae_forward": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    "hg_generator": (_I, [_P, _I, _P, _I, _P, _P]),
    "hg_mlp_net": (_I, [_P, _I, _P, _I, _P, _P]),
    "hg_assemble_prompts": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "hg_l2_normalize": (_I, [_P, _P, _I, _I, _P, _P]),
    "hg_vae_loss": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "hg_roi_align": (_I, [_P, _P, _I, _I, _I, _P, _I, C.c_float, _I, _P, _P, _P]),
    "hg_load_cache": (_I, [_P, _I, _P]),
    "hg_cache_logits": (_I, [_P, _I, _P, _I, _P, _P]),
    "hg_preprocess_crops": (_I, [_P, _P, _I, _I, _P, _I, _I, _I, C.c_uint32, _P, _P, _P]),
    "hg_workspace_bytes": (_I, [_P, C.POINTER(C.c_uint64)]),
    "hg_set_option": (_I, [_P, C.c_char_p, _I]),
    "hg_get_option": (_I, [_P, C.c_char_p, C.POINTER(C.c_int32)]),
    "hg_test_gemm": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "hg_test_gemm_ln": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hg_test_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "hg_test_qkv_attn": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "hg_test_gemm_hilo": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "hg_profile_begin": (_I, [_P, _I, _I]),
    "hg_profile_end": (_I, [_P, _P, _I, C.POINTER(C.c_int32)]),
}

_lib = None
_lock = threading.RLock()   # re-entrant: ctx() calls lib() under the lock
_ctxs: Dict[int, int] = {}


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    script = os.path.join(HERE, "csrc", "build.sh")
    if force:
        for f in os.listdir(os.path.join(HERE, "csrc")):
            if f.endswith(".o"):
                os.remove(os.path.join(HERE, "csrc", f))
    subprocess.run(["bash", script], check=True)
    return LIB_PATH


def lib():
    """The loaded shared library (raises RuntimeError if it has not been built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"hoigen_amd: native library {LIB_PATH} is missing. Build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                        "There is no CPU fallback.")
                try:
                    l = C.CDLL(LIB_PATH)
                except OSError as e:  # pragma: no cover
                    raise RuntimeError(f"hoigen_amd: cannot load {LIB_PATH}: {e}") from e
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = l
    return _lib


def ctx(device: int) -> int:
    """The per-device context handle (created on first use)."""
    h = _ctxs.get(device)
    if h is None:
        with _lock:
            h = _ctxs.get(device)
            if h is None:
                h = lib().hg_create(device)
                if not h:
                    raise RuntimeError(
                        f"hoigen_amd: hg_create({device}) failed - no HIP device. The hot path runs only on "
                        "an AMD GPU (gfx950); there is no CPU fallback.")
                _ctxs[device] = h
    return h


def check(device: int, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().hg_last_error(ctx(device))
        raise RuntimeError(f"hoigen_amd: {what} failed ({rc}): {msg.decode() if msg else '?'}")


def profile(handle: int, kind: int, max_launches: int, fn):
    """Run ``fn()`` with hipEvent pairs around the launches of ``kind`` (HG_PROF_ALL: every GEMM / attention /
    fused-VAE launch) of the context ``handle``; returns ``(fn's result, [(kind, M, N, K, ms), ...])``."""
    l = lib()
    rc = l.hg_profile_begin(handle, kind, max_launches)
    if rc:
        raise RuntimeError(f"hoigen_amd: hg_profile_begin failed ({rc})")
    try:
        out = fn()
    finally:
        recs = (hg_prof_rec * max(1, max_launches))()
        n = C.c_int32()
        rc = l.hg_profile_end(handle, recs, max_launches, C.byref(n))
    if rc:
        raise RuntimeError(f"hoigen_amd: hg_profile_end failed ({rc})")
    return out, [(r.kind, r.M, r.N, r.K, float(r.ms)) for r in recs[: n.value]]


def tensor(t: Optional["object"]) -> hg_tensor:
    """hg_tensor view of a contiguous CUDA(HIP) torch tensor (fp32 or fp16)."""
    import torch

    if t is None:
        return hg_tensor(None, 0)
    if not t.is_cuda:
        raise RuntimeError("hoigen_amd: weights must live on a HIP device (call .cuda() on the module)")
    if t.dtype == torch.float32:
        dt = HG_F32
    elif t.dtype == torch.float16:
        dt = HG_F16
    else:
        raise RuntimeError(f"hoigen_amd: unsupported weight dtype {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("hoigen_amd: weight tensors must be contiguous")
    return hg_tensor(t.data_ptr(), dt)
