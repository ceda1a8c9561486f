"""``load`` / ``tokenize`` / ``available_models`` with the reference's signatures
(/root/reference/clipnet/clip.py:29-228; the int32 ``tokenize`` twin of
/root/reference/CLIP_models_adapter_prior2.py:988-1034 is ``tokenize(..., dtype=torch.int32)``).
"""
from __future__ import annotations

import hashlib
import os
import urllib.request
import warnings
from typing import List, Union

import numpy as np
import torch

from .model import build_model
from .simple_tokenizer import SimpleTokenizer as _Tokenizer

__all__ = ["available_models", "load", "tokenize"]
_tokenizer = _Tokenizer()

_MODELS = {
    "ViT-B/32": "https://openaipublic.azureedge.net/clip/models/40d365715913c9da98579312b702a82c18be219cc2a73407c4526f58eba950af/ViT-B-32.pt",
    "ViT-B/16": "https://openaipublic.azureedge.net/clip/models/5806e77cd80f8b59890b7e101eabd078d9fb84e6937f9e85e4ecb61988df416f/ViT-B-16.pt",
}

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _download(url: str, root: str) -> str:
    """Checksum-verified fetch into ``root`` (clipnet/clip.py:39-72).  The SHA-256 is the URL's
    second-to-last path element."""
    os.makedirs(root, exist_ok=True)
    target = os.path.join(root, os.path.basename(url))
    expected = url.split("/")[-2]
    if os.path.exists(target) and not os.path.isfile(target):
        raise RuntimeError(f"{target} exists and is not a regular file")
    if os.path.isfile(target):
        if hashlib.sha256(open(target, "rb").read()).hexdigest() == expected:
            return target
        warnings.warn(f"{target} exists, but the SHA256 checksum does not match; re-downloading the file")
    with urllib.request.urlopen(url) as src, open(target, "wb") as dst:
        while True:
            buf = src.read(1 << 16)
            if not buf:
                break
            dst.write(buf)
    if hashlib.sha256(open(target, "rb").read()).hexdigest() != expected:
        raise RuntimeError("Model has been downloaded but the SHA256 checksum does not not match")
    return target


class _Transform:
    """Resize(n_px, bicubic) -> CenterCrop(n_px) -> RGB -> ToTensor -> Normalize(CLIP mean/std)
    (clipnet/clip.py:75-82), implemented on PIL + torch (torchvision is not a dependency)."""

    def __init__(self, n_px: int):
        self.n_px = n_px

    def __call__(self, image):
        from PIL import Image

        n = self.n_px
        w, h = image.size
        if w <= h:
            nw, nh = n, int(n * h / w)
        else:
            nw, nh = int(n * w / h), n
        image = image.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - n) / 2.0)), int(round((nh - n) / 2.0))
        image = image.crop((left, top, left + n, top + n)).convert("RGB")
        x = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
        mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
        std = torch.tensor(CLIP_STD).view(3, 1, 1)
        return (x - mean) / std


def _transform(n_px: int):
    return _Transform(n_px)


def available_models() -> List[str]:
    return list(_MODELS.keys())


def load(name: str, device: Union[str, torch.device] = "cuda" if torch.cuda.is_available() else "cpu",
         jit: bool = False, download_root: str = None):
    """Load a CLIP model (clipnet/clip.py:90-137).  ``name`` is a model name or a checkpoint path
    (TorchScript archive or plain state dict).  Returns ``(model, preprocess)``.

    The returned model computes only on a HIP device; with ``device="cpu"`` the weights are loaded
    (fp32, as the reference does) but ``encode_*`` raise until the model is moved to the GPU.
    """
    if name in _MODELS:
        model_path = _download(_MODELS[name], download_root or os.path.expanduser("~/.cache/clip"))
    elif os.path.isfile(name):
        model_path = name
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    if jit:
        warnings.warn("hoigen_amd runs its own HIP kernels; jit=True is ignored (non-JIT model returned)")
    try:
        archive = torch.jit.load(model_path, map_location="cpu").eval()
        state_dict = archive.state_dict()
    except RuntimeError:
        state_dict = torch.load(model_path, map_location="cpu")
    model = build_model(state_dict).to(device)
    if str(device) == "cpu":
        model.float()
    return model, _transform(model.visual.input_resolution)


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False,
             dtype: torch.dtype = torch.long) -> torch.Tensor:
    """clipnet/clip.py:192-228: ``[SOT] + bpe(text) + [EOT]``, zero padded to ``context_length``;
    RuntimeError when too long unless ``truncate`` (then cut and force the last token to EOT)."""
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = _tokenizer.encoder["<|startoftext|>"], _tokenizer.encoder["<|endoftext|>"]
    result = torch.zeros(len(texts), context_length, dtype=dtype)
    for i, text in enumerate(texts):
        tokens = [sot] + _tokenizer.encode(text) + [eot]
        if len(tokens) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {texts[i]} is too long for context length {context_length}")
            tokens = tokens[:context_length]
            tokens[-1] = eot
        result[i, :len(tokens)] = torch.tensor(tokens, dtype=dtype)
    return result
