"""Cache-model (Tip-adapter) logits on the embeddings produced by the hot path.

Counterpart of the expressions in upt_tip_cache_model_free_finetune_distill3.py:1158-1170
    phi    = features @ weight.T + bias                      # affinities to the S cached samples
    logits = (phi @ labels) / sample_lens [/ 2]              # per class
    logits_text = features @ text_weight.T
as two MFMA GEMMs behind ``hg_load_cache`` / ``hg_cache_logits``.  Inference only; CPU tensors raise
``RuntimeError`` (there is no CPU fallback).
"""
import threading
from typing import Optional

import torch

from . import _lib

HG_MAX_CACHE_SLOTS = 8                      # include/hoigen_amd.h
_free = list(range(HG_MAX_CACHE_SLOTS))     # slots of the shared per-device context not owned by a live object
_lock = threading.Lock()


def _dev_index(t: torch.Tensor) -> int:
    if t.device.type != "cuda":
        raise RuntimeError("hoigen_amd: the cache model runs only on a HIP device (there is no CPU fallback)")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


class CacheLogits:
    """``CacheLogits(weight [S,K], bias [S] | None, labels [S,C] | None, sample_lens [C] | None, post_div)``.

    ``__call__(features [R,K]) -> [R,C]`` = ``((features @ weight.T + bias) @ labels) / sample_lens / post_div``
    (``post_div=2`` for the human-object branch, upt…:1167); without ``labels`` it is the linear map
    ``features @ weight.T + bias -> [R,S]`` (``logits_text``, upt…:1170).  The tensors are copied into the native
    context at construction (call ``update`` after changing them)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                 sample_lens: Optional[torch.Tensor] = None, post_div: float = 1.0):
        with _lock:
            if not _free:
                raise RuntimeError(f"hoigen_amd: all {HG_MAX_CACHE_SLOTS} cache-model slots are in use; close() or "
                                   "delete a CacheLogits first")
            self.slot = _free.pop(0)
        self.post_div = float(post_div)
        try:
            self.update(weight, bias, labels, sample_lens)
        except Exception:
            self.close()
            raise

    def close(self):
        """Give the slot back (the object is unusable afterwards)."""
        with _lock:
            if getattr(self, "slot", None) is not None:
                _free.append(self.slot)
                _free.sort()
                self.slot = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover - interpreter shutdown
            pass

    def update(self, weight, bias=None, labels=None, sample_lens=None):
        if self.slot is None:
            raise RuntimeError("hoigen_amd: this CacheLogits has been closed")
        self.device = _dev_index(weight)
        if (labels is None) != (sample_lens is None):
            raise ValueError("labels and sample_lens go together")
        S, K = weight.shape
        self.S, self.K = int(S), int(K)
        self.C = int(labels.shape[1]) if labels is not None else 0
        keep = [t.detach().float().contiguous() if t is not None else None for t in (weight, bias, labels, sample_lens)]
        w = _lib.hg_cache_weights()
        w.weight, w.bias, w.labels, w.sample_lens = (_lib.tensor(t) for t in keep)
        w.S, w.K, w.C, w.post_div = self.S, self.K, self.C, self.post_div
        with torch.cuda.device(self.device):
            rc = _lib.lib().hg_load_cache(_lib.ctx(self.device), self.slot, _lib.C.byref(w))
        _lib.check(self.device, rc, "hg_load_cache")

    def __call__(self, features: torch.Tensor) -> torch.Tensor:
        if self.slot is None:
            raise RuntimeError("hoigen_amd: this CacheLogits has been closed")
        if _dev_index(features) != self.device:
            raise RuntimeError("features are on a different device than the cache model")
        if features.dim() != 2 or features.shape[1] != self.K:
            raise ValueError(f"features must be [R, {self.K}]")
        f = features.detach().float().contiguous()
        n_out = self.C if self.C else self.S
        out = torch.empty(f.shape[0], n_out, dtype=torch.float32, device=f.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().hg_cache_logits(_lib.ctx(self.device), self.slot, f.data_ptr(), f.shape[0], out.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream)
        _lib.check(self.device, rc, "hg_cache_logits")
        return out


@torch.no_grad()
def build_clip_cache_model(features: torch.Tensor, verbs, num_classes: int, num_shot: int):
    """Cache keys / values with per-class shot selection: the part of ``utils.build_clip_cache_model``
    (/root/reference/utils.py:31-61) that follows the encoder.

    ``features [n, D]``: the L2-NORMALISED global embeddings of the n training crops in loader order (the reference
    normalises ``feat_global`` row by row, :24-27, before this point); ``verbs``: for each crop the class indices
    present in its target (``target['verb']`` / ``['actions']``, :33-38).  Returns ``(cache_keys [D, S],
    cache_values [S, num_classes])`` with ``S = num_classes * num_shot`` for fully populated classes.

    Same arithmetic and the same draws from torch's GLOBAL generator in the same order as the reference (one
    ``torch.randperm(len)`` per class in class order, ``torch.randn(D)`` per missing shot of an empty class), so with
    ``torch.manual_seed(s)`` set by the caller the result equals the reference's for the same inputs.  Pure index /
    copy work on the host side of the device tensors; no kernels involved.
    """
    n, D = features.shape
    dev = features.device
    cache_keys = [[] for _ in range(num_classes)]
    cache_values = [[] for _ in range(num_classes)]
    for i in range(n):
        values = torch.zeros(num_classes)
        for j in verbs[i]:
            values[int(j)] = 1
        for k in torch.nonzero(values):                      # ascending class order, as torch.nonzero gives it
            cache_values[k.item()].append(values)
            cache_keys[k.item()].append(features[i, :])
    new_keys, new_values = [], []
    for c in range(num_classes):
        ks, vs = [], []
        topk_idx = torch.randperm(len(cache_values[c]))[:num_shot]
        for idx in topk_idx:
            vs.append(cache_values[c][idx])
            ks.append(cache_keys[c][idx])
        if not vs:                                           # class without samples: random keys, one-hot values (:52-57)
            for _ in range(num_shot):
                ks.append(torch.randn(D).to(dev))
                v = torch.zeros(num_classes)
                v[c] = 1
                vs.append(v)
        new_keys.append(torch.stack(ks))
        new_values.append(torch.stack(vs))
    keys = torch.cat(new_keys)
    values = torch.cat(new_values)
    keys = keys / keys.norm(dim=-1, keepdim=True)
    return keys.permute(1, 0), values
