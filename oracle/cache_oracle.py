"""TEST INFRASTRUCTURE ONLY - CPU restatement of the cache-model (Tip-adapter) logits that consume the
embeddings of the hot path.  Nothing under hoigen_amd/ imports this file.

Reference expressions (upt_tip_cache_model_free_finetune_distill3.py:1158-1170):
    phi_union_HO    = cat([human, object], -1) @ adapter_HO_weight.T + adapter_HO_bias        # [R, S]
    logits_cache_HO = ((phi_union_HO @ label_HO) / sample_lens_HO) / 2                         # [R, C]
    phi_union_U     = union @ adapter_U_weight.T + adapter_U_bias
    logits_cache_U  = (phi_union_U @ label_U) / sample_lens_U
    logits_text     = union @ adapter_union_weight.T
with adapter_*_weight = cached, L2-normalised embeddings [S, K] (:498,506,631-633), adapter_*_bias = -1 [S]
(:499,507), label_* = multi-hot [S, C] (:500,508), sample_lens_* [C].

Pinned: tests/golden/g7_cache.npz holds the outputs of those very source lines executed here on seeded inputs
(tests/golden/make_golden_cache.py); tests/test_cache_model.py compares this restatement with them.
"""
import numpy as np


def cache_logits(features, weight, bias, labels, sample_lens, post_div=1.0):
    """((features @ weight.T + bias) @ labels) / sample_lens / post_div, fp32."""
    f = np.asarray(features, np.float32)
    phi = f @ np.asarray(weight, np.float32).T + np.asarray(bias, np.float32)
    out = (phi @ np.asarray(labels, np.float32)) / np.asarray(sample_lens, np.float32)
    return out / np.float32(post_div) if post_div != 1.0 else out


def linear_logits(features, weight):
    """features @ weight.T (logits_text)."""
    return np.asarray(features, np.float32) @ np.asarray(weight, np.float32).T
