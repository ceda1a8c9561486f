"""The algebra behind the folded instance adapter (hg_elem.hip adapter_q_kernel, hg_adapter.hip fold epilogue),
as an executable statement in float64 numpy: no GPU, no library.

Reference: CLIP_models_adapter_prior2.py:184-203 (down_proj -> decoder layer -> up_proj, `up * self.scale`) and :456
(`x = x + adapt_x`), then ln_1 (:458)."""
import numpy as np

D, d = 768, 64


def _setup(seed):
    r = np.random.default_rng(seed)
    scale = r.normal(0.02, 0.1, D)
    w_up = r.normal(0, d ** -0.5, (D, d))
    b_up = r.normal(0, 0.02, D)
    g3, b3 = r.normal(1, 0.2, d), r.normal(0, 0.1, d)
    t = r.normal(0, 1.5, (5, d)) + r.normal(0, 1, (5, 1))          # what norm3 sees, five tokens
    z = (t - t.mean(1, keepdims=True)) / np.sqrt(t.var(1, keepdims=True) + 1e-5)
    x = r.normal(0, 2, (5, D)) + r.normal(0, 1, (5, 1))
    x[:, 7] *= 40                                                   # an outlier channel
    return scale, w_up, b_up, g3, b3, z, x


def _q(scale, w_up, b_up, g3, b3):
    p = scale[:, None] * w_up                                       # P = diag(scale) W_up
    q = np.empty((D, d))
    q[:, :63] = g3[None, :63] * p[:, :63] - g3[63] * p[:, 63:64]
    q[:, 63] = p @ b3 + scale * b_up
    return q


def test_update_is_q_times_e_without_a_bias():
    scale, w_up, b_up, g3, b3, z, x = _setup(0)
    dd = g3 * z + b3                                                # the decoder layer's output (norm3 with its affine part)
    a = scale * (dd @ w_up.T + b_up)                                # up_proj, `* self.scale`
    e = z.copy()
    assert np.allclose(z.sum(1), 0, atol=1e-12)
    e[:, 63] = 1.0
    assert np.allclose(e @ _q(scale, w_up, b_up, g3, b3).T, a, rtol=0, atol=1e-12)


def test_layernorm_statistics_of_the_updated_row_from_64_wide_products():
    scale, w_up, b_up, g3, b3, z, x = _setup(1)
    q = _q(scale, w_up, b_up, g3, b3)
    e = z.copy()
    e[:, 63] = 1.0
    y = x + e @ q.T
    c = x.mean(1) + 0.03                                            # centre of the fp16 copy: near the mean, not equal
    mean_x, var_x = x.mean(1), x.var(1)
    sa = e @ q.sum(0)                                               # qm
    cross = np.einsum("ti,ti->t", e, (x - c[:, None]) @ q)          # e . w',  w' = (x - c) Q
    quad = np.einsum("ti,ij,tj->t", e, q.T @ q, e)                  # e^T G e
    mean_y = mean_x + sa / D
    var_y = var_x + (2 * (cross - (mean_x - c) * sa) + quad - sa * sa / D) / D
    assert np.allclose(mean_y, y.mean(1), rtol=0, atol=1e-12)
    assert np.allclose(var_y, y.var(1), rtol=1e-11, atol=0)


def test_qkv_takes_the_update_as_64_more_k_columns():
    scale, w_up, b_up, g3, b3, z, x = _setup(2)
    r = np.random.default_rng(3)
    q = _q(scale, w_up, b_up, g3, b3)
    e = z.copy()
    e[:, 63] = 1.0
    gamma, beta = r.normal(1, 0.1, D), r.normal(0, 0.1, D)
    w, b = r.normal(0, D ** -0.5, (96, D)), r.normal(0, 0.1, 96)
    y = x + e @ q.T
    want = ((y - y.mean(1, keepdims=True)) / np.sqrt(y.var(1, keepdims=True) + 1e-5) * gamma + beta) @ w.T + b
    # folded operands: W' = W gamma, cs = row sums of W', b' = b + W beta; x16 = x - c
    wf = w * gamma[None, :]
    cs, bf = wf.sum(1), b + w @ beta
    c = x.mean(1) - 0.02
    acc = np.concatenate([x - c[:, None], e], 1) @ np.concatenate([wf, wf @ q], 1).T
    rstd = 1 / np.sqrt(y.var(1) + 1e-5)
    got = rstd[:, None] * (acc - (y.mean(1) - c)[:, None] * cs[None, :]) + bf
    assert np.allclose(got, want, rtol=0, atol=1e-10)
