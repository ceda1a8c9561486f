"""The CPU oracle (oracle/*.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerance: fp32 restatement of fp32 reference ->
max |d| <= 2e-5 * max|ref| (reduction-order noise only); integer artefacts exact."""
import json
import os

import numpy as np
import pytest
import torch

from hoigen_amd import synth
from oracle import clip_oracle as co
from oracle import vae_oracle as vo

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def close(a, b, rel=2e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b).max()
    assert err <= rel * max(np.abs(b).max(), 1e-6), (err, np.abs(b).max())


@pytest.fixture(scope="module")
def g1():
    return dict(np.load(f"{G}/g1_tiny.npz"))


@pytest.fixture(scope="module")
def tiny_sd():
    return co.reference_weight_rounding(synth.clip_state_dict(synth.TINY, 10))


def test_tiny_image_intermediates(g1, tiny_sd):
    img = torch.from_numpy(synth.crops(3, 32, seed=11))
    col = []
    out = co.encode_image(tiny_sd, img, collect=col)
    close(col[0], g1["img_ln_pre"])
    for i in range(2):
        close(col[1 + i], g1["img_blocks"][i])
    close(out, g1["img_out"])
    # patch embedding as a GEMM == conv2d (clipnet/model.py:220)
    w = tiny_sd["visual.conv1.weight"]
    pe = co.patchify(img, 16) @ w.reshape(w.shape[0], -1).T
    close(pe.reshape(3, 2, 2, -1).permute(0, 3, 1, 2), g1["img_conv"])
    # block-0 sub-ops
    pre = "visual.transformer.resblocks.0."
    x0 = col[0]
    h = co.layer_norm(x0, tiny_sd[pre + "ln_1.weight"], tiny_sd[pre + "ln_1.bias"])
    close(h, g1["img_b0_ln1"])
    att = co.attention(h, tiny_sd, pre, 2, False)
    close(att, g1["img_b0_attn"])
    x1 = x0 + att
    h2 = co.layer_norm(x1, tiny_sd[pre + "ln_2.weight"], tiny_sd[pre + "ln_2.bias"])
    close(h2, g1["img_b0_ln2"])
    close(co.quick_gelu(co.linear(h2, tiny_sd[pre + "mlp.c_fc.weight"], tiny_sd[pre + "mlp.c_fc.bias"])),
          g1["img_b0_fc_gelu"])


def test_patch_index_map_exact():
    # token 1+g*r+c <-> pixel block (16r..16r+15, 16c..16c+15); column = c*256 + ky*16 + kx
    img = torch.arange(2 * 3 * 32 * 32, dtype=torch.float32).reshape(2, 3, 32, 32)
    a = co.patchify(img, 16)
    for b, r, c, ch, ky, kx in [(0, 0, 0, 0, 0, 0), (1, 1, 0, 2, 5, 7), (0, 1, 1, 1, 15, 15)]:
        assert a[b, 2 * r + c, ch * 256 + ky * 16 + kx] == img[b, ch, 16 * r + ky, 16 * c + kx]


def test_tiny_text(g1, tiny_sd):
    toks = torch.from_numpy(g1["txt_tokens"])
    assert np.array_equal(toks.numpy(), synth.tiny_tokens(6, synth.TINY, 12))
    col = []
    out = co.encode_text(tiny_sd, toks, collect=col)
    for i in range(2):
        close(col[1 + i], g1["txt_blocks"][i])
    close(out, g1["txt_out"])
    # causal truncation: same selected outputs when run on max(EOT)+1 positions only
    Lp = int(co.eot_index(toks).max()) + 1
    close(co.encode_text(tiny_sd, toks[:, :Lp]), g1["txt_out"], rel=5e-5)


def test_tiny_variant_c(g1):
    sd = co.as_tensors(synth.clip_state_dict(synth.TINY, 10))      # variant C: no fp16 rounding
    sd.update(co.as_tensors(synth.adapter_state_dict(synth.TINY, 13)))
    img = torch.from_numpy(synth.crops(3, 32, seed=11))
    pri, mask = synth.priors(3, n=6, dim=64, n_pad=2, seed=14)
    prior = (torch.from_numpy(pri), torch.from_numpy(mask))
    g, l = co.visual_with_prior(sd, img, prior, adapter_layers=range(2))
    close(g, g1["c_prior_global"]); close(l, g1["c_prior_local"])
    g, l = co.visual_with_prior(sd, img, None, adapter_layers=range(2))
    close(g, g1["c_noprior_global"]); close(l, g1["c_noprior_local"])
    xin = torch.from_numpy(g1["adapter_in"])
    pre = "visual.transformer.resblocks.0.adaptermlp."
    close(co.adapter(xin, sd, pre, prior), g1["adapter_prior"])
    close(co.adapter(xin, sd, pre, None), g1["adapter_noprior"])
    # untrained adapters are an exact no-op
    sd0 = co.as_tensors(synth.clip_state_dict(synth.TINY, 10))
    sd0.update(co.as_tensors(synth.adapter_state_dict(synth.TINY, 13, trained=False)))
    g, l = co.visual_with_prior(sd0, img, prior, adapter_layers=range(2))
    close(g, g1["c_untrained_global"]); close(l, g1["c_untrained_local"])
    close(co.encode_text(sd, torch.from_numpy(g1["txt_tokens"])), g1["c_txt_out"])


def test_tiny_vae_chain(g1, tiny_sd):
    D = 128
    feats = torch.from_numpy(g1["vae_feats"])
    img = torch.from_numpy(synth.crops(5, 32, seed=16))
    close(co.l2_normalize(co.encode_image(tiny_sd, img)), g1["vae_feats"], rel=5e-5)
    se = co.as_tensors(synth.encoder_state_dict(17, dim=D, hidden=256, wstd=0.05))
    sg = co.as_tensors(synth.generator_state_dict(18, dim=D, hidden=384, wstd=0.05))
    eps = torch.from_numpy(synth.hg_normal((5, D), 19))
    mean, lv, z, bias = vo.vae_forward(se, sg, feats, eps)
    close(mean, g1["vae_mean"]); close(lv, g1["vae_log_var"]); close(z, g1["vae_z"]); close(bias, g1["vae_bias"])
    cls_tok = torch.from_numpy(g1["vae_cls_tokens"])
    emb = tiny_sd["token_embedding.weight"][cls_tok]
    ctx = torch.from_numpy(synth.hg_normal((3, D), 21, 0.02))
    target = torch.from_numpy(g1["vae_target"])
    prompts = vo.assemble_prompts(emb[:, :1], emb[:, 4:], ctx, bias, target)
    close(prompts, g1["vae_prompts"])
    tf = co.text_encoder_embeds(tiny_sd, prompts, cls_tok[target])
    close(tf, g1["vae_text_features"], rel=5e-5)
    loss = vo.vae_loss(co.l2_normalize(tf), feats, mean, lv)
    close(loss, g1["vae_loss"], rel=5e-5)


def test_g4_vae():
    g = dict(np.load(f"{G}/g4_vae.npz"))
    se, sg = co.as_tensors(synth.encoder_state_dict(2)), co.as_tensors(synth.generator_state_dict(3))
    x = co.l2_normalize(torch.from_numpy(synth.hg_normal((160, 512), 30)))
    eps = torch.from_numpy(synth.hg_normal((160, 512), 31))
    mean, lv, z, bias = vo.vae_forward(se, sg, x, eps)
    close(mean, g["mean"]); close(lv, g["log_var"]); close(z, g["z"]); close(bias, g["bias"])
    recon = co.l2_normalize(torch.from_numpy(synth.hg_normal((160, 512), 32)))
    close(vo.vae_loss(recon, x, mean, lv), g["vae_loss"])
    close(vo.generator(sg, torch.from_numpy(synth.hg_normal((64, 512), 33))), g["gen_from_z"])
    f = co.l2_normalize(torch.from_numpy(synth.hg_normal((64, 512), 34)))
    close(vo.mlp_net(co.as_tensors(synth.mlp_net_state_dict(4)), f), g["mlp_net"])


@pytest.fixture(scope="module")
def full_sd():
    return co.reference_weight_rounding(synth.clip_state_dict(synth.VIT_B16, 0))


def test_g2_vitb16_image(full_sd):
    """Config C1: encode_image on 4 crops, CPU path."""
    g = dict(np.load(f"{G}/g2_vitb16_image.npz"))
    img = torch.from_numpy(synth.crops(4, 224, seed=1234))
    col = []
    out = co.encode_image(full_sd, img, collect=col)
    assert out.shape == (4, 512) and out.dtype == torch.float32
    close(out, g["encode_image"], rel=1e-4)
    close(torch.stack([c[:, 0, :] for c in col[1:]]), g["cls_after_block"], rel=1e-4)
    close(col[-1][0], g["tok_after_block11_img0"], rel=1e-4)


def test_g2_vitb16_variant_c():
    g = dict(np.load(f"{G}/g2_vitb16_image.npz"))
    sd = co.as_tensors(synth.clip_state_dict(synth.VIT_B16, 0))
    sd.update(co.as_tensors(synth.adapter_state_dict(synth.VIT_B16, 1)))
    img = torch.from_numpy(synth.crops(4, 224, seed=1234))
    pri, mask = synth.priors(4, n=14, dim=64, n_pad=4, seed=99)
    gl, lo = co.visual_with_prior(sd, img, (torch.from_numpy(pri), torch.from_numpy(mask)), range(12))
    close(gl, g["c_prior_global"], rel=1e-4); close(lo, g["c_prior_local"], rel=1e-4)
    gl, lo = co.visual_with_prior(sd, img[:2], None, range(12))
    close(gl, g["c_noprior_global"], rel=1e-4); close(lo, g["c_noprior_local"], rel=1e-4)


def test_g3_vitb16_text_subset(full_sd):
    """First 64 of the 600 HICO prompts + the 81 object prompts (the full 600 run on the GPU test)."""
    g0 = json.load(open(f"{G}/g0_tokens.json"))
    g3 = dict(np.load(f"{G}/g3_vitb16_text.npz"))
    for name, n in (("hoi600", 64), ("obj81", 81)):
        ids = np.zeros((n, 77), np.int64)
        for i in range(n):
            r = g0[name]["ids"][i]
            ids[i, :len(r)] = r
        out = co.encode_text(full_sd, torch.from_numpy(ids))
        close(out, g3[name][:n], rel=1e-4)


def test_g5_prompt_learner_text(full_sd):
    g0 = json.load(open(f"{G}/g0_tokens.json"))
    g5 = dict(np.load(f"{G}/g5_prompt_text.npz"))
    rows = g0["coop_hoi600"]["ids"]
    tok = np.zeros((600, 77), np.int64)
    for i, r in enumerate(rows):
        tok[i, :len(r)] = r
    tok = torch.from_numpy(tok)
    target = torch.from_numpy(g5["target"])
    assert np.array_equal(tok[target].numpy(), g5["tokenized_target"])
    emb = full_sd["token_embedding.weight"][tok]
    ctx = torch.from_numpy(synth.hg_normal((5, 512), 40, 0.02))
    prompts = vo.assemble_prompts(emb[:, :1], emb[:, 6:], ctx, torch.from_numpy(g5["bias"]), target)
    close(prompts[0], g5["prompts_row0"])
    close(co.text_encoder_embeds(full_sd, prompts, tok[target]), g5["text_features"], rel=1e-4)


def test_g8_stress_outlier_channels():
    """Outlier residual channels (x67 the median channel) and c_fc pre-activations ~ 90: the oracle against the
    reference's own outputs (tests/golden/make_golden_stress.py).  The fixture records how hard the case is."""
    g = dict(np.load(f"{G}/g8_stress.npz"))
    rms = g["stream_channel_rms_block5"]
    assert rms.max() / np.median(rms) > 50 and float(g["c_fc_preact_absmax_block0"]) > 60
    sd = co.reference_weight_rounding(synth.stress_clip_state_dict(synth.VIT_B16, 0))
    img = torch.from_numpy(synth.crops(2, 224, seed=1234)[:2])
    close(co.encode_image(sd, img), g["encode_image"][:2], rel=1e-4)
    g0 = json.load(open(f"{G}/g0_tokens.json"))
    rows = g0["hoi600"]["ids"][:8]
    ids = np.zeros((8, 77), np.int64)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
    close(co.encode_text(sd, torch.from_numpy(ids)), g["encode_text"][:8], rel=1e-4)


def test_g10_two_adapter_layers():
    """adapter_num_layers = 2 (reference fixture make_golden_adapter_layers.py): the prior path chains both layers."""
    g = dict(np.load(f"{G}/g10_adapter_layers.npz"))
    raw = synth.clip_state_dict(synth.TINY, 10)
    raw.update(synth.adapter_state_dict(synth.TINY, 13, num_layers=2))
    sd = co.as_tensors(raw)
    img = torch.from_numpy(synth.crops(3, 32, seed=11))
    pri, mask = synth.priors(3, n=6, dim=64, n_pad=2, seed=14)
    gl, ll = co.visual_with_prior(sd, img, (torch.from_numpy(pri), torch.from_numpy(mask)), adapter_layers=range(2))
    close(gl, g["prior_global"], rel=1e-4)
    close(ll, g["prior_local"], rel=1e-4)
    gl, _ = co.visual_with_prior(sd, img, None, adapter_layers=range(2))
    close(gl, g["noprior_global"], rel=1e-4)
