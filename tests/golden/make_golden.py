#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, never on the GPU box).  Imports the
reference's model code with the three stubs SURVEY.md §8c describes (ftfy, torchvision.transforms,
transformer_module), feeds it the deterministic synthetic state dicts of ``hoigen_amd.synth`` and
stores inputs that cannot be regenerated (prompt strings / token ids) plus the reference's outputs.
Nothing of the reference's source is stored: fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--skip-full]

Fixtures (see tests/golden/README.md):
  g0_tokens.json          prompt strings + token ids (reference clipnet.tokenize)
  g1_tiny.npz             tiny config: every intermediate of image / text / adapter / VAE chain
  g2_vitb16_image.npz     ViT-B/16: encode_image of 4 crops, CLS after every block, variant C
  g3_vitb16_text.npz      ViT-B/16: encode_text of the 600+81+117 prompts
  g4_vae.npz              Encoder/Generator/mlp_net/vae_loss on seeded rows
  g5_prompt_text.npz      PromptLearner_hoi -> TextEncoder on 32 targets (full text tower)
"""
import argparse
import importlib.util
import json
import os
import sys
import types
from collections import OrderedDict

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from hoigen_amd import synth  # noqa: E402


# ------------------------------------------------------------------------------------------
# reference import harness
# ------------------------------------------------------------------------------------------

def install_stubs():
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s                       # all prompts are ASCII
    sys.modules["ftfy"] = ftfy
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    for n in ("Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "RandomResizedCrop",
              "RandomHorizontalFlip"):
        setattr(tvt, n, lambda *a, **k: None)
    tvt.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    tm = types.ModuleType("transformer_module")
    tm.TransformerDecoderLayer = object
    tm.TransformerDecoderLayer_womhsa = object
    sys.modules["transformer_module"] = tm


def load_reference():
    install_stubs()
    sys.path.insert(0, REF)
    import clipnet                                     # variant A  (clipnet/clip.py, model.py)
    import CLIP_models_adapter_prior2 as adapter_mod   # variant C
    return clipnet, adapter_mod


def exec_slices(path, ranges, ns):
    """exec 1-based inclusive line ranges of a reference file that cannot be imported whole
    (main_coop_vae.py:13 imports a module that does not exist)."""
    lines = open(path).read().split("\n")
    for a, b in ranges:
        exec(compile("\n".join(lines[a - 1:b]), f"{path}:{a}-{b}", "exec"), ns)
    return ns


def load_vae_classes(clipnet):
    ns = {"torch": torch, "nn": nn, "clip": clipnet,
          "_tokenizer": clipnet.simple_tokenizer.SimpleTokenizer()}
    exec_slices(f"{REF}/main_coop_vae.py", [(32, 39), (45, 63), (66, 128), (131, 193), (196, 258),
                                            (261, 303)], ns)
    exec_slices(f"{REF}/finetune_ship.py", [(302, 314)], ns)
    return ns


def t(sd_np):
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd_np.items())


def build_ref_A(clipnet, cfg, seed=0):
    sd = t(synth.clip_state_dict(cfg, seed))
    model = clipnet.model.build_model(sd)              # fp16 round trip inside (model.py:430)
    return model.float().eval()                        # CPU path: clipnet/clip.py:135-136


def build_ref_C(adapter_mod, cfg, seed=0, adapter_seed=1, adapter_pos="all", trained=True):
    sd = t(synth.clip_state_dict(cfg, seed))
    model = adapter_mod.build_model(sd, use_adapter=True, adapter_pos=adapter_pos, adapter_num_layers=1)
    layers = {"all": range(cfg["vision_layers"]),
              "front": range(cfg["vision_layers"] // 2),
              "end": range(cfg["vision_layers"] // 2, cfg["vision_layers"]),
              "last": range(cfg["vision_layers"] - 1, cfg["vision_layers"])}[adapter_pos]
    asd = t(synth.adapter_state_dict(cfg, adapter_seed, layers=layers, trained=trained))
    missing, unexpected = model.load_state_dict(asd, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "adaptermlp" in k], "adapter keys not covered"
    return model.eval()


def hook_blocks(blocks, store, tuple_out=False):
    hs = []
    for blk in blocks:
        def f(_m, _i, out):
            o = out[0] if tuple_out else out
            store.append(o.detach().permute(1, 0, 2).contiguous().numpy().copy())
        hs.append(blk.register_forward_hook(f))
    return hs


# ------------------------------------------------------------------------------------------
# fixtures
# ------------------------------------------------------------------------------------------

def make_g0(clipnet, adapter_mod):
    ns = {}
    exec(open(f"{REF}/hico_text_label.py").read(), ns)
    lab = {}
    exec(open(f"{REF}/hico_label.py").read(), lab)
    lst = {}
    exec(open(f"{REF}/hico_list.py").read(), lst)
    groups = OrderedDict()
    groups["hoi600"] = list(ns["hico_text_label"].values())
    groups["obj81"] = [s for _, s in ns["hico_obj_text_label"]]
    groups["hum81"] = list(ns["hico_hum_text_label"])
    groups["verb117"] = list(lst["hico_verbs_sentence"])
    # CoOp prompts, main_coop_vae.py:103-108: "X X X X X <name>." with '_' -> ' '
    groups["coop_hoi600"] = ["X X X X X " + n.replace("_", " ") + "." for n in lab["all_classnames"]]
    groups["coop_obj80"] = ["X X X X " + n.replace("_", " ") + "." for n in lab["object_name"]]
    groups["coop_hum80"] = ["X X X X " + n.replace("_", " ") + "." for n in lab["human_name"]]
    groups["misc"] = ["a photo of a person riding a bicycle", "Hello,   World!! it's 3 o'clock &amp; fine",
                      "  leading/trailing   spaces  ", "naïve café — déjà vu", "", "A" * 30 + " 12345 #hash_tag",
                      "<|startoftext|> literal <|endoftext|>", "don't we'll they've I'm he'd you're"]
    out = OrderedDict()
    for g, strs in groups.items():
        ids = clipnet.tokenize(strs)                                  # LongTensor [N,77]
        ids32 = adapter_mod.tokenize(strs)                            # variant C: int32
        assert ids.dtype == torch.int64 and ids32.dtype == torch.int32
        assert torch.equal(ids, ids32.long())
        eot = ids.argmax(dim=-1)                                      # clipnet/model.py:350
        n_tok = [int(torch.nonzero(row).max()) + 1 for row in ids]    # rows are zero padded
        out[g] = {"text": strs,
                  "ids": [row[:n].tolist() for row, n in zip(ids, n_tok)],
                  "eot": eot.tolist()}
    # names whose class index feeds PromptLearner (name_lens in main_coop_vae.py:106)
    out["_classnames"] = {"hoi": lab["all_classnames"], "obj": lab["object_name"], "hum": lab["human_name"]}
    # error behaviour: too-long input raises RuntimeError (clipnet/clip.py:225); truncate=True path
    long_s = " ".join(["word"] * 100)
    tr = clipnet.tokenize([long_s], truncate=True)
    out["_truncate"] = {"text": long_s, "ids": tr[0].tolist()}
    json.dump(out, open(f"{HERE}/g0_tokens.json", "w"), ensure_ascii=False, separators=(",", ":"))
    print("g0: ", {k: len(v["text"]) for k, v in out.items() if not k.startswith("_")})
    return out


@torch.no_grad()
def make_g1(clipnet, adapter_mod, vae_ns):
    cfg = synth.TINY
    res = {}
    mA = build_ref_A(clipnet, cfg, seed=10)
    img = torch.from_numpy(synth.crops(3, cfg["image_resolution"], seed=11))
    store = []
    h = [mA.visual.ln_pre.register_forward_hook(lambda m, i, o: store.append(o.detach().numpy().copy()))]
    h += hook_blocks(mA.visual.transformer.resblocks, store)
    res["img_out"] = mA.encode_image(img).numpy()
    for x in h:
        x.remove()
    res["img_ln_pre"] = store[0]
    res["img_blocks"] = np.stack(store[1:])
    # sub-op intermediates of block 0 (reference modules called one by one)
    x0 = torch.from_numpy(store[0]).permute(1, 0, 2)                  # [L,B,D]
    b0 = mA.visual.transformer.resblocks[0]
    res["img_b0_ln1"] = b0.ln_1(x0).permute(1, 0, 2).numpy()
    att = b0.attention(b0.ln_1(x0))
    res["img_b0_attn"] = att.permute(1, 0, 2).numpy()
    x1 = x0 + att
    res["img_b0_ln2"] = b0.ln_2(x1).permute(1, 0, 2).numpy()
    res["img_b0_fc_gelu"] = b0.mlp.gelu(b0.mlp.c_fc(b0.ln_2(x1))).permute(1, 0, 2).numpy()
    res["img_conv"] = mA.visual.conv1(img).numpy()                    # [B,D,g,g]

    toks = torch.from_numpy(synth.tiny_tokens(6, cfg, seed=12))
    store = []
    h = hook_blocks(mA.transformer.resblocks, store)
    res["txt_out"] = mA.encode_text(toks).numpy()
    for x in h:
        x.remove()
    res["txt_blocks"] = np.stack(store)
    res["txt_tokens"] = toks.numpy()

    # variant C (adapters on every layer, trained-like weights)
    mC = build_ref_C(adapter_mod, cfg, seed=10, adapter_seed=13)
    pri, mask = synth.priors(3, n=6, dim=64, n_pad=2, seed=14)
    prior = (torch.from_numpy(pri), torch.from_numpy(mask))
    g, l = mC.visual(img, prior)
    res["c_prior_global"], res["c_prior_local"] = g.numpy(), l.contiguous().numpy()
    g, l = mC.visual(img, None)
    res["c_noprior_global"], res["c_noprior_local"] = g.numpy(), l.contiguous().numpy()
    a0 = mC.visual.transformer.resblocks[0].adaptermlp
    xin = torch.from_numpy(synth.hg_normal((5, 3, cfg["vision_width"]), 15))       # [L,B,D]
    res["adapter_in"] = xin.permute(1, 0, 2).numpy()
    res["adapter_prior"] = a0(xin, prior).permute(1, 0, 2).numpy()
    res["adapter_noprior"] = a0(xin, None).permute(1, 0, 2).numpy()
    # variant C with untrained adapters == pristine: store the pristine C output
    mC0 = build_ref_C(adapter_mod, cfg, seed=10, adapter_seed=13, trained=False)
    g0, l0 = mC0.visual(img, prior)
    res["c_untrained_global"], res["c_untrained_local"] = g0.numpy(), l0.contiguous().numpy()
    # text tower of variant C (fp32 weights, no fp16 rounding: adapter...:980) for int32 ids
    res["c_txt_out"] = mC.encode_text(toks.int()).numpy()

    # VAE training-step forward chained on the tiny CLIP (main_coop_vae.py:437-468), dims = 128
    D = cfg["transformer_width"]
    R, n_cls, n_ctx = 5, 4, 3
    feats = mA.encode_image(torch.from_numpy(synth.crops(R, cfg["image_resolution"], seed=16))).float()
    feats = feats / feats.norm(dim=-1, keepdim=True)
    We = synth.encoder_state_dict(17, dim=D, hidden=256, wstd=0.05)
    Wg = synth.generator_state_dict(18, dim=D, hidden=384, wstd=0.05)
    E = nn.Sequential()
    E.net = nn.Sequential(nn.Linear(D, 256), nn.ReLU())
    E.mean = nn.Linear(256, D)
    E.log_var = nn.Linear(256, D)
    E.load_state_dict(t(We))
    G = nn.Sequential()
    G.net = nn.Sequential(nn.Linear(D, 384), nn.ReLU(), nn.Linear(384, D))
    G.load_state_dict(t(Wg))
    h1 = E.net(feats)
    mean, log_var = E.mean(h1), E.log_var(h1)
    eps = torch.from_numpy(synth.hg_normal((R, D), 19))
    z = torch.exp(0.5 * log_var) * eps + mean                          # main_coop_vae.py:445-447
    bias = G.net(z)
    cls_tok = torch.from_numpy(synth.tiny_tokens(n_cls, cfg, seed=20))
    emb = mA.token_embedding(cls_tok)
    prefix, suffix = emb[:, :1, :], emb[:, 1 + n_ctx:, :]
    ctx = torch.from_numpy(synth.hg_normal((n_ctx, D), 21, 0.02))
    target = torch.tensor([0, 3, 1, 1, 2])
    prompts = torch.cat([prefix[target], ctx.unsqueeze(0) + bias.unsqueeze(1), suffix[target]], dim=1)
    te = vae_ns["TextEncoder"](mA)
    tf = te(prompts, cls_tok[target])
    tfn = tf / tf.norm(dim=-1, keepdim=True)
    loss = vae_ns["vae_loss"](tfn, feats, mean, log_var, target)
    res.update(vae_feats=feats.numpy(), vae_mean=mean.numpy(), vae_log_var=log_var.numpy(), vae_z=z.numpy(),
               vae_bias=bias.numpy(), vae_cls_tokens=cls_tok.numpy(), vae_target=target.numpy(),
               vae_prompts=prompts.numpy(), vae_text_features=tf.numpy(), vae_loss=loss.numpy())
    np.savez_compressed(f"{HERE}/g1_tiny.npz", **res)
    print("g1: ", {k: v.shape for k, v in res.items()})


@torch.no_grad()
def make_g2(clipnet, adapter_mod):
    cfg = synth.VIT_B16
    res = {}
    mA = build_ref_A(clipnet, cfg, seed=0)
    img = torch.from_numpy(synth.crops(4, 224, seed=1234))
    store = []
    h = hook_blocks(mA.visual.transformer.resblocks, store)
    res["encode_image"] = mA.encode_image(img).numpy()
    for x in h:
        x.remove()
    res["cls_after_block"] = np.stack([s[:, 0, :] for s in store])     # [12,4,768]
    res["tok_after_block11_img0"] = store[-1][0]                        # [197,768]
    del mA
    mC = build_ref_C(adapter_mod, cfg, seed=0, adapter_seed=1)
    pri, mask = synth.priors(4, n=14, dim=64, n_pad=4, seed=99)
    g, l = mC.visual(img, (torch.from_numpy(pri), torch.from_numpy(mask)))
    res["c_prior_global"], res["c_prior_local"] = g.numpy(), l.contiguous().numpy()
    g, l = mC.visual(img[:2], None)
    res["c_noprior_global"], res["c_noprior_local"] = g.numpy(), l.contiguous().numpy()
    np.savez_compressed(f"{HERE}/g2_vitb16_image.npz", **res)
    print("g2: ", {k: v.shape for k, v in res.items()})


@torch.no_grad()
def make_g3(clipnet, g0):
    mA = build_ref_A(clipnet, synth.VIT_B16, seed=0)
    res = {}
    for g in ("hoi600", "obj81", "verb117"):
        ids = clipnet.tokenize(g0[g]["text"])
        outs = [mA.encode_text(ids[i:i + 100]) for i in range(0, len(ids), 100)]
        res[g] = torch.cat(outs).numpy()
    np.savez_compressed(f"{HERE}/g3_vitb16_text.npz", **res)
    print("g3: ", {k: v.shape for k, v in res.items()})
    return mA


@torch.no_grad()
def make_g4(vae_ns):
    res = {}
    E, G, M = vae_ns["Encoder"](), vae_ns["Generator"](), vae_ns["mlp_net"](512, 512, 512)
    E.load_state_dict(t(synth.encoder_state_dict(2)))
    G.load_state_dict(t(synth.generator_state_dict(3)))
    M.load_state_dict(t(synth.mlp_net_state_dict(4)))
    R = 160
    x = torch.from_numpy(synth.hg_normal((R, 512), 30))
    x = x / x.norm(dim=-1, keepdim=True)                               # main_coop_vae.py:438
    eps = torch.from_numpy(synth.hg_normal((R, 512), 31))
    mean, log_var = E(x)
    z = torch.exp(0.5 * log_var) * eps + mean
    bias = G(z)
    res.update(mean=mean.numpy(), log_var=log_var.numpy(), z=z.numpy(), bias=bias.numpy())
    recon = torch.from_numpy(synth.hg_normal((R, 512), 32))
    recon = recon / recon.norm(dim=-1, keepdim=True)
    res["vae_loss"] = vae_ns["vae_loss"](recon, x, mean, log_var, None).numpy()
    zz = torch.from_numpy(synth.hg_normal((64, 512), 33))              # sampling: z ~ N(0,I)
    res["gen_from_z"] = G(zz).numpy()
    f = torch.from_numpy(synth.hg_normal((64, 512), 34))
    f = f / f.norm(dim=-1, keepdim=True)
    res["mlp_net"] = M(f).numpy()
    np.savez_compressed(f"{HERE}/g4_vae.npz", **res)
    print("g4: ", {k: v.shape for k, v in res.items()})


@torch.no_grad()
def make_g5(clipnet, vae_ns, mA, g0):
    # PromptLearner_hoi calls .cuda(): drive it on CPU by making .cuda() the identity
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    names = g0["_classnames"]["hoi"]
    pl = vae_ns["PromptLearner_hoi"](names, mA).float()
    ctx = torch.from_numpy(synth.hg_normal((5, 512), 40, 0.02))
    pl.ctx.data.copy_(ctx)
    G = vae_ns["Generator"]()
    G.load_state_dict(t(synth.generator_state_dict(3)))
    target = torch.tensor([(37 * i + 5) % 600 for i in range(32)])
    z = torch.from_numpy(synth.hg_normal((32, 512), 41))
    bias = G(z)
    pl.get_prefix_suffix_token(names, mA)
    prompts = pl(bias, target)
    te = vae_ns["TextEncoder"](mA).float()
    tf = te(prompts, pl.tokenized_prompts[target])
    res = dict(target=target.numpy(), bias=bias.numpy(), text_features=tf.numpy(),
               prompts_row0=prompts[0].numpy(), tokenized_target=pl.tokenized_prompts[target].numpy(),
               name_lens=np.array(pl.name_lens))
    # same for the 4-token human / object learners (token split differs: n_ctx = 4)
    plo = vae_ns["PromptLearner_o"](g0["_classnames"]["obj"], mA).float()
    plo.ctx.data.copy_(torch.from_numpy(synth.hg_normal((4, 512), 42, 0.02)))
    tgt_o = torch.tensor([(7 * i + 3) % 80 for i in range(16)])
    pr_o = plo(bias[:16], tgt_o)
    res["o_target"] = tgt_o.numpy()
    res["o_text_features"] = te(pr_o, plo.tokenized_prompts[tgt_o]).numpy()
    np.savez_compressed(f"{HERE}/g5_prompt_text.npz", **res)
    print("g5: ", {k: v.shape for k, v in res.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true", help="only g0/g1/g4 (fast)")
    ap.add_argument("--only-g0", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    clipnet, adapter_mod = load_reference()
    vae_ns = load_vae_classes(clipnet)
    g0 = make_g0(clipnet, adapter_mod)
    if args.only_g0:
        return
    make_g1(clipnet, adapter_mod, vae_ns)
    make_g4(vae_ns)
    if not args.skip_full:
        make_g2(clipnet, adapter_mod)
        mA = make_g3(clipnet, g0)
        make_g5(clipnet, vae_ns, mA, g0)


if __name__ == "__main__":
    main()
