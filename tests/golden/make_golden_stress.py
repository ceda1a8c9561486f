#!/usr/bin/env python3
"""g8_stress.npz: the REFERENCE (clipnet, fp32 CPU path) on the outlier / large-activation state dict of
``hoigen_amd.synth.stress_clip_state_dict`` (residual channels x50-100, c_fc pre-activations in the
hundreds).  Build container only (needs /root/reference); the fixture holds outputs plus a few statistics that
document how hard the case is.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stress.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from hoigen_amd import synth  # noqa: E402


@torch.no_grad()
def main():
    torch.set_num_threads(os.cpu_count())
    clipnet, _ = mg.load_reference()
    sd = mg.t(synth.stress_clip_state_dict(synth.VIT_B16, 0))
    model = clipnet.model.build_model(sd).float().eval()
    img = torch.from_numpy(synth.crops(4, 224, seed=1234))
    stream, pre = [], []
    hs = mg.hook_blocks(model.visual.transformer.resblocks, stream)
    hs.append(model.visual.transformer.resblocks[0].mlp.c_fc.register_forward_hook(
        lambda m, i, o: pre.append(o.detach().numpy().copy())))
    res = {"encode_image": model.encode_image(img).numpy()}
    for h in hs:
        h.remove()
    x = stream[5]                                            # residual stream after block 5: [4,197,768]
    ch_rms = np.sqrt((x ** 2).mean(axis=(0, 1)))
    res["stream_channel_rms_block5"] = ch_rms.astype(np.float32)
    res["c_fc_preact_absmax_block0"] = np.float32(np.abs(pre[0]).max())
    g0 = json.load(open(f"{HERE}/g0_tokens.json"))
    ids = clipnet.tokenize(g0["hoi600"]["text"][:64])
    res["encode_text"] = model.encode_text(ids).numpy()
    np.savez_compressed(f"{HERE}/g8_stress.npz", **res)
    med = np.median(ch_rms)
    print("g8:", {k: getattr(v, "shape", v) for k, v in res.items()})
    print(f"residual channel rms after block 5: median {med:.3f}, max {ch_rms.max():.1f} (x{ch_rms.max() / med:.0f}); "
          f"|c_fc pre-activation| max in block 0: {res['c_fc_preact_absmax_block0']:.0f}")


if __name__ == "__main__":
    main()
