"""Stress of the MLP pair launch from two streams of one process: encode_image on one stream, encode_text on another (two contexts), 40
rounds per batch size; with HG_LIB_PATH=ab/nogate.so (-DHG_NO_PAIR_GATE build of hg_api.hip) the same without the library's cross-stream gate."""
import os, sys, time, json
sys.path.insert(0, "/root/repo")
import torch
from hoigen_amd import synth, clip
from hoigen_amd.model import build_model
torch.set_grad_enabled(False)
d = torch.device("cuda:0")
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(d)
m.truncate_text = False
g0 = json.load(open("/root/repo/tests/golden/g0_tokens.json"))
ids = clip.tokenize(g0["hoi600"]["text"]).to(d)
bad = 0
for crops in (256, 40, 8):
    x = torch.randn(crops, 3, 224, 224, device=d)
    wi, wt = m.encode_image(x).clone(), m.encode_text(ids).clone()
    torch.cuda.synchronize()
    ss = [torch.cuda.Stream(device=d) for _ in range(3)]
    t0 = time.time()
    try:
        for it in range(40):
            outs = []
            with torch.cuda.stream(ss[0]): a = m.encode_image(x)
            with torch.cuda.stream(ss[1]): t = m.encode_text(ids)
            a2 = a      # (a context serves ONE stream at a time: the vision tower's context on ss[0], the text tower's on ss[1])
            if it % 10 == 9:
                torch.cuda.synchronize()
                if not (torch.equal(a, wi) and torch.equal(t, wt) and torch.equal(a2, wi)): bad += 1
        torch.cuda.synchronize()
    except RuntimeError as e:
        print("RAISED", str(e)[:200]); bad += 100
    print(f"{crops} crops: 40 rounds x (image | text) on two streams: {time.time() - t0:.2f} s, mismatches/errors {bad}", flush=True)
try:
    m.encode_image(x); m.encode_text(ids); torch.cuda.synchronize(); print("after: no error pending")
except RuntimeError as e:
    print("RAISED after", str(e)[:200])
