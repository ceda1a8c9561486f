"""Soak test of the MLP pair launch's hand-off: many launches, every output compared bit for bit with the two-launch result.
Vision tower at 256 / 171 / 40 crops over several tile orders / work splits (they shift which workgroup waits for which), then the text
tower.  A stale read through the ready counters would show as a mismatch in some launch; a wait that gave up as an error from the library
(HG_ERR_HIP on the next call).  usage: python tools/mlp_pair_soak.py [steps per configuration]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import synth
from hoigen_amd.model import build_model

torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = torch.device("cuda:0")
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(d)
m.visual.set_option("last_block_row0", 0)
g = torch.Generator(device=d).manual_seed(5)
total = bad = 0
t0 = time.time()
for crops in (256, 171, 40):
    x = torch.randn(crops, 3, 224, 224, device=d, generator=g)
    m.visual.set_option("mlp_pair", 0)
    want = m.encode_image(x).clone()
    for pair, ch, slots in ((1, 32, 32), (2, 32, 32), (1, 8, 30), (1, 3, 24), (1, 1, 32), (2, 25, 28)):
        m.visual.set_option("mlp_pair", pair); m.visual.set_option("mlp_pair_chunk", ch); m.visual.set_option("mlp_pair_fc_slots", slots)
        mism = 0
        for i in range(N):
            got = m.encode_image(x)
            if not torch.equal(got, want):
                mism += 1
        total += N; bad += mism
        print(f"vision {crops} crops, mlp_pair {pair} chunk {ch} fc_slots {slots}: {N} launches x 11 pair kernels, mismatching outputs {mism}", flush=True)
m.visual.set_option("mlp_pair", 1); m.visual.set_option("mlp_pair_chunk", 32); m.visual.set_option("mlp_pair_fc_slots", 32)
for prompts, L in ((600, 77), (81, 77), (600, 16)):
    tok = torch.zeros(prompts, 77, dtype=torch.int64, device=d)
    tok[:, 0] = 49406
    body = torch.randint(1000, 40000, (prompts, 77), device=d, generator=g)
    eot = torch.randint(4, L, (prompts,), device=d, generator=g)
    ar = torch.arange(77, device=d)[None]
    tok = torch.where((ar > 0) & (ar < eot[:, None]), body, tok)
    tok[torch.arange(prompts, device=d), eot] = 49407
    m.set_option("mlp_pair", 0)
    want = m.encode_text(tok).clone()
    for pair in (1, 2):
        m.set_option("mlp_pair", pair)
        mism = 0
        for i in range(N):
            if not torch.equal(m.encode_text(tok), want):
                mism += 1
        total += N; bad += mism
        print(f"text {prompts} prompts (EOT < {L}), mlp_pair {pair}: {N} calls x 11 pair kernels, mismatching outputs {mism}", flush=True)
torch.cuda.synchronize()
print(f"TOTAL {total} tower calls ({total * 11} pair launches), mismatching {bad}, {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
