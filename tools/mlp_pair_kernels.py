"""Per-kernel means of the headline step (256 crops, every row) under the library / options the environment selects (HG_LIB_PATH,
HG_MLP_PAIR, HG_MLP_PAIR_LAG, HG_PAIR_ONLY ...): hipEvent pairs around every GEMM / attention launch (hg_profile), 6 steps; then the
step time without events over 20 steps.  One line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib, synth
from hoigen_amd.model import build_model

torch.set_grad_enabled(False)
d = torch.device("cuda:0")
m = build_model(synth.to_torch(synth.clip_state_dict(synth.VIT_B16, 0))).to(d)
m.visual.set_option("last_block_row0", 0)
x = torch.randn(256, 3, 224, 224, device=d)
for _ in range(3): m.encode_image(x)
torch.cuda.synchronize()
steps = 6
_, recs = _lib.profile(m.visual._ctx.handle, _lib.HG_PROF_ALL, steps * 128, lambda: [m.encode_image(x) for _ in range(steps)])
agg = {}
for kind, M, N, K, ms in recs:
    a = agg.setdefault((kind, M, N, K), [0, 0.0]); a[0] += 1; a[1] += ms
names = {(9, 50432, 3072, 768): "c_fc", (10, 50432, 768, 3072): "c_proj", (10, 50432, 768, 768): "out_proj", (103, 50432, 3072, 768): "PAIR",
         (101, 256, 197, 12): "qkv_attn", (3, 50432, 768, 3072): "c_proj_last", (5, 50176, 768, 768): "patch"}
parts = []
for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if k in names: parts.append(f"{names[k]} {tot / n * 1e3:.1f}us x{n // steps}")
for _ in range(3): m.encode_image(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): m.encode_image(x)
e1.record(); torch.cuda.synchronize()
print(f"[{os.environ.get('TAG', '?')}] step {e0.elapsed_time(e1) / 20:.3f} ms | " + " | ".join(parts), flush=True)
