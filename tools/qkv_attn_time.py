"""Diagnostics: the fused in_proj + attention kernel (hg_qkv_attn.hip) against the two kernels it replaces, ViT-B/16 shape
(256 sequences x 197 tokens x 12 heads), hipEvent pairs around every launch (hg_profile_*), alternating, random operands.
GSZ="6 3 2 1": XCD group sizes to time.  In an -DHG_EXPERIMENTS build HG_QA_MODE knocks parts out (hg_qkv_attn.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
L_ = _lib.lib()
n_seq, L, heads = int(os.environ.get("NSEQ", 256)), int(os.environ.get("L", 197)), int(os.environ.get("HEADS", 12))
D = heads * 64
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(n_seq * L, D, device="cuda", generator=g)
w = torch.randn(3 * D, D, device="cuda", generator=g) * D ** -0.5
bias = torch.randn(3 * D, device="cuda", generator=g) * 0.3
cs = w.half().float().sum(1)
mr = torch.stack([torch.randn(n_seq * L, device="cuda", generator=g) * 0.05, torch.rand(n_seq * L, device="cuda", generator=g) + 0.5], 1).contiguous()
out = torch.empty(n_seq * L, D, device="cuda")


def run(fused):
    rc = L_.hg_test_qkv_attn(ctx, a.data_ptr(), w.data_ptr(), bias.data_ptr(), cs.data_ptr(), mr.data_ptr(), n_seq, L, heads, fused,
                             out.data_ptr(), None)
    assert rc == 0, L_.hg_last_error(ctx)


def timed(fused, iters):
    def body():
        for _ in range(iters):
            run(fused)
        torch.cuda.synchronize()
    _, recs = _lib.profile(ctx, _lib.HG_PROF_ALL, 4 * iters + 8, body)
    by = {}
    for kind, M, N, K, ms in recs:
        by.setdefault(kind, []).append(ms * 1e3)
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in by.items()}


iters = int(os.environ.get("ITERS", 12))
for _ in range(2):
    run(0); run(1)
for rnd in range(int(os.environ.get("ROUNDS", 3))):
    u = timed(0, iters)
    line = f"round {rnd}: separate: " + "  ".join(f"kind {k}: median {m:.1f} min {mn:.1f} us" for k, (m, mn) in sorted(u.items()))
    line += f"  sum {sum(m for m, _ in u.values()):.1f} us |"
    for gsz in [int(x) for x in os.environ.get("GSZ", "0").split()]:
        L_.hg_set_option(ctx, b"qkv_attn_gsz", gsz)
        f = timed(1, iters)
        line += f" fused gsz {gsz}: median {f[_lib.HG_PROF_QKV_ATTN][0]:.1f} min {f[_lib.HG_PROF_QKV_ATTN][1]:.1f} us |"
    print(line, flush=True)
