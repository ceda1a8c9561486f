"""Diagnostics: c_fc + QuickGELU on sequence tiles (hg_gemm_seq.hip) against the 256 x 256 ring, ViT-B/16 shape (256 x 197 rows,
N = 3072, K = 768; QKV = "8 2304"), hipEvent pairs around every launch, alternating.  GSZ="0 4 2": XCD group sizes to time; STAG="0 18000:3": start staggers (cycles:phases)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
L_ = _lib.lib()
n_seq, L = int(os.environ.get("NSEQ", 256)), 197
epi, N = [int(v) for v in os.environ.get("SHAPE", "9 3072").split()]
K, M = 768, n_seq * L
p = lambda t: t.data_ptr()
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(M, K, device="cuda", generator=g)
w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
bias = torch.randn(N, device="cuda", generator=g)
cs = w.half().float().sum(1)
mr = torch.stack([torch.randn(M, device="cuda", generator=g) * 0.05, torch.rand(M, device="cuda", generator=g) + 0.5], 1).contiguous()
out = torch.empty(M, N, device="cuda")


def run(kernel):
    rc = L_.hg_test_gemm_ln(ctx, p(a), p(w), p(bias), p(out), M, N, K, epi, kernel, p(cs), p(mr), None, None, None, None, None, None)
    assert rc == 0, L_.hg_last_error(ctx)


def timed(kernel, iters=6):
    run(kernel)
    _, recs = _lib.profile(ctx, epi, 64, lambda: [run(kernel) for _ in range(iters)] and torch.cuda.synchronize())
    us = sorted(r[4] * 1e3 for r in recs)
    return us[len(us) // 2], us[0]


for rnd in range(int(os.environ.get("ROUNDS", 3))):
    line = "round %d: ring: median %.1f min %.1f us |" % ((rnd,) + timed(2))
    for gsz in [int(x) for x in os.environ.get("GSZ", "0").split()]:
        L_.hg_set_option(ctx, b"seq_fc_gsz", gsz)
        for st in os.environ.get("STAG", "0").split():
            cyc, ph = ([int(v) for v in st.split(":")] + [0])[:2]
            L_.hg_set_option(ctx, b"seq_fc_stagger", (cyc << 4) | ph)
            line += " seq gsz %d stag %s: median %.1f min %.1f us |" % ((gsz, st) + timed(1000 + L))
    print(line, flush=True)
