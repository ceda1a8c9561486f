"""Diagnostics: the residual GEMMs on sequence tiles (hg_gemm_seq.hip) against gemm_ring2, ViT-B/16 shapes (256 x 197 rows, N = 768,
K = 768 / 3072), five chained launches per call (fp32 stream and centre + hi + lo), hipEvent pairs around every launch, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
L_ = _lib.lib()
n_seq, L, N = int(os.environ.get("NSEQ", 256)), 197, 768
M = n_seq * L
p = lambda t: t.data_ptr()
for K in [int(k) for k in os.environ.get("KS", "768 3072").split()]:
    g = torch.Generator(device="cuda").manual_seed(K)
    a = torch.randn(M, K, device="cuda", generator=g)
    if K == 3072:
        a = a * torch.sigmoid(1.702 * a)          # the MLP's QuickGELU output: half of it near zero (what c_proj really reads)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g)
    x0 = torch.randn(M, N, device="cuda", generator=g)
    mu0 = x0.mean(1)
    for hilo in (0, 1):
        def run(kernel):
            x, mu = x0.clone(), mu0.clone()
            if kernel == "seq":
                rc = L_.hg_test_gemm_seq(ctx, p(a), p(w), p(bias), p(x), n_seq, L, N, K, 5, hilo, p(mu), None, None, None)
            else:
                rc = L_.hg_test_gemm_hilo(ctx, p(a), p(w), p(bias), p(x), M, N, K, 5, hilo, p(mu), None, None, None)
            assert rc == 0, L_.hg_last_error(ctx)
        res = {}
        for rnd in range(3):
            for kernel in ("ring2", "seq"):
                run(kernel)
                _, recs = _lib.profile(ctx, 10, 64, lambda: [run(kernel) for _ in range(3)] and torch.cuda.synchronize())
                us = sorted(r[4] * 1e3 for r in recs)
                res.setdefault(kernel, []).append((us[len(us) // 2], us[0]))
        print(f"K={K} hilo={hilo}: " + " | ".join(f"{k}: median {min(m for m, _ in v):.1f} min {min(n for _, n in v):.1f} us" for k, v in res.items()), flush=True)
