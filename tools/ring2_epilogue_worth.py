import os, sys
sys.path.insert(0, "/root/repo")
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0); L_ = _lib.lib()
M, N = 256 * 197, 768
p = lambda t: t.data_ptr()
for K in (768, 3072):
    g = torch.Generator(device="cuda").manual_seed(K)
    a = torch.randn(M, K, device="cuda", generator=g)
    if K == 3072: a = a * torch.sigmoid(1.702 * a)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g)
    x0 = torch.randn(M, N, device="cuda", generator=g); mu0 = x0.mean(1)
    def run():
        x, mu = x0.clone(), mu0.clone()
        rc = L_.hg_test_gemm_hilo(ctx, p(a), p(w), p(bias), p(x), M, N, K, 5, 1, p(mu), None, None, None)
        assert rc == 0
    run()
    _, recs = _lib.profile(ctx, 10, 64, lambda: [run() for _ in range(4)] and torch.cuda.synchronize())
    us = sorted(r[4] * 1e3 for r in recs)
    print(f"K={K} HG_RING_MODE={os.environ.get('HG_RING_MODE','0')}: median {us[len(us)//2]:.1f} min {us[0]:.1f} us", flush=True)
