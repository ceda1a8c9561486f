"""Time per launch of the row-owner residual GEMM (kernel 5) / ring2 (kernel 2) at the ViT-B/16 shapes, hipEvents via hg_profile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0); lib = _lib.lib()
M = int(os.environ.get("M", 197 * 256))
for N, K in ((768, 768), (768, 3072)):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    x0 = torch.randn(M, N, device="cuda", generator=g); mu = x0.mean(1)
    out = x0.clone(); out2 = torch.empty(M, N, device="cuda"); mr = torch.empty(M, 2, device="cuda"); muo = torch.empty(M, device="cuda")
    res = []
    for kernel in [int(k) for k in os.environ.get("KERNELS", "5").split()]:
        def f():
            for _ in range(int(os.environ.get("ITERS", 8))):
                rc = lib.hg_test_gemm_ln(ctx, a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, 10, kernel, None, None,
                                         mu.data_ptr(), None, out2.data_ptr(), mr.data_ptr(), muo.data_ptr(), None)
                assert rc == 0
        f()
        _, recs = _lib.profile(ctx, 10, 64, f)
        ts = sorted(r[4] for r in recs if r[1] == M)
        res.append(f"k{kernel} {ts[len(ts)//2]*1e3:.1f}")
    print(f"K={K}: " + " ".join(res), end="  |  ")
print()
