"""Diagnostics: the two residual GEMM shapes of ViT-B/16 through hg_test_gemm (KERNEL = 2 ring, 3 ring + stream-K
flags), a few launches each; run under `rocprofv3 --kernel-trace --stats` and read the kernel durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
M = int(os.environ.get("M", 197 * 256))
shapes = {"cproj": (768, 3072, 3), "outproj": (768, 768, 3), "qkv": (2304, 768, 0), "cfc": (3072, 768, 1)}
for name in os.environ.get("SHAPES", "cproj outproj").split():
    N, K, epi = shapes[name]
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.02
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    for it in range(int(os.environ.get("ITERS", 6))):
        rc = _lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi,
                                     int(os.environ.get("KERNEL", 2)), None)
        assert rc == 0, rc
    torch.cuda.synchronize()
