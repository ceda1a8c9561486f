"""Diagnostics: run the ViT-B/16 GEMM shapes through hg_test_gemm with HG_STAMPS=1 (stderr gets the per-phase cycles)."""
import os, sys, time
os.environ["HG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
def say(*a):
    print("[%.1fs]" % (time.time() - t0), *a, flush=True)
import torch
say("torch imported")
from hoigen_amd import _lib
ctx = _lib.ctx(0)
say("ctx")
M = int(os.environ.get("M", 197 * 256))
for (N, K, epi) in [(2304, 768, 0), (3072, 768, 1), (768, 3072, 3), (768, 768, 3)][int(os.environ.get("SHAPE0", 0)):int(os.environ.get("NSHAPES", 4))]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.02
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    torch.cuda.synchronize()
    say("inputs", N, K)
    for it in range(2):
        rc = _lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, int(os.environ.get("KERNEL", 2)), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        say("done", N, K, epi, it)
