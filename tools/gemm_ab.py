"""A/B timing of the GEMM kernels on the four ViT-B/16 block shapes (M = 197 * 256) through hg_test_gemm:
kernel 1 = simple 128x128, 2 = ring dispatcher (256x256 / 128x256 ring2), 3 = duo (two workgroups per CU).
Interleaved rounds in one process, hipEvent pairs around the kernel launch only (hg_profile_*).
    SHAPES="outproj cproj qkv cfc" KERNELS="2 3" ROUNDS=5 python tools/gemm_ab.py
"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hoigen_amd import _lib

h = _lib.lib().hg_create(0)
M = int(os.environ.get("M", 197 * 256))
shapes = {"cproj": (768, 3072, 3), "outproj": (768, 768, 3), "qkv": (2304, 768, 0), "cfc": (3072, 768, 1),
          "cproj4": (768, 3072, 4), "outproj4": (768, 768, 4), "cproj10": (768, 3072, 10), "outproj10": (768, 768, 10),
          "vae1": (2048, 512, 2), "vae3": (4096, 512, 2)}
kernels = [int(k) for k in os.environ.get("KERNELS", "2 3").split()]
rounds = int(os.environ.get("ROUNDS", 5))
for name in os.environ.get("SHAPES", "outproj cproj qkv cfc").split():
    N, K, epi = shapes[name]
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.02
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    res = {k: [] for k in kernels}
    outs = {}
    for r in range(rounds + 1):
        for k in kernels:
            out.zero_()
            def call():
                rc = _lib.lib().hg_test_gemm(h, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, k, None)
                assert rc == 0, (rc, _lib.lib().hg_last_error(h))
            _, recs = _lib.profile(h, _lib.HG_PROF_ALL, 4, call)
            if r:
                res[k].append(recs[0][4] * 1e3)
            else:
                outs[k] = out.clone()
    fl = 2.0 * M * N * K
    line = f"{name:9s} M={M} N={N} K={K} epi={epi}: "
    for k in kernels:
        med = statistics.median(res[k])
        line += f" k{k}: {med:7.1f} us (min {min(res[k]):.1f}) {fl / med / 1e6:7.1f} TF/s |"
    if len(kernels) > 1:
        line += " equal=" + str(all(torch.equal(outs[kernels[0]], outs[k]) for k in kernels[1:]))
    print(line, flush=True)
