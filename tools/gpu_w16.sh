#!/bin/bash
python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from hoigen_amd import _lib
h = _lib.lib().hg_create(0)
ok = True
for (M, N, K) in [(2048, 768, 768), (197 * 256, 2304, 768), (8192 - 60, 2048, 320), (256 * 33 + 100, 2048, 256)]:
    for epi in (0, 1, 4):
        g = torch.Generator(device="cuda").manual_seed(M + epi)
        a = torch.randn(M, K, device="cuda", generator=g); w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
        b = torch.randn(N, device="cuda", generator=g)
        outs = []
        for k in (1, 4):
            out = torch.empty(M, N, device="cuda")
            rc = _lib.lib().hg_test_gemm(h, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, k, None)
            assert rc == 0, (rc, _lib.lib().hg_last_error(h))
            torch.cuda.synchronize(); outs.append(out)
        eq = torch.equal(outs[0], outs[1]); ok &= eq
        print(M, N, K, epi, "equal" if eq else ("DIFF max %.3e" % (outs[0] - outs[1]).abs().max().item()))
print("ALL EQUAL" if ok else "MISMATCH")
PY
SHAPES="qkv cfc" KERNELS="2 3 4" ROUNDS=5 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids
