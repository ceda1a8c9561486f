#!/bin/bash
# ring2 (kernel 2) vs the lock-step flow loop (kernel 4, ab/flow.so built from tools/experiments/round3) on few workgroups,
# where the clock is pinned at its maximum and wall time = cycles: us per launch, plain residual epilogue (epi 3)
export HG_LIB_PATH=/root/repo/ab/flow.so
for g in 16 256; do for k in 2 4; do
HG_RING2_GRID=$g KERNEL=$k python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from hoigen_amd import _lib
ctx = _lib.ctx(0)
k = int(os.environ["KERNEL"]); g = os.environ["HG_RING2_GRID"]
res = []
for (M, N, K) in [(128 * 64, 768, 3072), (128 * 64, 768, 768), (197 * 256, 768, 3072), (197 * 256, 768, 768)]:
    if int(g) == 16 and M > 10000: continue
    if int(g) == 256 and M < 10000: continue
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda")
    for it in range(3):
        _lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 3, k, None)
    torch.cuda.synchronize()
    _, recs = _lib.profile(ctx, _lib.HG_PROF_ALL, 16, lambda: [_lib.lib().hg_test_gemm(ctx, a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 3, k, None) for _ in range(5)])
    res.append("M=%d K=%d: %.1f us" % (M, K, sum(r[4] for r in recs) / len(recs) * 1e3))
print("grid", g, "kernel", k, " | ".join(res))
PY
done; done
